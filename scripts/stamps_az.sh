#!/bin/bash
# cycle stamps of the self-play kernels (diagnostic TW_ABLATE build).  Run on the GPU box from the repo root.
set -e
out=$PWD/gpurun_out/azst; mkdir -p $out
TW_ABLATE=1 python3 -m twisterl_amd.build --force > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
for cfg in "1024 100" "4096 100"; do
  set -- $cfg
  echo "== $1 x $2"
  TW_STAMPS=1 python3 scripts/bench_az.py --envs $1 --searches $2 --steps 1 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $out/stamps.log
done
