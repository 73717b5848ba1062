#!/bin/bash
# cycle stamps of the self-play kernels (diagnostic TW_ABLATE build).  Run on the GPU box from the repo root.
set -e
out=$PWD/gpurun_out/azst; mkdir -p $out
export TW_ABLATE=1   # the instrumented library lives in twisterl_amd/lib/ablate/ and is loaded only while this is set
python3 -m twisterl_amd.build > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
# AZ_VARIANTS: TW_OPT_AZ_VARIANT values to stamp (default: the automatic shape); AZ_SEARCHES: searches per move
for E in 1024 4096; do
  for v in ${AZ_VARIANTS:-0}; do
    echo "== $E x ${AZ_SEARCHES:-100}, variant $v" | tee -a $out/stamps.log
    TW_STAMPS=1 python3 scripts/bench_az.py --envs $E --searches ${AZ_SEARCHES:-100} --steps 1 --variant $v 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $out/stamps.log
  done
done
