#!/bin/bash
# cycle stamps of the 16-column engines' forward (diagnostic TW_ABLATE build), with and without the weight / table streams.
# Run on the GPU box from the repo root.
set -e
out=$PWD/gpurun_out/smallst; mkdir -p $out
export TW_ABLATE=1   # the instrumented library lives in twisterl_amd/lib/ablate/ and is loaded only while this is set
python3 -m twisterl_amd.build > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
for dbg in 0; do
  echo "== TW_ENG_DBG=$dbg"
  TW_ENG_DBG=$dbg TW_STAMPS=1 python3 scripts/bench_small_rollout.py 2>&1 | grep -v amdgpu.ids | grep -A1 '"envs": 4096' | tee -a $out/stamps.log
done
