#!/bin/bash
# cycle stamps of the generic engine's forward (diagnostic TW_ABLATE build).  Run on the GPU box from the repo root.
set -e
out=$PWD/gpurun_out/genst; mkdir -p $out
export TW_ABLATE=1   # the instrumented library lives in twisterl_amd/lib/ablate/ and is loaded only while this is set
python3 -m twisterl_amd.build > $out/build.log 2>&1
TW_STAMPS=1 python3 scripts/bench_generic_engine.py 2>&1 | grep -v amdgpu.ids | tee $out/stamps.log
