#!/bin/bash
# Timing-only ablations of the 8-wave rollout kernel (TW_ABLATE build, wrong results): kernel ms at 65,536 envs (one round of 256
# workgroups) without the gather (1), the A-operand reads (2), the weight streams (4), the heads / sampling (8).  GPU box, repo root.
out=$PWD/gpurun_out/ablate; mkdir -p $out
export TW_ABLATE=1   # the instrumented library lives in twisterl_amd/lib/ablate/ and is loaded only while this is set
python3 -m twisterl_amd.build > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
for dbg in 0 1 2 4 8; do
  echo "dbg $dbg: $(TW_ROLLOUT_DBG=$dbg python3 bench.py --envs 65536 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys, json; print(json.loads(sys.stdin.readline())["roofline"]["kernel_ms"], "ms")')" | tee -a $out/ablate.log
done
python3 -m twisterl_amd.build --force > $out/build_final.log 2>&1
