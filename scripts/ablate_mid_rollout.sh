#!/bin/bash
# Timing-only knock-outs of the mid-size rollout engines (TW_ABLATE build, wrong results): kernel ms at 32,768 envs.
# TW_ENG_DBG: 1 no gather reads, 2 no A-operand loads, 4 no table (/ W1) streams.  Run on the GPU box from the repo root.
out=$PWD/gpurun_out/ablate_mid; mkdir -p $out; : > $out/log.txt
export TW_ABLATE=1
for dbg in 0 1 2 4 6 7; do
  echo "G dbg $dbg: $(TW_MID_G=1 TW_ENG_DBG=$dbg python3 scripts/mid_one.py 32768 3 2>/dev/null | tail -1)" | tee -a $out/log.txt
done
for dbg in 0 4; do
  echo "S dbg $dbg: $(TW_ENG_DBG=$dbg python3 scripts/mid_one.py 32768 3 2>/dev/null | tail -1)" | tee -a $out/log.txt
done
