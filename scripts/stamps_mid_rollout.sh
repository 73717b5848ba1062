#!/bin/bash
# cycle stamps of the mid-size rollout engines' forward (diagnostic TW_ABLATE build), with and without memory traffic.
out=$PWD/gpurun_out/stamps_mid; mkdir -p $out; : > $out/log.txt
export TW_ABLATE=1 TW_STAMPS=1
for envs in 32768; do
for dbg in 0 7; do
  echo "== G (two workgroups per CU), $envs envs, TW_ENG_DBG=$dbg" | tee -a $out/log.txt
  TW_MID_G=1 TW_ENG_DBG=$dbg python3 scripts/mid_one.py $envs 2 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $out/log.txt
done
for dbg in 0 4; do
  echo "== S (one workgroup per CU), $envs envs, TW_ENG_DBG=$dbg" | tee -a $out/log.txt
  TW_ENG_DBG=$dbg python3 scripts/mid_one.py $envs 2 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $out/log.txt
done
done
