#!/bin/bash
# diagnostic: run the f16 rollout stamps for the experiment builds prepared under twisterl_amd/lib_exp<N>/
for n in "$@"; do
  cp twisterl_amd/lib_exp$n/libtwisterl_hip.so twisterl_amd/lib/libtwisterl_hip.so
  echo "== exp $n"
  TW_STAMPS=1 timeout -k 10 120 python bench.py --precision fp16 --envs 65536 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep -E "stamps16|kernel_ms" | sed 's/.*kernel_ms": \([0-9.]*\).*/kernel_ms \1/' | cut -c1-300
done
