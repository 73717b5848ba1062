#!/bin/bash
# Timing-only knock-outs of the 16-column engine's step (Engine3T, tw_engine.hpp: -DTW_KNOCK=bits -- 1 no row reads, 2 no A-operand loads,
# 4 no table streams, 8 no MFMAs, 16 no closing wait, 32 no barrier at the end of a step, 64 no add chains; results are wrong, only the times mean something), as variants of the PRODUCT build:
#   here:        for k in 8 6 9 31; do TW_VARIANT=knock$k TW_EXTRA_FLAGS=-DTW_KNOCK=$k python -m twisterl_amd.build; done
#                (twisterl_amd/lib/variants/ is in .gpurunignore: take that line out for the call, and delete the variants' asm/ and objects first)
#   GPU box:     bash scripts/ablate_small_rollout.sh
for k in "" knock8 knock6 knock9 knock31; do
  echo "variant ${k:-product}: $(TW_VARIANT=$k python3 scripts/bench_small_rollout.py 2>/dev/null | grep '"envs": 1024, "geom": 0' | sed -n 's/.*"rollout_ms": \([0-9.]*\).*/\1 ms/p')"
done
