#!/bin/bash
# self-play shape grid (diagnostic): walkers per workgroup x engine width, per batch size.  Run on the GPU box from the repo root.
# TW_OPT_AZ_VARIANT: 4 / 3 / 5 / 6 = one / two / four / eight walkers, + 16 = 16-column engine, + 32 = 32-column engine
out=$PWD/gpurun_out/azgrid; mkdir -p $out
S=${1:-100}
for E in ${AZ_GRID_E:-256 384 512 768 1024 1536 2048 3072 4096}; do
  line="$E x $S:"
  for v in 20 19 21 36 35 37 38; do
    ms=$(timeout -k 10 120 python3 scripts/bench_az.py --envs $E --searches $S --variant $v 2>/dev/null | sed -n 's/.*mcts_kernel_ms": \([0-9.]*\).*/\1/p')
    line="$line  v$v $(printf %.1f ${ms:-0})"
  done
  echo "$line" | tee -a $out/grid_$S.txt
done
