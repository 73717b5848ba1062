#!/bin/bash
# walker kernel: tree-walk budgets (max = stop at the next search boundary; min = stop there once another walker waits), ms per collect
t() { python3 scripts/bench_az.py --envs $1 --searches $2 --variant ${3:-0} --tree-budget $4 --tree-budget-min $5 2>/dev/null | sed -n 's/.*mcts_kernel_ms": \([0-9.]*\).*/\1/p'; }
for cfg in "$@"; do set -- $cfg
  line="$1 x $2 variant $3:"
  for mm in ${AZ_BUDGETS:-72000/72000 72000/16000 72000/32000 150000/16000 150000/32000 300000/16000 300000/32000 300000/64000}; do set -- $cfg ${mm%/*} ${mm#*/}
    line="$line  [$4/$5] $(printf %.1f $(t $1 $2 $3 $4 $5))"
  done
  echo "$line"
done
