#!/bin/bash
# quick self-play timing set (ms per collect): automatic shapes, or "E S variant" triples from the command line.  GPU box, repo root.
t() { python3 scripts/bench_az.py --envs $1 --searches $2 --variant ${3:-0} 2>/dev/null | sed -n 's/.*mcts_kernel_ms": \([0-9.]*\).*"threads": \([0-9]*\), "blocks": \([0-9]*\).*/\1 ms (\2 x \3)/p'; }
if [ $# -gt 0 ]; then
  for cfg in "$@"; do set -- $cfg; echo "$1 x $2, variant ${3:-0}: $(t $1 $2 $3)"; done
  exit 0
fi
for cfg in "256 100" "1024 100" "2048 100" "4096 100" "4096 1000" "512 1000" "6144 100"; do set -- $cfg; echo "$1 x $2: $(t $1 $2)"; done
