#!/bin/bash
# print the rollout kernel time of a short bench run for each precision given
for p in "$@"; do
  timeout -k 10 300 python bench.py --precision $p --envs 65536 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$p', 'kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'])"
done
