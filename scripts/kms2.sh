#!/bin/bash
for e in 65536 131072 262144; do
  timeout -k 10 300 python bench.py --precision fp16x2 --envs $e --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($e, 'kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'])"
done
