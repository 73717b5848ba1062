#!/bin/bash
# HBM traffic of the rollout kernels: separate rocprofv3 PMC passes for FETCH_SIZE and WRITE_SIZE (run on the GPU box)
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/traffic; mkdir -p $out
for p in fp32 fp16x2 fp16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $out/${p}_$c -o p -f csv -- python3 $OLDPWD/bench.py --precision $p --steps 1 --warmup 0 --no-cpu-baseline > $out/${p}_$c.log 2>&1)
  done
done
ls $out
