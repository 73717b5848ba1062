#!/bin/bash
# small-batch measurement set (run on the GPU box from the repo root): config 1 (Puzzle-8, 1,024 envs), the geometry sweep,
# AlphaZero self-play at the reference's per-GPU batch with rocprofv3 kernel stats
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/small; mkdir -p $out; rm -f $out/az_sweep.jsonl $out/ragged_mid.txt
python3 tests/tools/bench_c1.py 2>/dev/null | grep difficulty > $out/config1.jsonl
python3 scripts/geom_sweep.py 2>/dev/null | grep episodes > $out/geom_sweep.jsonl
python3 scripts/bench_az.py --envs 4096 --searches 100 --difficulty 8 --steps 3 2>/dev/null | grep metric > $out/az_4096x100.json
python3 scripts/bench_az.py --envs 4096 --searches 1000 --difficulty 8 --steps 1 2>/dev/null | grep metric > $out/az_4096x1000.json
for e in 16384 32768 49152 65536 262144; do python3 scripts/bench_az.py --envs $e --searches 32 --difficulty 4 --steps 2 2>/dev/null | grep metric >> $out/az_sweep.jsonl; done
for e in 16384 32768; do python3 scripts/ragged.py $e 2>/dev/null | grep records >> $out/ragged_mid.txt; done
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/az_stats -o s -f csv -- python3 $OLDPWD/scripts/bench_az.py --envs 4096 --searches 100 --difficulty 8 --steps 3 > $out/az_stats.log 2>&1)
ls $out $out/az_stats
