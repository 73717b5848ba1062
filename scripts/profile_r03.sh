#!/bin/bash
# round-3 measurement set (run on the GPU box from the repo root): default bench line (with the side entries of BASELINE configs
# 1, 2 and 5), rocprofv3 kernel stats of the same command, self-play bench lines + kernel stats + SQ counters of the walker kernel
# at 4,096 x 100 and x 1,000, the generic engine, the 5 x 5 board on the device, GPU tests, smoke.
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/r03; mkdir -p $out
python3 bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/stats -o s -f csv -- python3 $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1)
for cfg in "256 100 8" "512 100 8" "512 1000 8" "1024 100 8" "1024 1000 8" "2048 100 8" "4096 100 8" "4096 1000 8" "6144 100 8" "16384 100 8" "16384 32 4" "65536 32 4" "262144 32 4"; do set -- $cfg
  python3 scripts/bench_az.py --envs $1 --searches $2 --difficulty $3 --steps 2 2>/dev/null | grep metric >> $out/az_batches.jsonl
done
for S in 100 1000; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $out/az_stats_$S -o s -f csv -- python3 $OLDPWD/scripts/bench_az.py --envs 4096 --searches $S --steps 3 > $out/az_stats_$S.log 2>&1)
  (cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $out/az_pmc_$S -o p -f csv -- python3 $OLDPWD/scripts/bench_az.py --envs 4096 --searches $S --steps 1 > $out/az_pmc_$S.log 2>&1) || true
done
python3 scripts/bench_generic_engine.py 2>/dev/null | grep policy > $out/generic_engine.jsonl
python3 scripts/bench_big_board.py 2>/dev/null | grep board > $out/big_board.jsonl
python3 -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1
tail -2 $out/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ls $out
