#!/bin/bash
# round-4 measurement set (GPU box, repo root): default bench line with its side entries, rocprofv3 kernel stats of the same command,
# HBM traffic passes (FETCH_SIZE / WRITE_SIZE, separate) of the three rollout modes, self-play batches + kernel stats + SQ counters of the
# walker kernel at 4,096 x 100 and x 1,000.  Summaries: scripts/summarize_traffic.py, scripts/summarize_az_pmc.py.
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/r04; mkdir -p $out
python3 bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/stats -o s -f csv -- python3 $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1)
echo "[profile_r04] bench + kernel stats done"
mkdir -p $out/traffic
for p in fp32 fp16x2 fp16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $out/traffic/${p}_$c -o p -f csv -- python3 $OLDPWD/bench.py --precision $p --steps 1 --warmup 0 --no-cpu-baseline > $out/traffic/${p}_$c.log 2>&1)
  done
  echo "[profile_r04] traffic passes of $p done"
done
for cfg in "256 100 8" "512 100 8" "512 1000 8" "1024 100 8" "1024 1000 8" "2048 100 8" "2048 1000 8" "4096 100 8" "4096 400 8" "4096 1000 8" "6144 100 8" "16384 100 8" "16384 32 4" "65536 32 4"; do set -- $cfg
  python3 scripts/bench_az.py --envs $1 --searches $2 --difficulty $3 --steps 2 2>/dev/null | grep metric >> $out/az_batches.jsonl
done
echo "[profile_r04] self-play batches done"
for S in 100 1000; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $out/az_stats_$S -o s -f csv -- python3 $OLDPWD/scripts/bench_az.py --envs 4096 --searches $S --steps 3 > $out/az_stats_$S.log 2>&1)
  # (counter collection serialises kernels, and the split shape's two kernels need each other: the counters are those of the decoupled shape inside one workgroup, variant 1024)
  (cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $out/az_pmc_$S -o p -f csv -- python3 $OLDPWD/scripts/bench_az.py --envs 4096 --searches $S --steps 1 --variant 1024 > $out/az_pmc_$S.log 2>&1) || true
  echo "[profile_r04] self-play counters x $S done"
done
python3 scripts/bench_generic_engine.py 2>/dev/null | grep policy > $out/generic_engine.jsonl || true
ls $out
