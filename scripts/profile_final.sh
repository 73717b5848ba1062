#!/bin/bash
# round-1 final measurement set (run on the GPU box from the repo root): default bench line, rocprofv3 kernel stats of the
# same command, split-f16 mode stats
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/final; mkdir -p $out
python3 bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/stats -o s -f csv -- python3 $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1)
python3 bench.py --precision fp16x2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_f16x2.json 2>> $out/bench.err
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $out/pmc_f16x2 -o p -f csv -- python3 $OLDPWD/bench.py --precision fp16x2 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_f16x2.log 2>&1)
python3 -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1
tail -2 $out/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ls $out $out/stats
