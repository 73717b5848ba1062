#!/bin/bash
# round-1 measurement set for the f16-input mode (run on the GPU box from the repo root)
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/f16prof; mkdir -p $out
python3 bench.py --precision fp16 --steps 5 --warmup 2 > $out/bench_f16_p15.json 2> $out/bench_f16_p15.err
python3 bench.py --precision fp16 --puzzle 8 --envs 65536 --difficulty 32 --no-twists --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_f16_p8_65k.json 2>> $out/bench_f16_p15.err
python3 bench.py --precision fp32 --puzzle 8 --envs 65536 --difficulty 32 --no-twists --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_f32_p8_65k.json 2>> $out/bench_f16_p15.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/stats -o s -f csv -- python3 $OLDPWD/bench.py --precision fp16 --steps 5 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1)
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $out/pmc -o p -f csv -- python3 $OLDPWD/bench.py --precision fp16 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc.log 2>&1)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o p -f csv -- python3 $OLDPWD/bench.py --precision fp16 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o p -f csv -- python3 $OLDPWD/bench.py --precision fp16 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_write.log 2>&1)
ls $out $out/stats
