#!/bin/bash
# SQ counters of the exact-f32 rollout kernel (one launch): matrix-pipe busy cycles, co-execution, waits, LDS conflicts
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc32; mkdir -p $out
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $out/a -o p -f csv -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/a.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace -d $out/b -o p -f csv -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/b.log 2>&1) || true
ls $out/a $out/b 2>/dev/null
