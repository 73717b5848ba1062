#!/bin/bash
# SQ counters of the 16-column rollout kernel (Engine3T): instruction mix, LDS, waits.  Run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/smallpmc; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace -d $out/s$i -o p -f csv -- python3 $OLDPWD/scripts/bench_small_rollout.py > $out/s$i.log 2>&1) || echo "set $i: rocprofv3 failed"
  f=$(ls $out/s$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "rollout_f32_kernel" in k and "-16" in k and r["Grid_Size"] == str(256 * 256):
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for key, d in tot.items():
    print(key[:60], {k: f"{v:.4g}" for k, v in sorted(d.items())})
PY
done
