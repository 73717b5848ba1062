#!/bin/bash
# SQ counters of the generic (vector-ALU) engine's rollout kernel: where its cycles go.  Run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/genpmc; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace -d $out/s$i -o p -f csv -- python3 $OLDPWD/scripts/bench_generic_engine.py > $out/s$i.log 2>&1) || echo "set $i: rocprofv3 failed"
  f=$(ls $out/s$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "rollout_f32_kernel" in k:
        key = "generic" if "Lin64E" in k else "mfma"
        tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
for key, d in tot.items():
    print(key, {k: f"{v:.4g}" for k, v in sorted(d.items())})
PY
done
