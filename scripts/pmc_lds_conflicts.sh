#!/bin/bash
# Where do the LDS bank conflicts of the exact-f32 rollout kernel come from?  SQ_LDS_* counters of one launch of the full
# kernel and of its timing-only ablations (TW_ABLATE build: TW_ROLLOUT_DBG 1 no gather, 2 no A-operand reads, 4 no weight
# streams (LDS-DMA), 8 no heads).  Run on the GPU box from the repo root; builds the ablation library first.
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/lds; mkdir -p $out
export TW_ABLATE=1   # the instrumented library lives in twisterl_amd/lib/ablate/ and is loaded only while this is set
python3 -m twisterl_amd.build > $out/build.log 2>&1
for dbg in 0 1 2 4 8; do
  (cd /tmp && TW_ROLLOUT_DBG=$dbg rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $out/d$dbg -o p -f csv -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --envs 65536 > $out/d$dbg.log 2>&1) || echo "dbg $dbg: rocprofv3 failed"
  f=$(ls $out/d$dbg/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" $dbg <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); dur = 0
for r in csv.DictReader(open(sys.argv[1])):
    if "rollout_f32_kernel" in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
print("dbg", sys.argv[2], {k: f"{v:.4g}" for k, v in sorted(tot.items())},
      "conflict/idx_active = %.3f" % (tot["SQ_LDS_BANK_CONFLICT"] / max(1.0, tot["SQ_LDS_IDX_ACTIVE"])))
PY
done
