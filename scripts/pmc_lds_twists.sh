#!/bin/bash
# LDS bank conflicts of the exact-f32 rollout kernel with and without the twist (product build): one launch each, plus the
# bench line (kernel time) of both.  Run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/lds2; mkdir -p $out
for tw in "" "--no-twists"; do
  tag=${tw:+notwist}; tag=${tag:-twist}
  (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $out/$tag -o p -f csv -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline $tw > $out/$tag.log 2>&1) || echo "rocprofv3 failed"
  python3 - $out/$tag/p_counter_collection.csv $tag <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "rollout_f32_kernel" in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); meta = {k: r[k] for k in ("Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
print(sys.argv[2], meta, {k: f"{v:.4g}" for k, v in sorted(tot.items())}, "conflict/idx_active = %.3f" % (tot["SQ_LDS_BANK_CONFLICT"] / max(1.0, tot["SQ_LDS_IDX_ACTIVE"])))
PY
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $tw 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  bench', '$tag', 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'value', d['value'])"
done
