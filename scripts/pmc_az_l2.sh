#!/bin/bash
# L2 hit / miss counters of the walker kernel for two launch shapes (TW_OPT_AZ_VARIANT values).  GPU box, repo root.
export TMPDIR=/tmp
out=$PWD/gpurun_out/azl2; rm -rf $out; mkdir -p $out
for v in ${AZ_VARIANTS:-37 38}; do
  (cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $out/v$v -o p -f csv -- python3 $OLDPWD/scripts/bench_az.py --envs 4096 --searches 100 --steps 1 --variant $v > $out/v$v.log 2>&1) || true
  python3 - <<PY
import csv, collections
tot=collections.Counter()
for r in csv.DictReader(open("$out/v$v/p_counter_collection.csv")):
    if 'mcts_deep_kernel' in r['Kernel_Name']: tot[r['Counter_Name']]+=float(r['Counter_Value'])
print("variant $v", dict(tot))
PY
done
