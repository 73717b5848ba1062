#!/bin/bash
# instruction mix of the walker self-play kernel on a batch of lone walkers (64 x 1,000): instructions per search.  GPU box, repo root.
out=$PWD/gpurun_out/pmc_az_insts; mkdir -p $out
export TMPDIR=/tmp
E=${1:-64}; S=${2:-1000}
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-50)
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace -d $out/$n -o p -f csv -- python3 $OLDPWD/scripts/bench_az.py --envs $E --searches $S --steps 1 > $out/$n.log 2>&1) || echo "pass failed: $set"
done
python3 - $out <<'PY'
import csv, glob, sys, collections, json
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mcts_deep" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(json.dumps({k: int(v) for k, v in sorted(tot.items())}))
print("launches per counter:", dict(n))
PY
grep -h metric $out/*.log | tail -1
