#!/bin/bash
# bench.py with the product library and with a measurement variant (TW_VARIANT=<name>, twisterl_amd/build.py), same box, back to back:
# headline and side entries as "key value ms kernel_ms".  Usage (GPU box, repo root): scripts/bench_pair.sh <variant> [steps]
V=${1:?variant name}; K=${2:-10}
summ() { python3 - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline", round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["kernel_ms"], 3))
for k, v in d.items():
    if isinstance(v, dict) and "value" in v and k != "cpu_baseline":
        print(k, round(v["value"]), round(v.get("ms_per_step", 0), 3), round(v.get("kernel_ms", 0), 3))
PY
}
for round in 1 2; do
  python3 bench.py --steps $K --warmup 3 --no-cpu-baseline > gpurun_out/pair_product_$round.json 2>/dev/null && echo "== product (run $round)" && summ gpurun_out/pair_product_$round.json
  TW_VARIANT=$V python3 bench.py --steps $K --warmup 3 --no-cpu-baseline > gpurun_out/pair_${V}_$round.json 2>/dev/null && echo "== variant $V (run $round)" && summ gpurun_out/pair_${V}_$round.json
done
