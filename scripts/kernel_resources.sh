#!/bin/bash
# Register / scratch / LDS use of every kernel of one translation unit (compile-time remark; no GPU needed):
#   scripts/kernel_resources.sh tw_mcts_deep.hip [name filter]
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=400000 \
  -fno-gpu-rdc -I include -I twisterl_amd/csrc $TW_EXTRA_FLAGS -c "twisterl_amd/csrc/$1" -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = {}
for ln in sys.stdin:
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (.*?) \[-Rpass", ln)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
    cur[k] = v
    if k.startswith("LDS") and flt in cur.get("name", ""):
        print(cur["name"], "| VGPRs", cur.get("VGPRs"), "AGPRs", cur.get("AGPRs"), "SGPRs", cur.get("TotalSGPRs"), "scratch", cur.get("ScratchSize [bytes/lane]"), "SGPR spills", cur.get("SGPRs Spill"), "VGPR spills", cur.get("VGPRs Spill"), "occupancy", cur.get("Occupancy [waves/SIMD]"))
' "$2"
rm -f /tmp/kres_$$.o
