"""Round 3, VERDICT item 1: how the lane-per-episode self-play kernel finds the grandparent whose stored network output a node
that takes its parent's move back reuses.  For every case: the collect under TW_OPT_AZ_REUSE = 1 (no reuse: the baseline
bytes), 3 (parent links), 0 (product: search path in LDS, parent links below PATH_DEPTH levels), 2 (round 2's first form: the path
level whatever the depth), each repeated; and once under 4, which counts how the two ways disagree.
    python scripts/az_reuse_probe.py [repeats]"""
import sys
sys.path.insert(0, ".")
import torch
import bench
from tests.util import puzzle_transpose_twist
from twisterl_amd import twisterl, _lib
from twisterl_amd._lib import launch_option, TW_OPT_AZ_REUSE, TW_OPT_AZ_VARIANT, TW_OPT_FORCE_GEOM, TW_OPT_NO_PERSIST

def digest(d):
    t = d.to_torch()
    out = []
    for k in sorted(t):
        x = t[k]
        v = x.view(torch.uint8).to(torch.int64) if x.dtype in (torch.uint8, torch.int8) else x.contiguous().view(torch.int32).to(torch.int64)
        w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 1000003
        out.append(int((v.reshape(-1) * w).sum().item()))
    return tuple(out)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
op, ap = puzzle_transpose_twist(4)
pol_t = bench.build_policy(bench.synthetic_weights(16), op, ap)
pol_0 = bench.build_policy(bench.synthetic_weights(16), [], [])
NAMES = ["evals", "spec", "reused", "undo candidates", "of them below PATH_DEPTH", "plen<3", "path!=links (path fits)", "path!=links (below PATH_DEPTH)",
         "links: board/expansion check fails", "path: board/expansion check fails", "path tail != node", "path[-2] != parent", "tripwire"]
cases = [  # E, searches, max_expand_depth, difficulty, policy, force_geom
    (70, 24, 1, 4, pol_0, 0), (70, 100, 1, 8, pol_t, 0), (70, 400, 1, 8, pol_0, 0), (70, 400, 2, 8, pol_0, 0), (64, 2000, 1, 8, pol_0, 0),
    (2000, 400, 1, 8, pol_0, 8), (20000, 100, 1, 8, pol_0, 0), (65536, 32, 1, 8, pol_t, 0), (16384, 100, 1, 8, pol_t, 0), (70000, 32, 2, 8, pol_0, 0),
]
for E, S, MED, D, pol, fg in cases:
    env = twisterl.env.Puzzle(4, 4, D, 2, 256)
    coll = twisterl.collector.AZCollector(E, S, 1.41, MED, 1)
    res = {}
    with launch_option(TW_OPT_AZ_VARIANT, 2), launch_option(TW_OPT_FORCE_GEOM, fg):
        for mode in (1, 3, 0, 2):
            with launch_option(TW_OPT_AZ_REUSE, mode):
                ds = []
                for r in range(reps if mode != 1 else 1):
                    d = coll.collect(env, pol, seed=7)
                    ds.append(digest(d))
                res[mode] = (ds, d.stats["ms_rollout"], d.stats["reused_evals"], d.stats["forward_evals"], d.stats["rollout_blocks"], d.stats["rollout_threads"])
        with launch_option(TW_OPT_AZ_REUSE, 4):
            d4 = coll.collect(env, pol, seed=7)
            cnt = _lib.debug_counters(13)
            dg4 = digest(d4)
    base = res[1][0][0]
    print(f"E {E} searches {S} MED {MED} twists {pol is pol_t} launch {res[1][4]}x{res[1][5]}")
    for mode, name in ((1, "no reuse"), (3, "parent links"), (0, "product"), (2, "path whatever the depth")):
        ds, ms, reused, fwd, _, _ = res[mode]
        print(f"   mode {mode} ({name}): {ms:8.2f} ms  reused {reused}/{fwd}  same bytes as no-reuse: {[x == base for x in ds]}  repeats identical: {len(set(ds)) == 1}")
    print(f"   mode 4 same bytes as no-reuse: {dg4 == base}; " + "; ".join(f"{n} {c}" for n, c in zip(NAMES, cnt)), flush=True)
