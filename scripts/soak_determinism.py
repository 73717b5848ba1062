"""Soak: the same collect repeated must give bit-identical buffers (catches rare LDS-ring / queue races that a single
parity run can miss).  Hashes every field on the device."""
import sys
sys.path.insert(0, ".")
import torch
import bench
from tests.util import amd_policy, puzzle_transpose_twist
from twisterl_amd import twisterl

def digest(d):
    t = d.to_torch()
    out = []
    for k in sorted(t):
        x = t[k]
        if x.dtype in (torch.uint8, torch.int8): v = x.view(torch.uint8).to(torch.int64)
        elif x.element_size() == 2: v = x.contiguous().view(torch.int16).to(torch.int64)          # two-byte obs ids (boards above 16 cells)
        else: v = x.contiguous().view(torch.int32).to(torch.int64)
        w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 1000003
        out.append(int((v.reshape(-1) * w).sum().item()))
    return tuple(out)

op, ap = puzzle_transpose_twist(4)
pol = bench.build_policy(bench.synthetic_weights(16), op, ap)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for prec, env, E in (("fp32", twisterl.env.Puzzle(4, 4, 128, 2, 256), 262144), ("fp16x2", twisterl.env.Puzzle(4, 4, 128, 2, 256), 262144),
                     ("fp16", twisterl.env.Puzzle(4, 4, 128, 2, 256), 262144), ("fp32", twisterl.env.Puzzle(4, 4, 1, 64, 256), 200000),
                     ("fp16x2", twisterl.env.Puzzle(4, 4, 1, 64, 256), 200000)):
    coll = twisterl.collector.PPOCollector(E, 0.995, 0.995, 1, precision=prec)
    ref = None; bad = 0
    for i in range(n):
        dg = digest(coll.collect(env, pol, seed=5))
        if ref is None: ref = dg
        elif dg != ref: bad += 1
    print(prec, "E", E, "difficulty", env.difficulty, "repeats", n, "mismatching repeats", bad, flush=True)

# small and mid-size batches: the 16-episode shape, the 32-episode shape, the same with the episode queue; self-play likewise
pol0 = bench.build_policy(bench.synthetic_weights(16), [], [])
for E, rep in ((1024, 4 * n), (4096, 4 * n), (8000, 2 * n), (20000, 2 * n)):
    for env in (twisterl.env.Puzzle(4, 4, 16, 2, 256), twisterl.env.Puzzle(4, 4, 1, 32, 256)):
        coll = twisterl.collector.PPOCollector(E, 0.995, 0.995, 1)
        ref = None; bad = 0
        for i in range(rep):
            dg = digest(coll.collect(env, pol, seed=5))
            if ref is None: ref = dg
            elif dg != ref: bad += 1
        print("fp32 small E", E, "difficulty", env.difficulty, "repeats", rep, "mismatching repeats", bad, flush=True)
# (walker kernel: 256 episodes = one walker per workgroup, 800 / 1,024 two, 2,000 four on the 32-column engine, 4,096 and 6,000 x 100
#  eight in eight-wave workgroups, 512 x 1,000 lone walkers on long streaks of searches, 1,100 x 400 four walkers + the episode queue in
#  longest-first order; 20,000 and 70,000: lane-per-episode kernel)
for cfg in ((256, 60, n, pol0), (800, 40, n, pol), (1024, 50, n, pol0), (2000, 40, n, pol), (4096, 30, n, pol0), (4096, 100, n, pol),
                     (6000, 100, max(2, n // 2), pol0), (512, 1000, max(2, n // 4), pol0), (1100, 400, max(2, n // 4), pol0),
                     (20000, 16, n, pol0), (70000, 8, max(2, n // 3), pol0),
                     # round 4: from 2,048 episodes on these run the SPLIT shape (walker kernel + engine kernel, mailboxes in device memory; which
                     # engine serves which request when differs from run to run) -- and the same batches pinned to the decoupled shapes inside one workgroup
                     (4096, 1000, max(2, n // 4), pol0), (16384, 100, max(2, n // 3), pol0), (8192, 200, max(2, n // 3), pol), (2048, 400, max(2, n // 3), pol0),
                     (4096, 100, n, pol, 1024), (2000, 40, n, pol, 1024), (4096, 1000, max(2, n // 4), pol0, 1024), (1024, 1000, max(2, n // 4), pol0, 1024)):
    (E, S, rep, p), variant = cfg[:4], (cfg[4] if len(cfg) > 4 else 0)
    env = twisterl.env.Puzzle(4, 4, 4 if S < 100 else 8, 2, 256)
    coll = twisterl.collector.AZCollector(E, S, 1.41, 1, 1)
    ref = None; bad = 0
    from twisterl_amd import _lib as _L
    with _L.launch_option(_L.TW_OPT_AZ_VARIANT, variant):
        for i in range(rep):
            d = coll.collect(env, p, seed=7)
            dg = digest(d)
            if ref is None: ref = dg
            elif dg != ref: bad += 1
    print("self-play E", E, "searches", S, "twists" if p is pol else "no twists", "variant", variant, "launch", d.stats["rollout_blocks"], "x", d.stats["rollout_threads"],
          "repeats", rep, "mismatching repeats", bad, flush=True)
# lane-per-episode kernel with the reused outputs active (>= 32 searches; hidden 32 / 64 / 256; plain, persistent, every pinned
# geometry, deep trees whose search path outgrows the LDS levels)
from twisterl_amd import _lib
from tests.util import make_policy_arrays
small = {h: amd_policy(make_policy_arrays(16, seed=3, emb=64, hidden=h)) for h in (32, 64)}
for E, S, MED, p, fg, nop, rep in ((65536, 32, 1, pol, 0, 0, max(2, n // 3)), (16384, 100, 1, pol, 0, 0, max(2, n // 3)), (70000, 32, 2, pol0, 0, 0, max(2, n // 3)),
                                    (70000, 32, 1, pol0, 0, 1, max(2, n // 3)), (3000, 48, 1, small[32], 0, 0, n), (3000, 48, 1, small[64], 0, 0, n),
                                    (2000, 64, 1, pol0, 1, 0, n), (2000, 64, 1, pol0, 8, 0, n), (2000, 64, 1, pol, 32, 0, n), (70, 400, 2, pol0, 0, 0, n),
                                    (300, 2000, 1, small[64], 0, 0, max(2, n // 3))):
    env = twisterl.env.Puzzle(4, 4, 8, 2, 256)
    coll = twisterl.collector.AZCollector(E, S, 1.41, MED, 1)
    ref = None; bad = 0
    with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 2), _lib.launch_option(_lib.TW_OPT_FORCE_GEOM, fg), _lib.launch_option(_lib.TW_OPT_NO_PERSIST, nop):
        for i in range(rep):
            d = coll.collect(env, p, seed=7)
            dg = digest(d)
            if ref is None: ref = dg
            elif dg != ref: bad += 1
    print("self-play, lane per episode: E", E, "searches", S, "max_expand_depth", MED, "hidden", p.common.layers[0].out_features,
          "force_geom", fg, "no_persist", nop, "launch", d.stats["rollout_blocks"], "x", d.stats["rollout_threads"], "reused", d.stats["reused_evals"], "of", d.stats["forward_evals"],
          "repeats", rep, "mismatching repeats", bad, flush=True)
# policies of any depth (EngineV: inline-asm MFMA chains, weights and activations prefetched through running pointers)
from tests.util import make_deep_policy_arrays
for kw, E, rep in ((dict(emb=508, common=(256,)), 65536, n), (dict(emb=512, common=(256, 256), policy_layers=(64,), value_layers=(64,)), 20000, n),
                   (dict(emb=96, common=(96, 32)), 4096, 4 * n)):
    polg = amd_policy(make_deep_policy_arrays(16, seed=0, **kw))
    env = twisterl.env.Puzzle(4, 4, 16, 2, 256)
    coll = twisterl.collector.PPOCollector(E, 0.995, 0.995, 1)
    ref = None; bad = 0
    for i in range(rep):
        dg = digest(coll.collect(env, polg, seed=5))
        if ref is None: ref = dg
        elif dg != ref: bad += 1
    print("generic policy", kw, "E", E, "repeats", rep, "mismatching repeats", bad, flush=True)
# boards above 16 cells on the device (tw_rollout_big.hip, tw_mcts_big.hip) and self-play with a generic policy on a 4 x 4 board
for (w, h), kw, E, S, rep in (((5, 5), dict(emb=64, common=(128,)), 20000, 0, n), ((6, 4), dict(emb=32, common=(48, 32)), 5000, 0, n), ((8, 8), dict(emb=64, common=(64,)), 8192, 0, n),
                              ((6, 4), dict(emb=32, common=(48, 32)), 600, 24, n), ((6, 6), dict(emb=32, common=(64, 32)), 400, 16, n), ((8, 8), dict(emb=32, common=(32,)), 300, 12, n),
                              ((4, 4), dict(emb=64, common=(96, 32)), 2000, 24, n)):
    polb = amd_policy(make_deep_policy_arrays(w * h, seed=1, scale=2.0, **kw))
    env = twisterl.env.Puzzle(w, h, 6, 2, 256)
    coll = twisterl.collector.AZCollector(E, S, 1.41, 1, 1) if S else twisterl.collector.PPOCollector(E, 0.995, 0.995, 1)
    ref = None; bad = 0
    for i in range(rep):
        d = coll.collect(env, polb, seed=5)
        dg = digest(d)
        if ref is None: ref = dg
        elif dg != ref: bad += 1
    print("board", w, "x", h, "self-play %d searches" % S if S else "PPO", kw, "E", E, "launch", d.stats["rollout_blocks"], "x", d.stats["rollout_threads"],
          "repeats", rep, "mismatching repeats", bad, flush=True)
