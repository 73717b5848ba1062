#!/usr/bin/env python3
"""PPO rollout of 5 x 5, 6 x 6 and 8 x 8 boards (25 / 36 / 64 cells: tw_rollout_big.hip on the device) against the same collect through the host-stepped
path (tw_ppo_collect_env, pinned with TW_OPT_FORCE_GEOM), 65,536 envs.  Diagnostic; GPU box:  python scripts/bench_big_board.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import amd_policy, make_deep_policy_arrays
from twisterl_amd import twisterl, _lib

for w, D in ((5, 16), (6, 16), (8, 16)):
    n2 = w * w
    pol = amd_policy(make_deep_policy_arrays(n2, seed=0, emb=512, common=(256,)))
    env = twisterl.env.Puzzle(w, w, D, 2, 256)
    for name, E, geom in (("device (tw_rollout_big.hip)", 65536, 0), ("host-stepped (tw_ppo_collect_env)", 4096, 1)):
        with _lib.launch_option(_lib.TW_OPT_FORCE_GEOM, geom):
            coll = twisterl.collector.PPOCollector(E, 0.995, 0.995, 32)
            coll.collect(env, pol, seed=1)
            t0 = time.perf_counter(); d = coll.collect(env, pol, seed=2); dt = time.perf_counter() - t0
        print(json.dumps({"board": "%dx%d, policy %d->512->256->4|1, difficulty %d" % (w, w, n2 * n2, D), "path": name, "envs": E, "records": len(d), "wall_ms": dt * 1e3,
                          "rollout_ms": d.stats["ms_rollout"], "records_per_s": len(d) / dt, "blocks": d.stats["rollout_blocks"], "threads": d.stats["rollout_threads"]}))

# AlphaZero self-play of such boards: the device kernel (tw_mcts_big.hip) against the host-stepped collector
for w, E, S in ((5, 4096, 100), (8, 4096, 100)):
    n2 = w * w
    pol = amd_policy(make_deep_policy_arrays(n2, seed=0, emb=512, common=(256,)))
    env = twisterl.env.Puzzle(w, w, 8, 2, 256)
    for name, EE, geom in (("device (tw_mcts_big.hip)", E, 0), ("host-stepped (tw_az_collect_env)", E // 16, 1)):
        with _lib.launch_option(_lib.TW_OPT_FORCE_GEOM, geom):
            coll = twisterl.collector.AZCollector(EE, S, 1.41, 1, 32)
            coll.collect(env, pol, seed=1)
            t0 = time.perf_counter(); d = coll.collect(env, pol, seed=2); dt = time.perf_counter() - t0
        print(json.dumps({"board": "%dx%d self-play, policy %d->512->256->4|1, difficulty 8, %d searches" % (w, w, n2 * n2, S), "path": name, "episodes": EE, "records": len(d),
                          "wall_ms": dt * 1e3, "leaf_evaluations_per_s": d.stats["forward_evals"] / dt if d.stats["forward_evals"] else None,
                          "records_per_s": len(d) / dt, "blocks": d.stats["rollout_blocks"], "threads": d.stats["rollout_threads"]}))
