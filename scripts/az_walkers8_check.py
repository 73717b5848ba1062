#!/usr/bin/env python3
"""Eight walkers per workgroup (TW_OPT_AZ_VARIANT 32 + 6) against the automatic shape: same bytes?  GPU box, repo root."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import build_policy, synthetic_weights, transpose_twist
from twisterl_amd import twisterl, _lib

for side, E, S, twists in ((3, 64, 6, False), (4, 300, 20, True), (4, 2500, 10, False)):
    obs_perms, act_perms = transpose_twist(side) if twists else ([], [])
    policy = build_policy(synthetic_weights(side * side, seed=0), obs_perms, act_perms)
    env = twisterl.env.Puzzle(side, side, 6, 2, 256)
    auto = twisterl.collector.AZCollector(E, S, 1.41, 1, 1).collect(env, policy, seed=5)
    with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 32 + 6):
        d = twisterl.collector.AZCollector(E, S, 1.41, 1, 1).collect(env, policy, seed=5)
    a, b = auto.to_numpy(), d.to_numpy()
    ok = all(np.array_equal(a[k], b[k]) for k in a)
    print(side, E, S, twists, "threads", d.stats["rollout_threads"], "blocks", d.stats["rollout_blocks"], "same bytes" if ok else "MISMATCH", flush=True)
