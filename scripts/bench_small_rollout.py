#!/usr/bin/env python3
"""Small-batch PPO rollouts (the reference's own 1,024-env case, and up to one 16-episode workgroup per CU): the 16-column
forward on 4 waves (Engine3T, automatic) against the 32-column shape (Engine3S, TW_OPT_FORCE_GEOM 32); checks that both give
the same bytes.  Run on the GPU box: python scripts/bench_small_rollout.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import build_policy, synthetic_weights, transpose_twist
from twisterl_amd import _lib, twisterl

obs_perms, act_perms = transpose_twist(4)
policy = build_policy(synthetic_weights(16, seed=0), obs_perms, act_perms)
env = twisterl.env.Puzzle(4, 4, 32, 2, 256)
for envs in (256, 1024, 4096):
    ref = None
    for geom in (32, 0):
        with _lib.launch_option(_lib.TW_OPT_FORCE_GEOM, geom):
            coll = twisterl.collector.PPOCollector(envs, 0.995, 0.995, 32)
            coll.collect(env, policy, seed=1)
            ms = []
            for i in range(5):
                d = coll.collect(env, policy, seed=7); ms.append(d.stats["ms_rollout"])
            h = d.to_numpy()
        same = None
        if ref is None: ref = h
        else: same = all(np.array_equal(ref[k].view(np.uint8), h[k].view(np.uint8)) for k in ref)
        print(json.dumps({"envs": envs, "geom": geom, "records": len(d), "rollout_ms": min(ms), "records_per_s": len(d) / (min(ms) * 1e-3),
                          "threads": d.stats["rollout_threads"], "blocks": d.stats["rollout_blocks"], "same_bytes_as_geom32": same}))
