#!/usr/bin/env python3
"""Rollout throughput (exact f32 mode, Puzzle-15, untrained policy) against the number of episodes, for the two launch
geometries: the throughput shape (8 waves x 32 episodes per workgroup) and the small-batch shape (4 waves sharing 32 episodes).
Diagnostic; pins the shape with tw_set_launch_option(TW_OPT_FORCE_GEOM).  Run on the GPU box:  python scripts/geom_sweep.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import time, torch
    from bench import build_policy, synthetic_weights
    from twisterl_amd import _lib, twisterl
    n = int(sys.argv[1])
    _lib.check(_lib.lib().tw_set_launch_option(_lib.TW_OPT_FORCE_GEOM, int(os.environ.get("TW_FORCE_GEOM", "0"))))
    pol = build_policy(synthetic_weights(16, seed=0), [], [])
    env = twisterl.env.Puzzle(4, 4, 8, 2, 64)
    coll = twisterl.collector.PPOCollector(n, 0.995, 0.995, 32)
    coll.collect(env, pol, seed=1)
    ms, rec = [], 0
    for i in range(3):
        d = coll.collect(env, pol, seed=10 + i); rec = len(d); ms.append(d.stats["ms_rollout"])
    print(json.dumps({"episodes": n, "geom": os.environ.get("TW_FORCE_GEOM", "auto"), "records": rec, "rollout_ms": min(ms),
                      "records_per_s": rec / (min(ms) * 1e-3)}))
else:
    for n in (1024, 4096, 8192, 16384, 32768, 49152, 65536):
        for g in ("8", "1"):
            env = dict(os.environ, TW_FORCE_GEOM=g)
            subprocess.run([sys.executable, __file__, str(n)], env=env, check=True)
