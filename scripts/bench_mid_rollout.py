import json, os, sys
sys.path.insert(0, ".")
import numpy as np
from bench import build_policy, synthetic_weights, transpose_twist
from twisterl_amd import _lib, twisterl
obs_perms, act_perms = transpose_twist(4)
policy = build_policy(synthetic_weights(16, seed=0), obs_perms, act_perms)
env = twisterl.env.Puzzle(4, 4, 32, 2, 256)
for envs in (8192, 16384, 32768, 65536):
    coll = twisterl.collector.PPOCollector(envs, 0.995, 0.995, 32)
    coll.collect(env, policy, seed=1)
    ms = []
    for i in range(5):
        d = coll.collect(env, policy, seed=7); ms.append(d.stats["ms_rollout"])
    print(json.dumps({"envs": envs, "records": len(d), "rollout_ms": min(ms), "records_per_s": len(d) / (min(ms) * 1e-3), "threads": d.stats["rollout_threads"], "blocks": d.stats["rollout_blocks"],
                      "mfma_frac": len(d) * 272896 / (min(ms) * 1e-3) / 157.3e12}))
