# one mid-size PPO rollout batch (for rocprofv3 --pmc): python scripts/mid_one.py [envs] [repeats]
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import build_policy, synthetic_weights, transpose_twist
from twisterl_amd import twisterl
envs = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
obs_perms, act_perms = transpose_twist(4)
policy = build_policy(synthetic_weights(16, seed=0), obs_perms, act_perms)
env = twisterl.env.Puzzle(4, 4, 32, 2, 256)
coll = twisterl.collector.PPOCollector(envs, 0.995, 0.995, 32)
ms = []
for i in range(reps):
    d = coll.collect(env, policy, seed=7); ms.append(d.stats["ms_rollout"])
print(json.dumps({"envs": envs, "records": len(d), "rollout_ms": min(ms), "threads": d.stats["rollout_threads"], "blocks": d.stats["rollout_blocks"]}))
