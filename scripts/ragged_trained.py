"""Ragged PPO rollouts with the reference's trained Puzzle-8 policy (episodes end when solved): rollout kernel ms."""
import sys, json
sys.path.insert(0, ".")
from tests.util import amd_policy, trained_puzzle8_arrays
from twisterl_amd import twisterl
gp = amd_policy(trained_puzzle8_arrays())
for E, diff in ((262144, 32), (1048576, 32), (262144, 12)):
    env = twisterl.env.Puzzle(3, 3, diff, 2, 256)
    coll = twisterl.collector.PPOCollector(E, 0.995, 0.995, 32)
    coll.collect(env, gp, seed=1)
    ms = []
    for i in range(3):
        d = coll.collect(env, gp, seed=2 + i); ms.append(d.stats["ms_rollout"]); n = len(d); del d
    print(json.dumps({"envs": E, "difficulty": diff, "records": n, "mean_len": n / E, "rollout_ms": min(ms)}))
