#!/usr/bin/env python3
"""The reference's four default evaluations (src/twisterl/defaults.py:27-57: 100 episodes each -- greedy, sampled, best of 10 sampled,
greedy over 100 MCTS searches per move) on Puzzle-15 with the benchmark-size policy: ms per call.  GPU box:  python scripts/bench_evaluate.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_policy, synthetic_weights
from twisterl_amd import twisterl
policy = build_policy(synthetic_weights(16, seed=0), [], [])
env = twisterl.env.Puzzle(4, 4, 8, 2, 256)
for name, kw in (("ppo_deterministic", dict(deterministic=True, num_searches=1, num_mcts_searches=0)), ("ppo_1", dict(deterministic=False, num_searches=1, num_mcts_searches=0)),
                 ("ppo_10", dict(deterministic=False, num_searches=10, num_mcts_searches=0)), ("mcts_100", dict(deterministic=True, num_searches=1, num_mcts_searches=100))):
    args = dict(num_episodes=100, seed=0, C=1.41, max_expand_depth=1, num_cores=32, **kw)
    twisterl.collector.evaluate(env, policy, **args)
    ts = []
    for i in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = twisterl.collector.evaluate(env, policy, **args)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(json.dumps({"evaluation": name, "episodes": 100, "ms": round(min(ts) * 1e3, 3), "success_rate": r[0], "mean_reward": r[1]}))
