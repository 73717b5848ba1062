#!/usr/bin/env python3
"""AlphaZero self-play bench (BASELINE config 5, single GPU): MCTS leaf evaluations/s and records/s.

    python scripts/bench_az.py [--envs 4096] [--searches 100] [--difficulty 8]

Not the headline metric (bench.py is); numbers go to BASELINE.md row C5.
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_policy, synthetic_weights

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--searches", type=int, default=100)
ap.add_argument("--difficulty", type=int, default=8)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--geom", type=int, default=0, help="diagnostic: tw_set_launch_option(TW_OPT_FORCE_GEOM)")
ap.add_argument("--variant", type=int, default=0, help="diagnostic: tw_set_launch_option(TW_OPT_AZ_VARIANT)")
ap.add_argument("--tree-budget", type=int, default=0, help="diagnostic: tw_set_launch_option(TW_OPT_AZ_TREE_BUDGET)")
ap.add_argument("--tree-budget-min", type=int, default=0, help="diagnostic: tw_set_launch_option(TW_OPT_AZ_TREE_BUDGET_MIN)")
args = ap.parse_args()

import twisterl_amd
from twisterl_amd import _lib, twisterl
_lib.check(_lib.lib().tw_set_launch_option(_lib.TW_OPT_FORCE_GEOM, args.geom))
_lib.check(_lib.lib().tw_set_launch_option(_lib.TW_OPT_AZ_VARIANT, args.variant))
_lib.check(_lib.lib().tw_set_launch_option(_lib.TW_OPT_AZ_TREE_BUDGET, args.tree_budget))
_lib.check(_lib.lib().tw_set_launch_option(_lib.TW_OPT_AZ_TREE_BUDGET_MIN, args.tree_budget_min))
policy = build_policy(synthetic_weights(16, seed=0), [], [])      # AZ clears the twists (rl/az.py:24-26)
env = twisterl.env.Puzzle(4, 4, args.difficulty, 2, 256)
coll = twisterl.collector.AZCollector(args.envs, args.searches, 1.41, 1, 32)
coll.collect(env, policy, seed=1)
t0 = time.perf_counter(); rec = ev = 0; ms = []
for i in range(args.steps):
    d = coll.collect(env, policy, seed=100 + i)
    rec += len(d); ev += d.stats["forward_evals"]; ms.append(d.stats["ms_rollout"])
dt = time.perf_counter() - t0
print(json.dumps({"metric": "MCTS leaf evaluations/s (Puzzle-15 AlphaZero self-play, 1 GPU)", "value": ev / dt,
                  "records_per_s": rec / dt, "episodes": args.envs, "num_mcts_searches": args.searches,
                  "difficulty": args.difficulty, "mcts_kernel_ms": sum(ms) / len(ms), "speculative_evals": d.stats["speculative_evals"], "reused_evals": d.stats["reused_evals"],
                  "threads": d.stats["rollout_threads"], "blocks": d.stats["rollout_blocks"],
                  "mfma_tflops": ev / len(ms) * 272896 / (sum(ms) / len(ms) * 1e-3) / 1e12}))
