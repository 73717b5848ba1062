#!/usr/bin/env python3
"""Throughput of the generic vector-ALU engine (tw_engine_generic.hpp) against the MFMA engine on (almost) the same network:
Puzzle-15 rollout, 65,536 envs, difficulty 32; embedding 512 (MFMA shape) vs 508 (not a multiple of 32 -> generic engine)
and a deeper stack.  Diagnostic; run on the GPU box:  python scripts/bench_generic_engine.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests.util import amd_policy, make_deep_policy_arrays
from twisterl_amd import twisterl

env = twisterl.env.Puzzle(4, 4, 32, 2, 256)
for name, kw in (("mfma 256->512->256", dict(emb=512, common=(256,))), ("generic 256->508->256", dict(emb=508, common=(256,))),
                 ("generic 256->512->256->256, heads 64/64", dict(emb=512, common=(256, 256), policy_layers=(64,), value_layers=(64,)))):
    pol = amd_policy(make_deep_policy_arrays(16, seed=0, **kw))
    coll = twisterl.collector.PPOCollector(65536, 0.995, 0.995, 32)
    coll.collect(env, pol, seed=1)
    ms, n = [], 0
    for i in range(3):
        d = coll.collect(env, pol, seed=2 + i); n = len(d); ms.append(d.stats["ms_rollout"])
    print(json.dumps({"policy": name, "records": n, "rollout_ms": min(ms), "records_per_s": n / (min(ms) * 1e-3),
                      "threads": d.stats["rollout_threads"], "blocks": d.stats["rollout_blocks"]}))
