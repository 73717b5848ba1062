#!/usr/bin/env python3
"""ms of the scan and of the GAE + compaction kernel (tw_finalize.hip) at the headline size: 262,144 Puzzle-15 episodes, <= 257 records each.
   python scripts/bench_finalize.py [collects]"""
import sys
sys.path.insert(0, ".")
from bench import build_policy, synthetic_weights, transpose_twist
from twisterl_amd import twisterl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pol = build_policy(synthetic_weights(16, seed=0), *transpose_twist(4))
env = twisterl.env.Puzzle(4, 4, 128, 2, 256)
c = twisterl.collector.PPOCollector(**{"num_episodes": 262144, "gamma": 0.995, "lambda": 0.995, "num_cores": 32})
for i in range(n):
    d = c.collect(env, pol, seed=10 + i)
    st = d.stats
    print(f"collect {i}: records {len(d)}, rollout {st['ms_rollout']:.2f} ms, scan {st['ms_scan']:.3f} ms, finalize {st['ms_finalize']:.3f} ms", flush=True)
# checksums of ragged / odd-sized collects (compare across builds or forms of the kernel)
import zlib, numpy as np
for side, E, diff in ((4, 4099, 6), (3, 1001, 5), (4, 37, 128), (3, 5, 1)):
    p = build_policy(synthetic_weights(side * side, seed=1), *transpose_twist(side))
    ev = twisterl.env.Puzzle(side, side, diff, 2, 64 if side == 3 else 256)
    cc = twisterl.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.99, "lambda": 0.95, "num_cores": 32})
    a = cc.collect(ev, p, seed=3).to_numpy()
    print(f"check {side}x{side} E={E} d={diff}: records {len(next(iter(a.values())))} " + " ".join(f"{k}={zlib.crc32(np.ascontiguousarray(a[k]).tobytes()):08x}" for k in sorted(a)), flush=True)
