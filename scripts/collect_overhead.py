import sys, time
sys.path.insert(0, "/root/repo")
import bench, torch
from twisterl_amd import twisterl
pol = bench.build_policy(bench.synthetic_weights(9), [], [])
for D in (1, 8, 32):
    env = twisterl.env.Puzzle(3, 3, D, 2, 256)
    coll = twisterl.collector.PPOCollector(1024, 0.995, 0.995, 32)
    for i in range(5): coll.collect(env, pol, seed=i)
    ts = []
    for i in range(50):
        t0 = time.perf_counter(); d = coll.collect(env, pol, seed=100 + i); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(D, "median total us %.0f" % (ts[25] * 1e6), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.stats.items()})
