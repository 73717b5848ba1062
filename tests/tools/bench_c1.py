"""BASELINE config 1: Puzzle-8 PPO rollout, 1,024 envs, the CPU restatement of the reference collector
on the host cores (kind "port": the Rust reference cannot be built here), D in {1, 8, 32}; next to it the HIP path on
the same tiny batch (plumbing-sized: 32 one-wave workgroups).  Median of 10 timed calls after 3 warm-ups."""
import json, os, sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
from oracle import oracle as O
from twisterl_amd import twisterl

threads = min(16, os.cpu_count() or 1)
arrs = bench.synthetic_weights(9)
pol_cpu = O.Policy(*arrs)
pol_gpu = bench.build_policy(arrs, [], [])
rows = []
for D in (1, 8, 32):
    env_c, env_g = O.Puzzle(3, 3, D, 2, 256), twisterl.env.Puzzle(3, 3, D, 2, 256)
    def cpu(i):
        return len(O.ppo_collect(env_c, pol_cpu, 1024, 0.995, 0.995, seed=i, arith=O.ARITH_REF, num_threads=threads).values)
    coll = twisterl.collector.PPOCollector(1024, 0.995, 0.995, 32)
    def gpu(i):
        return len(coll.collect(env_g, pol_gpu, seed=i))
    res = {}
    for name, f in (("cpu", cpu), ("gpu", gpu)):
        for i in range(3):
            f(i)
        ts, n = [], 0
        for i in range(10):
            t0 = time.perf_counter(); n = f(100 + i); ts.append(time.perf_counter() - t0)
        res[name] = {"records": n, "median_ms": float(np.median(ts)) * 1e3, "records_per_s": n / float(np.median(ts))}
    rows.append({"difficulty": D, "threads": threads, **res})
    print(json.dumps(rows[-1]), flush=True)
