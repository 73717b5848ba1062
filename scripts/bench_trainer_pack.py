"""Measure the trainer hand-off kernels (tw_trainer.hip) on a Puzzle-15 collect: one-hot + log-prob pack of
`--rows` records, reported in GB/s of HBM written against the 8 TB/s roofline."""
import argparse, json, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from twisterl_amd import twisterl, trainer

ap = argparse.ArgumentParser(); ap.add_argument("--envs", type=int, default=65536); ap.add_argument("--rows", type=int, default=8_000_000)
args = ap.parse_args()
from tests.util import puzzle_transpose_twist
op, ap_ = puzzle_transpose_twist(4)
pol = bench.build_policy(bench.synthetic_weights(16), op, ap_)
env = twisterl.env.Puzzle(4, 4, 128, 2, 256)
data = twisterl.collector.PPOCollector(args.envs, 0.995, 0.995, 1).collect(env, pol)
n = min(args.rows, len(data))
out = trainer.ppo_data_to_torch(data, 256, True, rows=(0, n))
reps, ts = 5, []
for _ in range(reps):
    out = None                                   # (the previous result goes back to torch's allocator first: one result resident at a time)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = trainer.ppo_data_to_torch(data, 256, True, rows=(0, n))
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
dt = sorted(ts)[len(ts) // 2]                    # median
bytes_written = n * (256 * 4 + 4 + 8 + 8 + 4 + 4)
print(json.dumps({"rows": n, "ms": dt * 1e3, "GB_per_s_written": bytes_written / dt / 1e9, "frac_of_8TBps": bytes_written / dt / 8e12}))
