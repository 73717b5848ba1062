"""Per-step wall time of repeated collects (diagnostic for allocator behaviour)."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from tests.util import puzzle_transpose_twist
from twisterl_amd import twisterl
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
op, ap = puzzle_transpose_twist(4)
pol = bench.build_policy(bench.synthetic_weights(16), op, ap)
env = twisterl.env.Puzzle(4, 4, 128, 2, 256)
coll = twisterl.collector.PPOCollector(262144, 0.995, 0.995, 1, precision=prec)
ts = []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d = coll.collect(env, pol, seed=1000 + i)
    t1 = time.perf_counter()
    st = d.stats
    del d
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1), round(st["ms_total"], 1)))
print(prec, ts)
