"""Ragged episode lengths (one scramble move, long horizon: most episodes end early): persistent-lane mode vs one
workgroup per 256 episodes.  Puzzle-15, benchmark-size policy, 262,144 episodes (or argv[1]), f32 exact mode."""
import os, sys, time
sys.path.insert(0, ".")
import torch
import bench
from tests.util import puzzle_transpose_twist
from twisterl_amd import twisterl
op, ap = puzzle_transpose_twist(4)
pol = bench.build_policy(bench.synthetic_weights(16), op, ap)
env = twisterl.env.Puzzle(4, 4, 1, 128, 256)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
coll = twisterl.collector.PPOCollector(E, 0.995, 0.995, 1)
for mode in ("persistent", "plain"):
    if mode == "plain":
        from twisterl_amd import _lib
        _lib.check(_lib.lib().tw_set_launch_option(_lib.TW_OPT_NO_PERSIST, 1))
    coll.collect(env, pol, seed=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for i in range(3):
        d = coll.collect(env, pol, seed=1 + i); n += len(d); st = d.stats
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "records/collect", n // 3, "mean len %.1f" % (n / 3 / E), "ms/collect %.1f" % (dt / 3 * 1e3), "rollout ms %.1f" % st["ms_rollout"],
          "records/s %.3g" % (n / dt))
