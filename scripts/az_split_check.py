#!/usr/bin/env python3
"""Split walker shape (TW_OPT_AZ_VARIANT + 512: walkers and engine as two kernels) against the in-workgroup shapes (+ 1024): same bytes.
   python scripts/az_split_check.py [E S]..."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from bench import build_policy, synthetic_weights
from twisterl_amd import _lib, twisterl
pol = build_policy(synthetic_weights(16, seed=0), [], [])
env = twisterl.env.Puzzle(4, 4, 8, 2, 256)
args = [int(x) for x in sys.argv[1:]] or [300, 20, 2000, 30]
for E, S in zip(args[0::2], args[1::2]):
    c = twisterl.collector.AZCollector(E, S, 1.41, 1, 32)
    with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 1024):
        ref = c.collect(env, pol, seed=5)
    a = ref.to_numpy()
    with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 512):
        t0 = time.perf_counter(); g = c.collect(env, pol, seed=5); dt = time.perf_counter() - t0
    b = g.to_numpy()
    same = all(np.array_equal(a[k], b[k]) for k in a)
    print(f"{E} x {S}: split {g.stats['rollout_blocks']} x {g.stats['rollout_threads']} {g.stats['ms_rollout']:.2f} ms | in-workgroup {ref.stats['rollout_blocks']} x {ref.stats['rollout_threads']} {ref.stats['ms_rollout']:.2f} ms | same bytes: {same}", flush=True)
