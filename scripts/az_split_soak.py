#!/usr/bin/env python3
"""Many repeats of split-shape self-play collects (walker kernel + engine kernel, mailboxes): every repeat must give the bytes of the first
one and no watchdog may fire -- which engine serves which request when differs from run to run.   python scripts/az_split_soak.py [repeats]"""
import sys, time
sys.path.insert(0, ".")
import torch
from bench import build_policy, synthetic_weights, transpose_twist
from twisterl_amd import _lib, twisterl

def digest(d):
    t = d.to_torch(); out = []
    for k in sorted(t):
        x = t[k]
        v = x.view(torch.uint8).to(torch.int64) if x.dtype in (torch.uint8, torch.int8) else x.contiguous().view(torch.int32).to(torch.int64)
        w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 1000003
        out.append(int((v.reshape(-1) * w).sum().item()))
    return tuple(out)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
op, ap = transpose_twist(4)
pols = {"no twists": build_policy(synthetic_weights(16, seed=0), [], []), "twists": build_policy(synthetic_weights(16, seed=0), op, ap)}
env = twisterl.env.Puzzle(4, 4, 8, 2, 256)
for E, S, rep, pk in ((4096, 100, n, "no twists"), (2048, 64, n, "twists"), (8192, 48, n, "no twists"), (3000, 200, max(4, n // 2), "twists"), (4096, 1000, max(3, n // 8), "no twists")):
    c = twisterl.collector.AZCollector(E, S, 1.41, 1, 32)
    ref, bad, t0 = None, 0, time.perf_counter()
    for i in range(rep):
        d = c.collect(env, pols[pk], seed=11)
        dg = digest(d)
        if ref is None: ref = dg
        elif dg != ref: bad += 1
    print(f"split soak: {E} x {S}, {pk}: launch {d.stats['rollout_blocks']} x {d.stats['rollout_threads']}, {rep} repeats in {time.perf_counter() - t0:.1f} s, "
          f"mismatching repeats {bad}, watchdog {_lib.debug_counters(14)[13]}", flush=True)
