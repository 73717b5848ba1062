#!/usr/bin/env python3
"""How well does a start board's distance from the solved one predict the length of its self-play episode, and what would a better
predictor buy?  Runs one collect per batch, then list-schedules the measured episode lengths (moves) on the walkers in the order of
each predictor.  GPU box:  python scripts/az_order_study.py"""
import heapq, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import build_policy, synthetic_weights
from twisterl_amd import twisterl

W = 4
def neighbours(b):
    z = b.index(0); x, y = z % W, z // W
    for dx, dy in ((-1, 0), (0, -1), (1, 0), (0, 1)):
        nx, ny = x + dx, y + dy
        if 0 <= nx < W and 0 <= ny < W:
            t = ny * W + nx; c = list(b); c[z], c[t] = c[t], c[z]; yield tuple(c)

policy = build_policy(synthetic_weights(16, seed=0), [], [])
env = twisterl.env.Puzzle(4, 4, 8, 2, 256)
for E, S, slots in ((512, 1000, 256), (4096, 1000, 1024), (4096, 100, 2048)):
    d = twisterl.collector.AZCollector(E, S, 1.41, 1, 32).collect(env, policy, seed=100).to_numpy()
    ep_len, start = d["ep_len"].astype(int), d["ep_start"].astype(int)
    boards = [tuple(int(v) % 16 for v in d["obs"][s]) for s in start]
    ident = None
    for b, n in zip(boards, ep_len):          # the solved board: where an episode of one record stands
        if n == 1: ident = b; break
    dist = {ident: 0}; frontier = [ident]
    for depth in range(1, 9):
        nxt = []
        for b in frontier:
            for c in neighbours(b):
                if c not in dist: dist[c] = depth; nxt.append(c)
        frontier = nxt
    pos = {v: i for i, v in enumerate(ident)}
    md = np.array([sum(abs(i % W - pos[v] % W) + abs(i // W - pos[v] // W) for i, v in enumerate(b) if v) for b in boards])
    ex = np.array([dist.get(b, 9) for b in boards])
    def makespan(order):
        h = [0] * slots; heapq.heapify(h)
        for e in order: heapq.heappush(h, heapq.heappop(h) + int(ep_len[e]))
        return max(h)
    idx = np.arange(E)
    res = {"episodes": E, "searches": S, "walkers": slots, "mean_len": float(ep_len.mean()), "max_len": int(ep_len.max()),
           "lower_bound": max(int(ep_len.max()), int(np.ceil(ep_len.sum() / slots))),
           "corr_manhattan": float(np.corrcoef(md, ep_len)[0, 1]), "corr_exact": float(np.corrcoef(ex, ep_len)[0, 1]),
           "makespan_by_index": makespan(idx), "makespan_manhattan": makespan(sorted(idx, key=lambda e: (-md[e], e))),
           "makespan_exact": makespan(sorted(idx, key=lambda e: (-ex[e], e))), "makespan_oracle": makespan(sorted(idx, key=lambda e: (-ep_len[e], e))),
           "mean_len_by_exact_distance": {int(k): round(float(ep_len[ex == k].mean()), 2) for k in sorted(set(ex))},
           "share_at_depth_limit_by_exact_distance": {int(k): round(float((ep_len[ex == k] == ep_len.max()).mean()), 3) for k in sorted(set(ex))}}
    print(json.dumps(res))
