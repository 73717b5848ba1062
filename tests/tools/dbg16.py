"""Debug aid: where do precision="fp16" logits deviate from the oracle's ARITH_F16 forward?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.util import amd_policy, make_policy_arrays, oracle_policy, puzzle_transpose_twist
import oracle.oracle as oracle
from twisterl_amd import twisterl as tw

w, h, diff, emb, hidden, E, twists = [int(x) for x in sys.argv[1:8]]
n2 = w * h
arrs = make_policy_arrays(n2, seed=1, emb=emb, hidden=hidden)
op_, ap_ = puzzle_transpose_twist(w) if twists else ((), ())
gp, op = amd_policy(arrs, op_, ap_), oracle_policy(oracle, arrs, op_, ap_)
a = tw.collector.PPOCollector(E, 0.995, 0.995, 1, seed=13, merge_order=False, precision="fp16").collect(tw.env.Puzzle(w, h, diff, 2, 256), gp, seed=13).to_numpy()
L = a["ep_len"].astype(int); st = np.concatenate([[0], np.cumsum(L)])
bad = {}
tot = {}
for e in range(E):
    for t in range(L[e]):
        r = st[e] + t
        obs = a["obs"][r].astype(np.int64)
        board = obs - np.arange(n2) * n2
        zi = int(np.where(board == 0)[0][0])
        masks = [zi % w > 0, zi // w > 0, zi % w < w - 1, zi // w < h - 1]
        lo, vo = op.forward(obs.tolist(), masks, perm=int(a["perms"][r]), arith=oracle.ARITH_F16)
        err = max(np.max(np.abs(np.asarray(lo, np.float32) - a["logits"][r])), abs(vo - a["values"][r]))
        key = ("tile", (e % 64) // 32, "t", min(t, 3))
        tot[key] = tot.get(key, 0) + 1
        if err > 1e-4:
            bad[key] = bad.get(key, 0) + 1
for k in sorted(tot):
    print(k, "bad", bad.get(k, 0), "of", tot[k])
