"""Experiment: is v_mfma_f32_32x32x16_f16 'exact products, ONE f32 rounding per MFMA (acc + sum of 16 products)'?
Emulate the f16-input engine's forward with that model in float64/float32 numpy and compare bit for bit with the GPU."""
import sys
sys.path.insert(0, ".")
import numpy as np
from tests.util import amd_policy, make_policy_arrays, puzzle_transpose_twist
from twisterl_amd import twisterl as tw

n2, emb, hidden = 16, 512, 256
arrs = make_policy_arrays(n2, seed=1, emb=emb, hidden=hidden)
gp = amd_policy(arrs)
we, be, (w1, b1, _), (wa, ba, _), (wv, bv, _) = arrs[0], arrs[1], arrs[2][0], arrs[3][0], arrs[4][0]
T16 = we.astype(np.float16).astype(np.float64)            # [obs][emb]
W16 = w1.reshape(emb, hidden).astype(np.float16).astype(np.float64)
A16 = wa.reshape(hidden, 4).astype(np.float16).astype(np.float64)
V16 = wv.reshape(hidden).astype(np.float16).astype(np.float64)
f32 = np.float32

def fwd(obs, model):
    # embedding: accumulator starts at the bias (C operand), one MFMA per cell: exactly one non-zero product
    e = be.astype(np.float32).copy()
    for c in range(16):
        e = (e.astype(np.float64) + T16[obs[c]]).astype(np.float32)
    h0 = np.maximum(e, 0).astype(np.float16).astype(np.float64)
    acc = b1.astype(np.float32).copy()
    for g in range(emb // 16):                              # one MFMA per 16 consecutive k
        k = slice(16 * g, 16 * g + 16)
        if model == "fused":
            acc = (acc.astype(np.float64) + h0[k] @ W16[k]).astype(np.float32)
        else:                                               # sequential f32 adds inside the MFMA
            for kk in range(16 * g, 16 * g + 16):
                acc = (acc.astype(np.float64) + h0[kk] * W16[kk]).astype(np.float32)
    h1 = np.maximum(acc, 0).astype(np.float16).astype(np.float64)
    la = np.zeros(4, np.float32); v = np.float32(0)
    for g in range(hidden // 16):
        k = slice(16 * g, 16 * g + 16)
        if model == "fused":
            la = (la.astype(np.float64) + h1[k] @ A16[k]).astype(np.float32)
            v = f32(np.float64(v) + h1[k] @ V16[k])
        else:
            for kk in range(16 * g, 16 * g + 16):
                la = (la.astype(np.float64) + h1[kk] * A16[kk]).astype(np.float32)
                v = f32(np.float64(v) + h1[kk] * V16[kk])
    return (la + ba.astype(np.float32)).astype(np.float32), f32(v + bv[0])

env = tw.env.Puzzle(4, 4, 6, 2, 256)
a = tw.collector.PPOCollector(32, 0.99, 0.95, 1, seed=3, merge_order=False, precision="fp16").collect(env, gp, seed=3).to_numpy()
for model in ("fused", "seq"):
    nbad = 0; worst = 0.0
    for r in range(min(200, a["obs"].shape[0])):
        lg, v = fwd(a["obs"][r].astype(int), model)
        m = a["logits"][r] != np.float32(-1e10)
        d = max(float(np.max(np.abs(lg[m] - a["logits"][r][m]))), abs(float(v) - float(a["values"][r])))
        worst = max(worst, d)
        if not (np.array_equal(lg[m].view(np.uint32), a["logits"][r][m].view(np.uint32)) and f32(v).view(np.uint32) == a["values"][r].view(np.uint32)):
            nbad += 1
    print(model, "records not bit-equal:", nbad, "of", min(200, a["obs"].shape[0]), "worst abs dev %.3g" % worst)
