import sys, numpy as np
sys.path.insert(0, ".")
from oracle import oracle as O
from twisterl_amd import twisterl
from tests.util import amd_policy, make_policy_arrays, oracle_policy
for (n2, w, emb, hid) in [(9,3,32,32),(9,3,64,64),(16,4,512,256)]:
    arrs = make_policy_arrays(n2, seed=1, emb=emb, hidden=hid)
    gp, op = amd_policy(arrs), oracle_policy(O, arrs)
    E, diff = 300, 5
    g = twisterl.collector.PPOCollector(E, 0.995, 0.995, 1, merge_order=False).collect(twisterl.env.Puzzle(w, w, diff, 2, 256), gp, seed=11).to_numpy()
    o = O.ppo_collect(O.Puzzle(w, w, diff, 2, 256), op, E, 0.995, 0.995, seed=11, arith=O.ARITH_CHAIN, det_log=True, merge_order=False)
    n = min(len(o.values), len(g["values"]))
    bits = lambda x: np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    dv = np.nonzero(bits(g["values"][:n]) != bits(o.values[:n]))[0]
    dl = np.nonzero((bits(g["logits"][:n]) != bits(o.logits[:n])).any(1))[0]
    print(n2, emb, hid, "records", len(g["values"]), len(o.values), "first value diff", dv[:3], "first logit diff", dl[:3])
    if len(dv):
        i = dv[0]; print("  v", g["values"][i], o.values[i], "logits", g["logits"][i], o.logits[i])
    elif len(dl):
        i = dl[0]; print("  logits", g["logits"][i], o.logits[i], "values", g["values"][i], o.values[i])
