"""Deviation of each precision mode's logits/values from the reference f32 arithmetic (oracle ARITH_REF) on the mode's own
trajectory: Puzzle-15, benchmark-size policy, twists.  Prints max and mean absolute deviation per mode."""
import sys
sys.path.insert(0, ".")
import numpy as np
from tests.util import amd_policy, make_policy_arrays, oracle_policy, puzzle_transpose_twist
import oracle.oracle as O
from twisterl_amd import twisterl as tw

arrs = make_policy_arrays(16, seed=1)
op_, ap_ = puzzle_transpose_twist(4)
gp, op = amd_policy(arrs, op_, ap_), oracle_policy(O, arrs, op_, ap_)
env = tw.env.Puzzle(4, 4, 8, 2, 256)
for prec in ("fp32", "fp16x2", "fp16"):
    a = tw.collector.PPOCollector(96, 0.995, 0.995, 1, seed=5, merge_order=False, precision=prec).collect(env, gp, seed=5).to_numpy()
    errs = []
    for r in range(a["obs"].shape[0]):
        obs = a["obs"][r].astype(np.int64)
        masks = (a["logits"][r] != np.float32(-1e10)).tolist()
        lr, vr = op.forward(obs.tolist(), masks, perm=int(a["perms"][r]), arith=O.ARITH_REF)
        errs.append(max(float(np.max(np.abs(a["logits"][r] - np.asarray(lr, np.float32)))), abs(float(a["values"][r]) - vr)))
    print(prec, "records", len(errs), "max |dev| vs reference f32 arithmetic %.3g" % max(errs), "mean %.3g" % (sum(errs) / len(errs)))
