"""Child process of test_split_shape_falls_back_when_its_kernels_cannot_run_side_by_side (tests/test_gpu_parity.py): with TW_OPT_AZ_VARIANT
+ 2048 (the split shape's two kernels launched one after the other -- what a counter-collecting profiler does to them) it runs one self-play
collect of a size that takes the split shape and prints a digest of the result, the launch shape, the same once more (the process keeps the
single-kernel shapes after the first failure) and the digest of the collect on the pinned single-kernel shape.  A process of its own because
the switch is for the life of the process.  Not collected by pytest."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.util import amd_policy, make_policy_arrays      # noqa: E402
from twisterl_amd import _lib, twisterl                    # noqa: E402


def digest(d):
    a = d.to_numpy()
    h = hashlib.sha256()
    for k in sorted(a):
        h.update(k.encode()); h.update(a[k].tobytes())
    return h.hexdigest()


pol = amd_policy(make_policy_arrays(9, seed=5, emb=64, hidden=128))
env = twisterl.env.Puzzle(3, 3, 3, 2, 256)
E = int(sys.argv[1])
c = twisterl.collector.AZCollector(E, 16, 1.41, 1, 1)         # (16 searches per move: the walker kernel takes up to 32 episodes per CU)
with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 2048):
    d = c.collect(env, pol, seed=3)                    # the split shape is what this size takes -- and cannot run this way
first = {"digest": digest(d), "launch": [d.stats["rollout_blocks"], d.stats["rollout_threads"]]}
d = c.collect(env, pol, seed=3)                        # the process keeps the single-kernel shapes
again = {"digest": digest(d), "launch": [d.stats["rollout_blocks"], d.stats["rollout_threads"]]}
with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 1024):
    ref = digest(c.collect(env, pol, seed=3))
print(json.dumps({"first": first, "again": again, "pinned_single_kernel": ref}), flush=True)
