#!/usr/bin/env python3
"""Writes tests/golden/ppo_puzzle8_v1_weights.npz: the eight tensors of the reference's trained Puzzle-8 checkpoint
(examples/ppo_puzzle8_v1.pt, Apache-2.0) as plain float32 arrays in torch layout ([out][in]).

Run in the build container only (the reference tree does not travel to the GPU box):
    python scripts/make_trained_fixture.py
The checkpoint is DATA (a state dict read with weights_only=True), not source.  The fixture pins obs encoding + weight
layout + action semantics end to end: a policy trained by the reference against ITS Puzzle only solves OUR Puzzle if all
three agree (tests/test_oracle_golden.py, tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/examples/ppo_puzzle8_v1.pt"
DST = os.path.join(ROOT, "tests", "golden", "ppo_puzzle8_v1_weights.npz")

sd = torch.load(SRC, map_location="cpu", weights_only=True)
want = {"embeddings.weight": (512, 81), "embeddings.bias": (512,), "common.0.weight": (256, 512), "common.0.bias": (256,),
        "action.0.weight": (4, 256), "action.0.bias": (4,), "value.0.weight": (1, 256), "value.0.bias": (1,)}
out = {}
for k, shape in want.items():
    t = sd[k]
    assert tuple(t.shape) == shape and t.dtype == torch.float32, (k, tuple(t.shape), t.dtype)
    out[k.replace(".", "__")] = np.ascontiguousarray(t.numpy())
extra = sorted(set(sd) - set(want))
assert not extra, extra
np.savez_compressed(DST, **out)
print(DST, os.path.getsize(DST), "bytes;", sum(v.size for v in out.values()), "floats")
