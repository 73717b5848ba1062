#!/usr/bin/env python3
"""Dumps the episode lengths and start boards of one self-play collect (scheduling study: does the start board predict the
length of the episode's chain of searches?).  Run on the GPU box: python scripts/az_episode_lengths.py --out gpurun_out/az_len.npz"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import build_policy, synthetic_weights
from twisterl_amd import twisterl

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--searches", type=int, default=100)
ap.add_argument("--difficulty", type=int, default=8)
ap.add_argument("--out", default="gpurun_out/az_len.npz")
args = ap.parse_args()
policy = build_policy(synthetic_weights(16, seed=0), [], [])
env = twisterl.env.Puzzle(4, 4, args.difficulty, 2, 256)
coll = twisterl.collector.AZCollector(args.envs, args.searches, 1.41, 1, 32)
d = coll.collect(env, policy, seed=100)
h = d.to_numpy()
print({k: (v.shape, v.dtype) for k, v in h.items()}, d.stats)
np.savez_compressed(args.out, ep_len=h["ep_len"], ep_start=h["ep_start"], obs=h["obs"])
