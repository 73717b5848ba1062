#!/usr/bin/env python3
"""CPU model of the walker kernel's look-ahead (tw_mcts_deep.hip): how many policy forwards does one move's search need when
every forward evaluates the demanded leaf plus `quota` more nodes of the tree that exist but hold no network output yet?
The search itself is the sequential one (its order does not depend on what is evaluated ahead), so a move is replayed from
the log of its node creations and leaf demands; candidate policies:
  creation  nodes in creation order from a cursor (what the kernel does)
  newest    most recently created first
  ucb       the unevaluated nodes a greedy UCB descent (statistics of the moment) would reach first
Run here (CPU, oracle only):  python scripts/spec_sim.py [--episodes 6] [--searches 100]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle
from tests.util import make_policy_arrays, oracle_policy

ap = argparse.ArgumentParser()
ap.add_argument("--episodes", type=int, default=6)
ap.add_argument("--searches", type=int, default=100)
ap.add_argument("--difficulty", type=int, default=8)
args = ap.parse_args()
f32 = np.float32
pol = oracle_policy(oracle, make_policy_arrays(16, seed=0, emb=512, hidden=256))
oracle.set_det_exp(True)
C_ = 1.41


def move_log(puz, seed, key, t, S):
    """sequential predict_probs_mcts (search.rs:104-189) on an oracle Puzzle; returns (probs, log) with log entries
    ('c', node, parent) for a creation and ('d', node) for a leaf that needs the network (the root's own evaluation is not in it)"""
    nodes = [dict(st=puz.clone(), parent=-1, action=-1, prior=f32(0), visit=1, vsum=f32(0), ch=[])]
    log = []
    def expand(i, pri):
        for a in range(4):
            if not pri[a] > 0: continue
            st = nodes[i]["st"].clone(); st.step(a)
            nodes.append(dict(st=st, parent=i, action=a, prior=f32(pri[a]), visit=0, vsum=f32(0), ch=[]))
            nodes[i]["ch"].append(len(nodes) - 1); log.append(("c", len(nodes) - 1, i))
    pr, _ = pol.full_predict(puz.observe(), puz.masks(), arith=oracle.ARITH_CHAIN)
    expand(0, pr)
    for it in range(S):
        i = 0
        while nodes[i]["ch"]:
            par, best, bu = nodes[i], None, f32(-np.inf)
            for c in par["ch"]:
                ch = nodes[c]
                q = f32(0) if ch["visit"] == 0 else f32(ch["vsum"] / f32(ch["visit"]))
                d = f32(f32(f32(C_) * f32(np.sqrt(f32(par["visit"])) / f32(f32(ch["visit"]) + f32(1)))) * ch["prior"])
                if f32(q + d) > bu: best, bu = c, f32(q + d)
            i = best
        st = nodes[i]["st"]; value = f32(st.reward())
        if not st.is_final():
            log.append(("d", i))
            pr, nv = pol.full_predict(st.observe(), st.masks(), arith=oracle.ARITH_CHAIN)
            expand(i, pr)
            w = oracle.philox4x32_10([key & 0xFFFFFFFF, key >> 32, it, 4 | (t << 8)], [seed & 0xFFFFFFFF, seed >> 32])
            i = nodes[i]["ch"][oracle.sample_weighted([nodes[c]["prior"] for c in nodes[i]["ch"]], float(f32(w[0] >> 8) * f32(1.0 / 16777216.0)))]
            value = f32(nv)
        j = i
        while j >= 0:
            nodes[j]["vsum"] = f32(nodes[j]["vsum"] + value); nodes[j]["visit"] += 1; j = nodes[j]["parent"]
    mp = np.zeros(4, np.float32)
    for c in nodes[0]["ch"]: mp[nodes[c]["action"]] = nodes[c]["visit"]
    final = {i: nodes[i]["st"].is_final() for i in range(len(nodes))}
    return (mp / mp.sum() if mp.sum() > 0 else np.full(4, 0.25, np.float32)), log, final


def forwards(log, final, quota, policy):
    """forwards needed for the leaf demands of one move (the root's evaluation not counted)"""
    exist, done, order, n_fwd = [], set(), [], 0
    parent = {}
    for ev in log:
        if ev[0] == "c":
            exist.append(ev[1]); parent[ev[1]] = ev[2]
            continue
        leaf = ev[1]
        if leaf in done:
            continue
        n_fwd += 1
        done.add(leaf)
        cand = [n for n in exist if n not in done and not final[n]]
        if policy == "creation": pick = cand[:quota]
        elif policy == "newest": pick = cand[::-1][:quota]
        else:   # siblings of the demanded leaf first, then its cousins (children of the parent's siblings), then creation order
            sib = [n for n in cand if parent.get(n) == parent.get(leaf)]
            cous = [n for n in cand if parent.get(parent.get(n, -1), -2) == parent.get(parent.get(leaf, -1), -3) and n not in sib]
            rest = [n for n in cand if n not in sib and n not in cous]
            pick = (sib + cous + rest)[:quota]
        done.update(pick)
    return n_fwd


tot = {}
demands = 0
for ep in range(args.episodes):
    puz = oracle.Puzzle(4, 4, args.difficulty, 2, 256); puz.reset(seed=100, episode=ep)
    t = 0
    while True:
        probs, log, final = move_log(puz, 100, ep, t, args.searches)
        demands += sum(1 for e in log if e[0] == "d")
        for quota in (0, 3, 7, 15):
            for policy in ("creation", "newest", "family"):
                tot[(quota, policy)] = tot.get((quota, policy), 0) + forwards(log, final, quota, policy)
        w = oracle.philox4x32_10([ep, 0, t, 3], [100, 0])
        act = oracle.sample_weighted(probs, float(f32(w[0] >> 8) * f32(1.0 / 16777216.0)))
        if puz.is_final(): break
        puz.step(act); t += 1
print("leaf demands:", demands)
for quota in (0, 3, 7, 15):
    print("quota", quota, {p: f"{tot[(quota, p)]} forwards ({tot[(quota, p)] / demands:.2f} per demand)" for p in ("creation", "newest", "family")})
