"""CPU: tw_gather_plan, the placement function tw_gather_submit calls (twisterl_amd/csrc/tw_comm.hip) -- pure host code behind
the C ABI, no device and no RCCL -- held against twisterl_amd.dist.plan_step (the twin the gloo tests run under
torch.distributed) and against the un-sharded collect, at world 2 / 3 / 4 / 8.

Every rank keeps its own tw_gather_state and plans every step from the same gathered counts, as the ranks of a real run do;
a numpy "transport" then moves each chunk's pieces to the offsets the ROOT's plan names, and the view
[front - tail, front + pos) of the root's buffers must be the reference merge order [E-1, 0, .., E-2]
(rust/src/collector/collector.rs:40-46) of an un-sharded collect, bit for bit.
"""
import ctypes as C

import numpy as np
import pytest

from twisterl_amd import _lib
from twisterl_amd.dist import chunk_range, plan_step, step_bounds

NCNT = 8   # TW_GATHER_COUNTS


class Piece(C.Structure):
    _fields_ = [("src_lo", C.c_uint64), ("src_hi", C.c_uint64), ("dst", C.c_uint64)]


class State(C.Structure):
    _fields_ = [("steps", C.c_uint32), ("step", C.c_uint32), ("max_records", C.c_uint64), ("max_episode_records", C.c_uint64),
                ("pos", C.c_uint64), ("front", C.c_uint64), ("cap", C.c_uint64), ("tail", C.c_uint64)]


def _plan(st, world, counts):
    """One tw_gather_plan call: (rc, tail_rank, pieces per rank as lists of (lo, hi, dst))."""
    L = _lib.lib()
    flat = (C.c_uint64 * (world * NCNT))(*[int(x) for row in counts for x in row])
    tr = C.c_int32(-7)
    npc = (C.c_uint32 * world)()
    pcs = (Piece * (2 * world))()
    rc = L.tw_gather_plan(C.cast(C.byref(st), C.c_void_p), world, flat, C.byref(tr), npc, C.cast(pcs, C.c_void_p))
    pieces = [[(int(pcs[2 * r + i].src_lo), int(pcs[2 * r + i].src_hi), int(pcs[2 * r + i].dst)) for i in range(npc[r])] for r in range(world)]
    return rc, tr.value, pieces


FIELDS = ("obs", "logits", "perms", "values", "rewards", "actions", "advs", "rets")


def _fields(d):
    return {"obs": d.obs, "logits": d.logits, "perms": d.perms, "values": d.values, "rewards": d.rewards, "actions": d.actions,
            "advs": d.additional_data["advs"], "rets": d.additional_data["rets"]}


@pytest.mark.parametrize("world,E,chunks", [(2, 37, 3), (3, 10, 4), (2, 2, 1), (4, 23, 2), (3, 2, 4), (4, 1, 3), (2, 9, 1), (2, 20, 0), (3, 11, 0),
                                            (8, 50, 3), (8, 100, 0), (8, 8, 1), (4, 64, 1)])
def test_plan_matches_dist_py_and_the_unsharded_merge_order(oracle, world, E, chunks):
    O = oracle
    from tests.util import make_policy_arrays
    pol = O.Policy(*make_policy_arrays(9, seed=5, emb=32, hidden=32))
    env = O.Puzzle(3, 3, 6, 2, 256)
    t_max = 2 * 6 + 1
    coll = lambda n, off, merge: O.ppo_collect(env, pol, n, 0.995, 0.995, seed=77, episode_offset=off, arith=O.ARITH_CHAIN, det_log=True, merge_order=merge)
    full = coll(E, 0, True)
    step_eps = 3 if chunks == 0 else None
    bounds = step_bounds(E, world, chunks, step_eps)
    K = len(bounds)
    states = [State(steps=K, step=0, max_records=E * t_max if K > 1 else 0, max_episode_records=t_max if K > 1 else 0) for _ in range(world)]
    py_front, py_pos, py_tail = None, 0, 0
    bufs, ep_len_root = None, np.zeros(E, dtype=np.int64)
    for s in range(K):
        chunks_s, counts = [], []
        for r in range(world):
            a, b = chunk_range(bounds, s, r, world)
            d = coll(b - a, a, False) if b > a else None
            chunks_s.append(d)
            n = int(d.obs.shape[0]) if d is not None else 0
            counts.append([n, int(d.ep_len[-1]) if d is not None else 0, b - a, a, 1 if n else 0, 4 if n else 0, 0, 0])
        # the Python twin (what TrajectoryGather.submit does)
        cnts, lasts = [c[0] for c in counts], [c[1] for c in counts]
        tail, tail_rank, _, _ = plan_step(cnts, lasts, s == K - 1, 0, py_pos)
        if s == 0:
            py_front = tail if K == 1 else t_max
        tail, tail_rank, py_pieces, py_pos = plan_step(cnts, lasts, s == K - 1, py_front, py_pos)
        if s == K - 1:
            py_tail = tail
        # every rank plans the step with its own state: all of them must agree with each other and with dist.py
        for r in range(world):
            rc, tr, pieces = _plan(states[r], world, counts)
            assert rc == 0, _lib.last_error()
            assert tr == tail_rank and pieces == py_pieces, (s, r)
            st = states[r]
            assert (st.step, st.pos, st.front) == (s + 1, py_pos, py_front)
        # transport: the chunks' pieces to the offsets of the ROOT's plan
        st0 = states[0]
        if bufs is None:
            ref = _fields(full)
            bufs = {k: np.zeros((st0.cap,) + ref[k].shape[1:], dtype=ref[k].dtype) for k in FIELDS}
        for r in range(world):
            if chunks_s[r] is None:
                continue
            f = _fields(chunks_s[r])
            for (lo, hi, dst) in py_pieces[r]:
                assert dst + hi - lo <= st0.cap
                for k in FIELDS:
                    bufs[k][dst:dst + hi - lo] = f[k][lo:hi]
            a = counts[r][3]
            ep_len_root[a:a + counts[r][2]] = chunks_s[r].ep_len
    st0 = states[0]
    assert st0.tail == py_tail and st0.step == K
    total, a0 = st0.pos + st0.tail, st0.front - st0.tail
    assert total == full.obs.shape[0]
    ref = _fields(full)
    for k in FIELDS:
        got = bufs[k][a0:a0 + total]
        assert got.tobytes() == ref[k].tobytes(), k
    index_order = coll(E, 0, False)
    assert np.array_equal(ep_len_root, index_order.ep_len)
    # one plan too many
    rc, _, _ = _plan(states[0], world, counts)
    assert rc == _lib.TW_ERR_INVALID


def test_plan_rejects_what_does_not_fit():
    row = lambda n, last, eps, off: [n, last, eps, off, 1, 4, 0, 0]
    # more records than max_records + slack
    st = State(steps=2, step=0, max_records=10, max_episode_records=5)
    rc, _, _ = _plan(st, 2, [row(8, 3, 2, 0), row(8, 2, 2, 2)])
    assert rc == _lib.TW_ERR_INVALID and st.step == 0
    # the last episode is longer than max_episode_records
    st = State(steps=2, step=1, max_records=100, max_episode_records=5, pos=10, front=5, cap=105)
    rc, _, _ = _plan(st, 2, [row(8, 3, 2, 0), row(8, 7, 2, 2)])
    assert rc == _lib.TW_ERR_INVALID
    # several steps need the bounds
    st = State(steps=2, step=0)
    rc, _, _ = _plan(st, 1, [row(8, 3, 2, 0)])
    assert rc == _lib.TW_ERR_INVALID
    # counts that contradict themselves
    st = State(steps=1, step=0)
    rc, _, _ = _plan(st, 2, [row(3, 4, 1, 0), row(2, 2, 1, 1)])
    assert rc == _lib.TW_ERR_INVALID
    # one step, one rank: exact size, episode E-1 first
    st = State(steps=1, step=0)
    rc, tr, pieces = _plan(st, 1, [row(9, 4, 3, 0)])
    assert rc == 0 and tr == 0 and (st.front, st.cap, st.pos, st.tail) == (4, 9, 5, 4)
    assert pieces == [[(5, 9, 0), (0, 5, 4)]]
    # ranks without episodes in the last step: the tail belongs to the last NON-EMPTY chunk
    st = State(steps=1, step=0)
    rc, tr, pieces = _plan(st, 4, [row(6, 2, 2, 0), row(3, 3, 1, 2), [0] * 8, [0] * 8])
    assert rc == 0 and tr == 1 and pieces == [[(0, 6, 3)], [(0, 3, 0)], [], []] and (st.front, st.pos, st.tail) == (3, 6, 3)
