"""GPU parity tests (run on the MI355X box with -m gpu).  Every test calls the product through
its public surface (twisterl_amd -> C ABI -> HIP kernels) and checks it against the CPU oracle.

Bars (BASELINE.json north_star): state transitions / masks / obs / rewards / actions BIT-EXACT;
in the f32 "exact" mode logits, values, advantages and returns are bit-exact too (the oracle's
TWO_ARITH_CHAIN order is what the f32 MFMA computes); against the reference's un-fused order the
tolerance is 1e-5.
"""
import numpy as np
import pytest

from tests.util import amd_policy, f32_bits, make_policy_arrays, oracle_policy, puzzle_transpose_twist
from twisterl_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tw():
    import twisterl_amd
    assert twisterl_amd.device_count() >= 1, "no GPU visible: the -m gpu tests need the MI355X box"
    return twisterl_amd.twisterl


def _assert_same_collect(g, o, n_cells):
    """g: twisterl_amd CollectedData (device), o: oracle Collected.  Bitwise equality."""
    a = g.to_numpy()
    assert a["obs"].shape == (o.obs.shape[0], n_cells)
    assert np.array_equal(a["ep_len"], o.ep_len)
    assert np.array_equal(a["obs"].astype(np.int64), o.obs)
    assert np.array_equal(a["actions"].astype(np.int64), o.actions)
    assert np.array_equal(a["perms"].astype(np.int32), o.perms)
    assert np.array_equal(f32_bits(a["rewards"]), f32_bits(o.rewards))
    assert np.array_equal(f32_bits(a["logits"]), f32_bits(o.logits))
    assert np.array_equal(f32_bits(a["values"]), f32_bits(o.values))
    assert np.array_equal(f32_bits(a["advs"]), f32_bits(o.additional_data["advs"]))
    assert np.array_equal(f32_bits(a["rets"]), f32_bits(o.additional_data["rets"]))


def _pair(oracle, n2, seed, emb, hidden, twists=False, scale=1.0):
    arrs = make_policy_arrays(n2, seed=seed, emb=emb, hidden=hidden, scale=scale)
    op, ap = puzzle_transpose_twist(int(round(n2 ** 0.5))) if twists else ((), ())
    return amd_policy(arrs, op, ap), oracle_policy(oracle, arrs, op, ap)


# ------------------------------------------------------------------------------ policy forward
@pytest.mark.parametrize("n2,emb,hidden", [(9, 64, 32), (16, 512, 256)])
def test_policy_evaluate_matches_oracle(tw, oracle, n2, emb, hidden):
    gp, op = _pair(oracle, n2, 3, emb, hidden, twists=True)
    rng = np.random.default_rng(0)
    n = 64
    boards = np.stack([rng.permutation(n2) for _ in range(n)])
    obs = np.arange(n2)[None, :] * n2 + boards
    masks = rng.integers(0, 2, size=(n, 4)).astype(np.uint8)
    masks[:, 0] = 1
    perms = rng.integers(-1, 2, size=n).astype(np.int32)
    from twisterl_amd import _lib
    la, va = gp.evaluate_batch(_lib.TW_EVAL_FORWARD, obs, masks, perms)
    pa, pv = gp.evaluate_batch(_lib.TW_EVAL_PREDICT, obs, masks, perms)
    fa, fv = gp.evaluate_batch(_lib.TW_EVAL_FULL_PREDICT, obs, masks)
    for i in range(n):
        lo, vo = op.forward(obs[i].tolist(), masks[i].tolist(), perm=int(perms[i]), arith=oracle.ARITH_CHAIN)
        assert np.array_equal(f32_bits(la[i]), f32_bits(lo)) and f32_bits(va[i]) == f32_bits(vo)   # bit-exact
        lr, vr = op.forward(obs[i].tolist(), masks[i].tolist(), perm=int(perms[i]), arith=oracle.ARITH_REF)
        np.testing.assert_allclose(la[i], lr, atol=1e-5, rtol=1e-5)                                  # vs reference order
        assert abs(va[i] - vr) <= 1e-5
        po, _ = op.predict(obs[i].tolist(), masks[i].tolist(), perm=int(perms[i]), arith=oracle.ARITH_CHAIN)
        np.testing.assert_allclose(pa[i], po, atol=1e-6, rtol=1e-5)      # spec exp vs the reference's libm expf
        fo, fvo = op.full_predict(obs[i].tolist(), masks[i].tolist(), arith=oracle.ARITH_CHAIN)
        np.testing.assert_allclose(fa[i], fo, atol=1e-6, rtol=1e-5)
        assert abs(fv[i] - fvo) <= 1e-6
        oracle.set_det_exp(True)                                          # same exp spec on both sides: bit-exact
        try:
            po, pvo = op.predict(obs[i].tolist(), masks[i].tolist(), perm=int(perms[i]), arith=oracle.ARITH_CHAIN)
            fo, fvo = op.full_predict(obs[i].tolist(), masks[i].tolist(), arith=oracle.ARITH_CHAIN)
        finally:
            oracle.set_det_exp(False)
        assert np.array_equal(f32_bits(pa[i]), f32_bits(po)) and f32_bits(pv[i]) == f32_bits(pvo)
        assert np.array_equal(f32_bits(fa[i]), f32_bits(fo)) and f32_bits(fv[i]) == f32_bits(fvo)
    # single-observation PyO3-style methods
    probs, val = gp.predict(obs[0].tolist(), [bool(m) for m in masks[0]], perm=0)
    assert len(probs) == 4 and abs(sum(probs) - 1.0) < 1e-4 and isinstance(val, float)


# ------------------------------------------------------------------------------ PPO collect
@pytest.mark.parametrize("w,h,diff,emb,hidden,E,twists", [
    (3, 3, 5, 32, 32, 300, False),      # Puzzle-8, tiny net, ragged workgroup tail
    (3, 3, 12, 64, 64, 129, True),      # twists, hidden 64
    (2, 2, 3, 32, 32, 64, False),       # the reference unit test's 2x2 board
    (3, 2, 4, 32, 128, 70, False),      # non-square board (6 cells -> padded to the 9-cell kernel)
    (4, 4, 6, 512, 256, 256, True),     # Puzzle-15 at the benchmark's network size, with twists
    (3, 3, 3, 32, 32, 12800, False),    # 200 workgroups of two waves (the mid-size launch geometry)
    (3, 3, 6, 96, 256, 100, True),      # 16-column engine: three chunk pairs -> one bubble step per forward
    (4, 4, 5, 160, 128, 90, False),     # 16-column engine, hidden 128 (head buffer outside the ring area), five chunk pairs
])
def test_ppo_collect_bit_exact_vs_oracle(tw, oracle, w, h, diff, emb, hidden, E, twists):
    n2 = w * h
    if twists and w != h:
        pytest.skip("transpose twist needs a square board")
    gp, op = _pair(oracle, n2, 1, emb, hidden, twists=twists)
    genv = tw.env.Puzzle(w, h, diff, 2, 256)
    oenv = oracle.Puzzle(w, h, diff, 2, 256)
    for merge_order in (True, False):
        coll = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32},
                                         seed=11, merge_order=merge_order)
        g = coll.collect(genv, gp, seed=11)
        o = oracle.ppo_collect(oenv, op, E, 0.995, 0.995, seed=11, arith=oracle.ARITH_CHAIN, det_log=True,
                               merge_order=merge_order)
        _assert_same_collect(g, o, n2)


@pytest.mark.parametrize("w,diff,slope,max_depth,E,form", [
    (4, 100, 2, 256, 37, "16 episodes per workgroup, observations stored straight"),
    (3, 3, 50, 256, 53, "16 per workgroup, observations staged (150 records)"),
    (3, 100, 2, 256, 37, "8 per workgroup, staged (200 records)"),
    (4, 300, 2, 1000, 21, "8 per workgroup (600 records)"),
    (3, 200, 2, 600, 19, "a wave per episode, staged (400 records)"),
    (4, 511, 2, 1022, 9, "8 per workgroup at the longest horizon the environment takes (1,022 records)"),
])
def test_gae_and_compaction_in_each_of_its_forms(tw, oracle, w, diff, slope, max_depth, E, form):
    """tw_finalize.hip picks its kernel by the horizon (records per episode that fit an LDS tile): every form against the oracle's
    GAE (ppo.rs:82-92), episode counts that fill no workgroup, both output orders."""
    n2 = w * w
    gp, op = _pair(oracle, n2, 4, 32, 32, twists=True)
    genv = tw.env.Puzzle(w, w, diff, slope, max_depth)
    oenv = oracle.Puzzle(w, w, diff, slope, max_depth)
    for merge_order in (True, False):
        coll = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.99, "lambda": 0.95, "num_cores": 32}, seed=5, merge_order=merge_order)
        g = coll.collect(genv, gp, seed=5)
        o = oracle.ppo_collect(oenv, op, E, 0.99, 0.95, seed=5, arith=oracle.ARITH_CHAIN, det_log=True, merge_order=merge_order)
        _assert_same_collect(g, o, n2)


def test_ppo_collect_within_1e5_of_reference_order(tw, oracle):
    """north_star tolerance: returns/advantages within 1e-5 of the reference arithmetic.  The
    reference order (un-fused, libm log) is replayed on the GPU's own trajectory: same boards and
    actions -> its logits/values at every record, then its GAE."""
    gp, op = _pair(oracle, 16, 2, 512, 256)
    genv = tw.env.Puzzle(4, 4, 5, 2, 256)
    g = tw.collector.PPOCollector(128, 0.995, 0.995, 1, seed=5, merge_order=False).collect(genv, gp, seed=5).to_numpy()
    pos = 0
    for n in g["ep_len"].astype(int):
        obs = g["obs"][pos:pos + n].astype(np.int64)
        vals = np.empty(n, np.float32)
        for t in range(n):
            board = obs[t] - np.arange(16) * 16
            zi = int(np.where(board == 0)[0][0])
            masks = [zi % 4 > 0, zi // 4 > 0, zi % 4 < 3, zi // 4 < 3]
            lr, vr = op.forward(obs[t].tolist(), masks, arith=oracle.ARITH_REF)
            np.testing.assert_allclose(g["logits"][pos + t], lr, atol=1e-5, rtol=1e-5)
            vals[t] = vr
        np.testing.assert_allclose(g["values"][pos:pos + n], vals, atol=1e-5)
        advs, rets = oracle.gae(g["rewards"][pos:pos + n], vals, 0.995, 0.995)
        np.testing.assert_allclose(g["advs"][pos:pos + n], advs, atol=1e-5)
        np.testing.assert_allclose(g["rets"][pos:pos + n], rets, atol=1e-5)
        pos += n


def test_sharding_invariance_and_offsets(tw, oracle):
    gp, _ = _pair(oracle, 9, 4, 64, 32)
    env = tw.env.Puzzle(3, 3, 6, 2, 256)
    full = tw.collector.PPOCollector(500, 0.99, 0.95, 1, merge_order=False).collect(env, gp, seed=9).to_numpy()
    a = tw.collector.PPOCollector(200, 0.99, 0.95, 1, merge_order=False, episode_offset=0).collect(env, gp, seed=9).to_numpy()
    b = tw.collector.PPOCollector(300, 0.99, 0.95, 1, merge_order=False, episode_offset=200).collect(env, gp, seed=9).to_numpy()
    for k in ("obs", "logits", "perms", "values", "rewards", "actions", "advs", "rets", "ep_len"):
        assert np.array_equal(full[k], np.concatenate([a[k], b[k]])), k


def test_edge_cases(tw, oracle):
    gp, op = _pair(oracle, 9, 6, 32, 32)
    # difficulty 0: every episode is born solved -> exactly one record, reward 1, adv = r - v
    d = tw.collector.PPOCollector(130, 0.9, 0.9, 1).collect(tw.env.Puzzle(3, 3, 0, 2, 256), gp, seed=1).to_numpy()
    assert d["ep_len"].tolist() == [1] * 130 and np.all(d["rewards"] == 1.0)
    assert np.array_equal(f32_bits(d["advs"]), f32_bits(d["rewards"] - d["values"]))
    # a single episode
    g = tw.collector.PPOCollector(1, 0.9, 0.9, 1).collect(tw.env.Puzzle(3, 3, 7, 2, 256), gp, seed=2)
    o = oracle.ppo_collect(oracle.Puzzle(3, 3, 7, 2, 256), op, 1, 0.9, 0.9, seed=2, arith=oracle.ARITH_CHAIN, det_log=True)
    _assert_same_collect(g, o, 9)
    # maximum depth of the benchmark config: difficulty 128 -> 257-record episodes
    g = tw.collector.PPOCollector(40, 0.995, 0.995, 1).collect(tw.env.Puzzle(3, 3, 128, 2, 256), gp, seed=3)
    o = oracle.ppo_collect(oracle.Puzzle(3, 3, 128, 2, 256), op, 40, 0.995, 0.995, seed=3, arith=oracle.ARITH_CHAIN, det_log=True)
    _assert_same_collect(g, o, 9)
    assert g.to_numpy()["ep_len"].max() == 257
    # the curriculum mutates env.difficulty between collects (algorithm.py:168): read it every call
    env = tw.env.Puzzle(3, 3, 1, 2, 256)
    c = tw.collector.PPOCollector(64, 0.9, 0.9, 1)
    assert c.collect(env, gp, seed=4).to_numpy()["ep_len"].max() <= 3
    env.difficulty = 10
    assert c.collect(env, gp, seed=4).to_numpy()["ep_len"].max() > 3


def test_reference_style_attribute_access(tw, oracle):
    """The duck-typed consumer of the reference trainer (src/twisterl/rl/ppo.py:27-36)."""
    gp, _ = _pair(oracle, 9, 7, 32, 32)
    data = tw.collector.PPOCollector(**{"num_cores": 32, "num_episodes": 16, "lambda": 0.995, "gamma": 0.995}).collect(
        tw.env.Puzzle(3, 3, 4, 2, 256), gp)
    obs, logits, acts = data.obs, data.logits, data.actions
    rets, advs = data.additional_data["rets"], data.additional_data["advs"]
    perms = getattr(data, "perms", [-1] * len(data.obs))
    n = len(obs)
    assert n == len(logits) == len(acts) == len(rets) == len(advs) == len(perms) == len(data.values) == len(data.rewards)
    assert all(len(o) == 9 and all(isinstance(x, int) for x in o) for o in obs)
    assert all(len(l) == 4 for l in logits) and set(perms) == {-1}
    np_obs = np.zeros((n, 81))
    for i, o in enumerate(obs):
        np_obs[i, o] = 1.0
    assert np.all(np_obs.sum(1) == 9)
    other = tw.collector.CollectedData([[0]], [[0.1]], [0.2], [0.3], [1])
    data.merge(other)
    assert len(data.obs) == n + 1 and data.actions[-1] == 1 and data.perms[-1] == -1


def test_zero_copy_torch_views_and_single_rank_gather(tw, oracle):
    """to_torch() aliases the device buffers (no copy); the sharded-collect path run with a
    one-rank process group gives the reference merge order."""
    import torch
    gp, _ = _pair(oracle, 9, 9, 32, 32)
    env = tw.env.Puzzle(3, 3, 5, 2, 256)
    d = tw.collector.PPOCollector(200, 0.99, 0.95, 1).collect(env, gp, seed=3)
    t = d.to_torch()
    a = d.to_numpy()
    assert t["obs"].is_cuda and t["obs"].dtype == torch.uint8 and tuple(t["logits"].shape) == a["logits"].shape
    for k in ("obs", "logits", "values", "actions", "advs", "rets", "perms", "ep_len"):
        assert np.array_equal(t[k].cpu().numpy(), a[k]), k
    assert t["values"].data_ptr() == d.device_arrays()["values"].__cuda_array_interface__["data"][0]
    # one-rank "sharded" collect == plain collect in merge order
    import os
    import torch.distributed as dist
    from twisterl_amd.dist import collect_sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        merged, local = collect_sharded(tw.collector.PPOCollector(200, 0.99, 0.95, 1), env, gp, seed=3)
        for k in ("obs", "logits", "values", "actions", "advs", "rets"):
            assert np.array_equal(merged[k].cpu().numpy(), a[k]), k
        # the same shard collected in 3 chunks with the pipelined gather: identical result
        piped, parts = collect_sharded(tw.collector.PPOCollector(200, 0.99, 0.95, 1), env, gp, seed=3, chunks=3, max_episode_records=11)
        assert isinstance(parts, list) and len(parts) == 3 and sum(len(p) for p in parts) == a["obs"].shape[0]
        for k in ("obs", "logits", "values", "actions", "advs", "rets", "perms", "rewards"):
            assert np.array_equal(piped[k].cpu().numpy(), a[k]), k
        # steps sized in episodes (what bench.py does at N > 1), with CUs reserved for the transfer kernels
        piped, parts = collect_sharded(tw.collector.PPOCollector(200, 0.99, 0.95, 1), env, gp, seed=3, step_episodes=64,
                                       max_episode_records=11, reserve_cus=8)
        assert len(parts) == 4 and [p.stats["episodes"] for p in parts] == [64, 64, 64, 8]
        for k in ("obs", "logits", "values", "actions", "advs", "rets", "perms", "rewards"):
            assert np.array_equal(piped[k].cpu().numpy(), a[k]), k
    finally:
        dist.destroy_process_group()


def test_rccl_exchange_behind_the_c_abi_single_rank(tw, oracle):
    """tw_comm_* / tw_gather_* (RCCL issued by the library) with a one-rank communicator: the unique-id / init plumbing, the
    policy broadcast, and the gather's placement (tail of the last episode first, chunks at their final offsets, ep_len /
    ep_start of the merged result) for one step and for several, PPO and AlphaZero data.  The multi-rank logic is the one the
    gloo tests cover in Python; the C++ twin is checked against it here through identical results."""
    import os
    import torch.distributed as dist
    from twisterl_amd.dist import Comm, collect_sharded
    gp, _ = _pair(oracle, 9, 9, 32, 32, twists=True)
    env = tw.env.Puzzle(3, 3, 5, 2, 256)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        comm = Comm()
        assert (comm.rank, comm.world) == (0, 1)
        comm.broadcast_policy(gp, root=0)
        coll = tw.collector.PPOCollector(300, 0.99, 0.95, 1)
        want = coll.collect(env, gp, seed=3).to_numpy()                       # merge order
        # a policy of another depth: its image holds a table of device pointers, which the broadcast must leave this process's own
        from tests.util import make_deep_policy_arrays
        gdeep = amd_policy(make_deep_policy_arrays(9, seed=4, emb=32, common=(48, 32), scale=2.0))
        dwant = coll.collect(env, gdeep, seed=3).to_numpy()
        comm.broadcast_policy(gdeep, root=0)
        dgot = coll.collect(env, gdeep, seed=3).to_numpy()
        for k in dwant:
            assert np.array_equal(dgot[k], dwant[k]), k
        for kw in ({}, {"chunks": 3, "max_episode_records": 11}, {"step_episodes": 64, "max_episode_records": 11, "reserve_cus": 8}):
            merged, parts = collect_sharded(coll, env, gp, seed=3, comm=comm, **kw)
            assert sum(len(p) for p in parts) == want["obs"].shape[0]
            for k in ("obs", "logits", "perms", "values", "rewards", "actions", "advs", "rets", "ep_len", "ep_start"):
                assert np.array_equal(merged[k].cpu().numpy(), want[k]), (kw, k)
        az = tw.collector.AZCollector(90, 12, 1.41, 1, 1)
        zwant = az.collect(env, gp, seed=5).to_numpy()
        merged, _ = collect_sharded(az, env, gp, seed=5, comm=comm, chunks=4, max_episode_records=11)
        for k in ("obs", "logits", "perms", "remaining_values", "ep_len", "ep_start"):
            assert np.array_equal(merged[k].cpu().numpy(), zwant[k]), k
        comm.close()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------ boards above 16 cells (puzzle.rs:34-42)
@pytest.mark.parametrize("w,h,diff,emb,common,E,twists", [(5, 5, 6, 64, (64,), 60, True), (6, 4, 5, 32, (48, 32), 33, False), (8, 8, 3, 32, (32,), 12, False)])
def test_ppo_collect_of_boards_above_16_cells(tw, oracle, w, h, diff, emb, common, E, twists):
    """The reference's Puzzle takes any width x height (rust/src/envs/puzzle.rs:34-42); the kernels pack a board as 16 nibbles.
    Larger boards collect through the any-environment path (the Puzzle steps on the host, one batched policy launch per time
    step, obs ids beyond 255 as two bytes): every field bit-equal to the oracle's native collectors, PPO and self-play; evaluate
    and solve (plain and MCTS-guided) give the oracle's numbers and action lists."""
    from tests.util import make_deep_policy_arrays
    n2 = w * h
    arrs = make_deep_policy_arrays(n2, seed=5, emb=emb, common=common, scale=2.0)
    op_, ap_ = puzzle_transpose_twist(w) if twists else ((), ())
    gp, op = amd_policy(arrs, op_, ap_), oracle_policy(oracle, arrs, op_, ap_)
    genv, oenv = tw.env.Puzzle(w, h, diff, 2, 256), oracle.Puzzle(w, h, diff, 2, 256)
    for merge_order in (True, False):
        coll = tw.collector.PPOCollector(E, 0.995, 0.995, 32)
        coll.merge_order = merge_order
        g = coll.collect(genv, gp, seed=17)
        o = oracle.ppo_collect(oenv, op, E, 0.995, 0.995, seed=17, arith=oracle.ARITH_CHAIN, det_log=True, num_threads=8, merge_order=merge_order)
        _assert_same_collect(g, o, n2)
    for S, med in ((10, 1), (6, 2)):
        z = tw.collector.AZCollector(min(E, 20), S, 1.41, med, 32).collect(genv, gp, seed=19)
        zo = oracle.az_collect(oenv, op, min(E, 20), S, 1.41, med, seed=19, arith=oracle.ARITH_CHAIN, num_threads=8, det_math=True)
        _assert_same_az(z, zo, n2)
    # evaluate / solve of such boards (tw_evaluate_env / tw_solve_env behind tw_evaluate / tw_solve), plain and MCTS-guided
    for det, ns, S in ((True, 1, 0), (False, 3, 0), (False, 2, 5)):
        g = tw.collector.evaluate(genv, gp, num_episodes=25, deterministic=det, num_searches=ns, num_mcts_searches=S, seed=7, C=1.41,
                                  max_expand_depth=1, num_cores=32)
        o = oracle.evaluate(oenv, op, 25, det, ns, num_mcts_searches=S, seed=7, Cc=1.41, max_expand_depth=1, arith=oracle.ARITH_CHAIN, det_math=True)
        assert f32_bits(g[0]) == f32_bits(o[0]) and f32_bits(g[1]) == f32_bits(o[1]), (det, ns, S, g, o)
    start = oracle.Puzzle(w, h, diff, 2, 256); start.reset(seed=9, episode=3)
    state = start.get_state()
    genv.set_state(state); oenv.set_state(state)
    before = genv.get_state()
    for det, ns, S in ((True, 1, 0), (False, 4, 0), (False, 2, 4)):
        (gs, gr), gact = tw.collector.solve(genv, gp, det, ns, S, 1.41, 1, seed=5)
        (os_, or_), oact = oracle.solve(oenv, op, det, ns, num_mcts_searches=S, Cc=1.41, max_expand_depth=1, seed=5, arith=oracle.ARITH_CHAIN, det_math=True)
        assert (gs, f32_bits(gr)) == (os_, f32_bits(or_)) and gact == oact, (det, ns, S)
    assert genv.get_state() == before


def test_boards_of_17_to_64_cells_roll_out_on_the_device(tw, oracle):
    """5 x 5 (and 6 x 4, 6 x 6, 7 x 5, 8 x 8) PPO collects run the device kernel of tw_rollout_big.hip -- 5-bit cells in a 128-bit register up
    to 25 cells, one byte per cell in 9 / 16 registers up to 36 / 64, two-byte obs ids, the generic engine -- not the host-stepped path: 65,536 envs of a 5 x 5 board in 4,096 workgroups of 256 threads, sampled
    episodes bit-equal to the oracle on every field (the RNG is keyed by the global episode index); a small batch with the
    transpose twist compared whole, in both orders; and the same bytes as the host-stepped path (pinned through TW_OPT_FORCE_GEOM).
    Plain evaluate() and AlphaZero self-play of these boards run on the device too."""
    from tests.util import make_deep_policy_arrays
    arrs = make_deep_policy_arrays(25, seed=5, emb=64, common=(128,), scale=2.0)
    gp, op = amd_policy(arrs), oracle_policy(oracle, arrs)
    E, D = 65_536, 6
    genv, oenv = tw.env.Puzzle(5, 5, D, 2, 256), oracle.Puzzle(5, 5, D, 2, 256)
    coll = tw.collector.PPOCollector(E, 0.995, 0.995, 32)
    g = coll.collect(genv, gp, seed=41)
    assert (g.stats["rollout_blocks"], g.stats["rollout_threads"]) == (E // 16, 256)
    a = g.to_numpy()
    assert a["obs"].dtype == np.uint16 and a["obs"].shape[1] == 25
    L, S = a["ep_len"].astype(np.int64), a["ep_start"].astype(np.int64)
    assert L.sum() == len(g) and L.min() >= 1 and L.max() <= 2 * D + 1
    order = np.concatenate([[E - 1], np.arange(E - 1)])
    assert np.array_equal(S[order], np.concatenate([[0], np.cumsum(L[order])[:-1]]))
    assert np.array_equal(a["obs"] // 25, np.broadcast_to(np.arange(25), a["obs"].shape))          # obs id of cell c in [25c, 25c + 25)
    assert np.all((a["obs"] % 25).sum(axis=1) == 300)                                             # every board a permutation of 0..24
    rng = np.random.default_rng(1)
    for e in sorted(set([0, 1, E - 2, E - 1] + [int(x) for x in rng.choice(E, size=28, replace=False)])):
        o = oracle.ppo_collect(oenv, op, 1, 0.995, 0.995, seed=41, episode_offset=e, arith=oracle.ARITH_CHAIN, det_log=True, merge_order=False)
        s, ln = int(S[e]), int(L[e])
        assert ln == int(o.ep_len[0]), e
        assert np.array_equal(a["obs"][s:s + ln].astype(np.int64), o.obs), e
        assert np.array_equal(a["actions"][s:s + ln].astype(np.int64), o.actions), e
        for k, ok in (("logits", o.logits), ("values", o.values), ("rewards", o.rewards), ("advs", o.additional_data["advs"]), ("rets", o.additional_data["rets"])):
            assert np.array_equal(f32_bits(a[k][s:s + ln]), f32_bits(ok)), (e, k)
    del a, g
    for (w, h, twists) in ((5, 5, True), (6, 4, False), (6, 6, True), (7, 5, False), (8, 8, True)):
        n2 = w * h
        arrs = make_deep_policy_arrays(n2, seed=6, emb=32, common=(64, 32), scale=2.0)
        op_, ap_ = puzzle_transpose_twist(w) if twists else ((), ())
        gp, op = amd_policy(arrs, op_, ap_), oracle_policy(oracle, arrs, op_, ap_)
        genv, oenv = tw.env.Puzzle(w, h, 5, 2, 256), oracle.Puzzle(w, h, 5, 2, 256)
        for merge_order in (True, False):
            coll = tw.collector.PPOCollector(150, 0.995, 0.995, 32)
            coll.merge_order = merge_order
            g = coll.collect(genv, gp, seed=17)
            assert g.stats["rollout_threads"] == 256 and g.stats["rollout_blocks"] == 10
            o = oracle.ppo_collect(oenv, op, 150, 0.995, 0.995, seed=17, arith=oracle.ARITH_CHAIN, det_log=True, num_threads=8, merge_order=merge_order)
            _assert_same_collect(g, o, n2)
        # evaluate() without MCTS runs on the device as well (solve_big_kernel): 3,000 episodes, best of 2 sampled attempts each, and
        # the greedy form, against the oracle -- and the same numbers as the host-stepped path
        for det, ns in ((False, 2), (True, 1)):
            ge = tw.collector.evaluate(genv, gp, num_episodes=3000, deterministic=det, num_searches=ns, num_mcts_searches=0, seed=7, C=1.41,
                                       max_expand_depth=1, num_cores=32)
            oe = oracle.evaluate(oenv, op, 3000, det, ns, seed=7, arith=oracle.ARITH_CHAIN, det_math=True)
            assert f32_bits(ge[0]) == f32_bits(oe[0]) and f32_bits(ge[1]) == f32_bits(oe[1]), (w, h, det, ge, oe)
        # ... and the MCTS-guided form (mcts_big_kernel in solve mode): greedy and sampled, both expansion depths
        for det, ns, S, med in ((True, 1, 6, 1), (False, 2, 5, 2)):
            ge = tw.collector.evaluate(genv, gp, num_episodes=60, deterministic=det, num_searches=ns, num_mcts_searches=S, seed=7, C=1.41,
                                       max_expand_depth=med, num_cores=32)
            oe = oracle.evaluate(oenv, op, 60, det, ns, num_mcts_searches=S, seed=7, Cc=1.41, max_expand_depth=med, arith=oracle.ARITH_CHAIN, det_math=True)
            assert f32_bits(ge[0]) == f32_bits(oe[0]) and f32_bits(ge[1]) == f32_bits(oe[1]), (w, h, det, S, med, ge, oe)
            with _lib.launch_option(_lib.TW_OPT_FORCE_GEOM, 1):              # the host-stepped path gives the same numbers
                he = tw.collector.evaluate(genv, gp, num_episodes=60, deterministic=det, num_searches=ns, num_mcts_searches=S, seed=7, C=1.41,
                                           max_expand_depth=med, num_cores=32)
            assert f32_bits(ge[0]) == f32_bits(he[0]) and f32_bits(ge[1]) == f32_bits(he[1])
        # solve() from a given state, plain and MCTS-guided: the oracle's (success, reward) and action list, the caller's env untouched
        start = oracle.Puzzle(w, h, 5, 2, 256); start.reset(seed=9, episode=3)
        state = start.get_state()
        genv.set_state(state); oenv.set_state(state)
        before = genv.get_state()
        for det, ns, S in ((True, 1, 0), (False, 4, 0), (False, 2, 4)):
            (gs, gr), gact = tw.collector.solve(genv, gp, det, ns, S, 1.41, 1, seed=5)
            (os_, or_), oact = oracle.solve(oenv, op, det, ns, num_mcts_searches=S, Cc=1.41, max_expand_depth=1, seed=5, arith=oracle.ARITH_CHAIN, det_math=True)
            assert (gs, f32_bits(gr)) == (os_, f32_bits(or_)) and gact == oact, (w, h, det, ns, S)
            with _lib.launch_option(_lib.TW_OPT_FORCE_GEOM, 1):
                (hs, hr), hact = tw.collector.solve(genv, gp, det, ns, S, 1.41, 1, seed=5)
            assert (gs, f32_bits(gr)) == (hs, f32_bits(hr)) and gact == hact
        assert genv.get_state() == before
        with _lib.launch_option(_lib.TW_OPT_FORCE_GEOM, 1):                  # the host-stepped path (tw_ppo_collect_env)
            he = tw.collector.evaluate(genv, gp, num_episodes=300, deterministic=False, num_searches=2, num_mcts_searches=0, seed=7, C=1.41,
                                       max_expand_depth=1, num_cores=32)
        de = tw.collector.evaluate(genv, gp, num_episodes=300, deterministic=False, num_searches=2, num_mcts_searches=0, seed=7, C=1.41,
                                   max_expand_depth=1, num_cores=32)
        assert f32_bits(he[0]) == f32_bits(de[0]) and f32_bits(he[1]) == f32_bits(de[1])
        # self-play runs on the device as well (tw_mcts_big.hip: nodes without a board, the state follows the actions): 40 episodes in three
        # workgroups, both expansion depths, against the oracle's native collector and the host-stepped path
        for S, med in ((8, 1), (5, 2)):
            acoll = tw.collector.AZCollector(num_episodes=40, num_mcts_searches=S, C=1.41, max_expand_depth=med, num_cores=32, seed=23)
            z = acoll.collect(genv, gp, seed=23)
            assert (z.stats["rollout_blocks"], z.stats["rollout_threads"]) == (3, 256) and z.stats["forward_evals"] > 40 * S
            zo = oracle.az_collect(oenv, op, 40, S, 1.41, med, seed=23, arith=oracle.ARITH_CHAIN, num_threads=8, det_math=True)
            _assert_same_az(z, zo, n2)
            with _lib.launch_option(_lib.TW_OPT_FORCE_GEOM, 1):
                zh = acoll.collect(genv, gp, seed=23)
            assert zh.stats["rollout_blocks"] != 3
            za, zb = z.to_numpy(), zh.to_numpy()
            for k in za:
                assert np.array_equal(za[k], zb[k]), k
        with _lib.launch_option(_lib.TW_OPT_FORCE_GEOM, 1):                  # the host-stepped path (tw_ppo_collect_env)
            hst = coll.collect(genv, gp, seed=17)
        assert hst.stats["rollout_blocks"] != 10
        ga, ha = g.to_numpy(), hst.to_numpy()
        for k in ga:
            assert np.array_equal(ga[k], ha[k]), k


# ------------------------------------------------------------------------------ any Sequential depth (modules.rs:28-34)
@pytest.mark.parametrize("n2,emb,common,pl,vl,twists", [
    (9, 64, (128, 64), (), (), True),          # two common layers
    (9, 128, (256,), (32,), (16,), False),     # hidden policy / value layers (config keys policy_layers / value_layers)
    (16, 256, (512,), (), (), True),           # 512 hidden units
    (9, 64, (96,), (), (), False),             # a width the MFMA tiles do not cover
    (4, 32, (), (24,), (), False),             # no common layer at all
    (16, 512, (256, 256, 128), (64, 32), (64,), True),
])
def test_policies_of_any_depth(tw, oracle, n2, emb, common, pl, vl, twists):
    """The reference's Policy holds any Sequential stacks (rust/src/nn/modules.rs:28-34; BasicPolicy builds them from
    common_layers / policy_layers / value_layers, src/twisterl/nn/policy.py:60-113).  Shapes beyond the one-common-layer
    BasicPolicy of the Puzzle configs run the generic engine: forward / predict / full_predict, PPO collect, self-play,
    evaluate and solve are bit-equal to the oracle here too."""
    from tests.util import make_deep_policy_arrays
    side = int(round(n2 ** 0.5))
    arrs = make_deep_policy_arrays(n2, seed=11, emb=emb, common=common, policy_layers=pl, value_layers=vl, scale=2.0)
    op_, ap_ = puzzle_transpose_twist(side) if twists else ((), ())
    gp, op = amd_policy(arrs, op_, ap_), oracle_policy(oracle, arrs, op_, ap_)
    rng = np.random.default_rng(0)
    n = 24
    boards = np.stack([rng.permutation(n2) for _ in range(n)])
    obs = np.arange(n2)[None, :] * n2 + boards
    masks = rng.integers(0, 2, size=(n, 4)).astype(np.uint8); masks[:, 0] = 1
    perms = rng.integers(-1, 2 if twists else 0, size=n).astype(np.int32)
    la, va = gp.evaluate_batch(_lib.TW_EVAL_FORWARD, obs, masks, perms)
    fa, fv = gp.evaluate_batch(_lib.TW_EVAL_FULL_PREDICT, obs, masks)
    oracle.set_det_exp(True)
    try:
        for i in range(n):
            lo, vo = op.forward(obs[i].tolist(), masks[i].astype(bool).tolist(), perm=int(perms[i]), arith=oracle.ARITH_CHAIN)
            assert np.array_equal(f32_bits(lo), f32_bits(la[i])) and f32_bits(vo) == f32_bits(va[i]), i
            po, pv = op.full_predict(obs[i].tolist(), masks[i].astype(bool).tolist(), arith=oracle.ARITH_CHAIN)
            assert np.array_equal(f32_bits(po), f32_bits(fa[i])) and f32_bits(pv) == f32_bits(fv[i]), i
    finally:
        oracle.set_det_exp(False)
    genv, oenv = tw.env.Puzzle(side, side, 5, 2, 256), oracle.Puzzle(side, side, 5, 2, 256)
    for E in (50, 300):
        g = tw.collector.PPOCollector(E, 0.995, 0.995, 32).collect(genv, gp, seed=41)
        o = oracle.ppo_collect(oenv, op, E, 0.995, 0.995, seed=41, arith=oracle.ARITH_CHAIN, det_log=True, num_threads=8)
        _assert_same_collect(g, o, n2)
    z = tw.collector.AZCollector(40, 14, 1.41, 1, 32).collect(genv, gp, seed=43)
    zo = oracle.az_collect(oenv, op, 40, 14, 1.41, 1, seed=43, arith=oracle.ARITH_CHAIN, num_threads=8, det_math=True)
    _assert_same_az(z, zo, n2)
    ge = tw.collector.evaluate(genv, gp, num_episodes=60, deterministic=False, num_searches=3, num_mcts_searches=0, seed=7, C=1.41,
                               max_expand_depth=1, num_cores=32)
    oe = oracle.evaluate(oenv, op, 60, False, 3, seed=7, arith=oracle.ARITH_CHAIN, det_math=True)
    assert f32_bits(ge[0]) == f32_bits(oe[0]) and f32_bits(ge[1]) == f32_bits(oe[1])
    with pytest.raises(RuntimeError, match="f16 modes"):
        tw.collector.PPOCollector(8, 0.9, 0.9, 1, precision="fp16").collect(genv, gp, seed=1)


def test_generic_policy_with_more_episodes_than_lanes_uses_the_queue(tw, oracle):
    """Policies of any depth run 16 episodes per workgroup, two workgroups per CU (their activation buffers are as large as the
    policy's layers, not as the widest policy allowed); with more episodes than that the lanes are persistent and take the next
    episode off the queue (as the MFMA shapes do): bit-equal to the oracle, two workgroups per CU."""
    import twisterl_amd
    from tests.util import make_deep_policy_arrays
    cus = twisterl_amd.device_info()["compute_units"]
    arrs = make_deep_policy_arrays(9, seed=3, emb=32, common=(48, 32), scale=2.0)
    gp, op = amd_policy(arrs), oracle_policy(oracle, arrs)
    genv, oenv = tw.env.Puzzle(3, 3, 4, 2, 256), oracle.Puzzle(3, 3, 4, 2, 256)
    E = cus * 32 + 1500
    g = tw.collector.PPOCollector(E, 0.995, 0.995, 32).collect(genv, gp, seed=29)
    assert (g.stats["rollout_blocks"], g.stats["rollout_threads"]) == (2 * cus, 256)
    o = oracle.ppo_collect(oenv, op, E, 0.995, 0.995, seed=29, arith=oracle.ARITH_CHAIN, det_log=True, num_threads=8)
    _assert_same_collect(g, o, 9)


# ------------------------------------------------------------------------------ any environment
@pytest.mark.parametrize("w,h,steps,emb,common,E", [(3, 3, 8, 64, (64,), 120), (5, 5, 12, 64, (128, 32), 60)])
def test_ppo_collect_of_a_python_environment(tw, oracle, w, h, steps, emb, common, E):
    """PPOCollector.collect on an environment the library does not implement (the reference collects any Box<dyn Env>;
    its example is examples/grid_world): a GridWorld written in Python behind twisterl.env.PyEnv.  The env's code runs on the
    host, the policy forward of every time step is one batched launch; the whole CollectedData is bit-equal to the oracle's
    generic restatement of ppo.rs:41-126 running the same env class.  5 x 5: 625 obs ids (two-byte ids in the result)."""
    from tests.gridworld_env import GridWorld
    from tests.util import make_deep_policy_arrays
    n = w * h
    rng = np.random.default_rng(13)
    lin = lambda i, o, relu: (np.ascontiguousarray(rng.uniform(-2 / np.sqrt(i), 2 / np.sqrt(i), size=(o, i)).astype(np.float32).T).reshape(-1),
                              rng.uniform(-1 / np.sqrt(i), 1 / np.sqrt(i), size=o).astype(np.float32), relu)
    we = rng.uniform(-0.3, 0.3, size=(n * n, emb)).astype(np.float32)
    be = rng.uniform(-0.1, 0.1, size=emb).astype(np.float32)
    cs, width = [], emb
    for hdim in common:
        cs.append(lin(width, hdim, True)); width = hdim
    arrs = (we, be, cs, [lin(width, 4, False)], [lin(width, 1, False)])
    gp, op = amd_policy(arrs), oracle_policy(oracle, arrs)
    proto = GridWorld(w, h, steps)
    env = tw.env.PyEnv(proto)
    env.difficulty = 3
    assert env.num_actions() == 4 and env.obs_shape() == [n, n]
    for merge_order in (True, False):
        g = tw.collector.PPOCollector(E, 0.99, 0.95, 4, merge_order=merge_order).collect(env, gp, seed=77)
        o = oracle.ppo_collect_env(proto, op, E, 0.99, 0.95, seed=77, difficulty=3, merge_order=merge_order)
        _assert_same_collect(g, o, n)
    a = g.to_numpy()
    assert a["obs"].dtype == (np.uint16 if n * n > 256 else np.uint8)
    assert (a["ep_len"] <= steps + 1).all() and a["ep_len"].min() >= 1
    # reference-style access and the consumers' attributes
    assert len(g.obs) == len(g.actions) == len(g.additional_data["rets"]) and set(g.perms) == {-1}
    # self-play of the same environment (tw_az_collect_env: host trees, batched full_predict) against the oracle's restatement
    oracle.set_det_exp(True)
    try:
        for S, med in ((8, 1), (5, 2)):
            z = tw.collector.AZCollector(min(E, 16), S, 1.41, med, 4).collect(env, gp, seed=79)
            zo = oracle.az_collect_env(proto, op, min(E, 16), S, 1.41, med, seed=79, difficulty=3)
            _assert_same_az(z, zo, n)
        # evaluate / solve of the same environment (tw_evaluate_env / tw_solve_env), plain and MCTS-guided
        for det, ns, S in ((True, 1, 0), (False, 3, 0), (False, 2, 4)):
            ge = tw.collector.evaluate(env, gp, num_episodes=12, deterministic=det, num_searches=ns, num_mcts_searches=S, seed=5, C=1.41,
                                       max_expand_depth=1, num_cores=4)
            oe = oracle.evaluate_env(proto, op, 12, det, ns, S, 1.41, 1, seed=5, difficulty=3)
            assert f32_bits(ge[0]) == f32_bits(oe[0]) and f32_bits(ge[1]) == f32_bits(oe[1]), (det, ns, S, ge, oe)
            cur = proto.copy(); cur.seed_episode(3, 1); cur.reset(3)
            (gs, gr), gact = tw.collector.solve(tw.env.PyEnv(cur), gp, det, ns, S, 1.41, 1, seed=9)
            (os_, or_), oact = oracle.solve_env(cur, op, det, ns, S, 1.41, 1, seed=9)
            assert (gs, f32_bits(gr)) == (os_, f32_bits(or_)) and gact == list(oact), (det, ns, S)
    finally:
        oracle.set_det_exp(False)

    class Broken(GridWorld):
        def next(self, action):
            raise ValueError("boom")

        def copy(self):
            c = Broken(self.width, self.height, self.max_steps)
            c.steps_left, c.agent, c.goal, c.trap = self.steps_left, self.agent, self.goal, self.trap
            return c
    with pytest.raises(ValueError, match="boom"):           # an exception in the env's code surfaces, the collect is abandoned
        tw.collector.PPOCollector(4, 0.99, 0.95, 1).collect(tw.env.PyEnv(Broken(w, h, steps)), gp, seed=1)


@pytest.mark.parametrize("w,h,diff,emb,hidden,E,S,med,twists", [
    (3, 3, 3, 64, 128, 70, 24, 1, False),      # Puzzle-8, fewer episodes than one walker workgroup holds
    (3, 3, 4, 64, 128, 300, 40, 2, True),      # max_expand_depth 2, full_predict over two twists (two engine passes per forward)
    (4, 4, 5, 128, 256, 500, 16, 1, False),    # Puzzle-15, many more requests than an engine packs into one forward
    (2, 2, 2, 32, 128, 40, 10, 1, False),
])
def test_split_walker_shape_bit_exact_vs_oracle(tw, oracle, w, h, diff, emb, hidden, E, S, med, twists):
    """The split shape of the walker kernel (TW_OPT_AZ_VARIANT + 512; automatic from eight episodes per CU on): walker waves in a kernel of
    their own -- 72 registers per lane, sixteen or twenty-four per CU --, engine workgroups in another, requests and outputs through
    per-walker mailboxes in device memory.  Same reference semantics (rust/src/collector/az.rs:51-109 over rust/src/rl/search.rs:104-189),
    same bytes as the oracle; a busy engine evaluates fewer look-ahead boards, which never changes a result."""
    n2 = w * h
    gp, op = _pair(oracle, n2, 21, emb, hidden, twists=twists)
    genv, oenv = tw.env.Puzzle(w, h, diff, 2, 256), oracle.Puzzle(w, h, diff, 2, 256)
    o = oracle.az_collect(oenv, op, E, S, 1.41, med, seed=31, arith=oracle.ARITH_CHAIN, num_threads=8, det_math=True)
    with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 512):
        g = tw.collector.AZCollector(E, S, 1.41, med, 32).collect(genv, gp, seed=31)
    assert g.stats["rollout_threads"] in (768, 1024) and g.stats["rollout_blocks"] == -(-E // (g.stats["rollout_threads"] // 64))
    _assert_same_az(g, o, n2)
    assert _lib.debug_counters(14)[13] == 0                      # no wait ran into its watchdog
    with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 512):        # determinism: which engine serves which request when must not matter
        h2 = tw.collector.AZCollector(E, S, 1.41, med, 32).collect(genv, gp, seed=31).to_numpy()
    a = g.to_numpy()
    for k in a:
        assert np.array_equal(a[k], h2[k]), k


def test_split_shape_falls_back_when_its_kernels_cannot_run_side_by_side(tw):
    """The split shape's walker kernel and engine kernel need each other.  Launched one after the other (TW_OPT_AZ_VARIANT + 2048: what
    `rocprofv3 --pmc` does to them) the collect must run into its watchdogs -- seconds, never a hang --, say so on stderr, and come back
    with the bytes of the single-kernel shapes, which the process keeps afterwards."""
    import json
    import os
    import subprocess
    import sys
    import twisterl_amd
    cus = twisterl_amd.device_info()["compute_units"]
    E = 8 * cus + 100                                         # (eight episodes per CU: where the split shape starts)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "tests.tools.split_fallback_worker", str(E)], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "did not run side by side" in r.stderr and r.stderr.count("did not run side by side") == 1       # reported once
    assert out["first"]["digest"] == out["again"]["digest"] == out["pinned_single_kernel"]
    assert out["first"]["launch"][1] in (512, 768) and out["again"]["launch"] == out["first"]["launch"]      # the decoupled shapes inside one workgroup
    # ... while this process, whose kernels do run side by side, takes the split shape for the same batch
    from tests.util import make_policy_arrays
    g = tw.collector.AZCollector(E, 16, 1.41, 1, 1).collect(tw.env.Puzzle(3, 3, 3, 2, 256), amd_policy(make_policy_arrays(9, seed=5, emb=64, hidden=128)), seed=3)
    assert g.stats["rollout_threads"] == 768 and g.stats["rollout_blocks"] == min(-(-E // 12), (cus - cus // 2) * 2)
    import hashlib
    a = g.to_numpy(); h = hashlib.sha256()
    for k in sorted(a):
        h.update(k.encode()); h.update(a[k].tobytes())
    assert h.hexdigest() == out["first"]["digest"]


def test_solve_returns_the_solution_an_environment_tracks_itself(tw, oracle):
    """`Env::track_solution` / `Env::solution` through tw_env_vtable (rust/src/rl/env.rs:61-66): single_solve asks once, before
    the first move, and then returns the environment's own record instead of the actions it played (rust/src/rl/solve.rs:28,
    57-64).  A GridWorld that records 1000 x cell + action: ((success, reward), solution) equal to the oracle's generic solve
    over the same class, for greedy, best-of-N sampled and MCTS-guided solves; the 8-bit entry point refuses such a record; an
    environment that does not track still gets the played actions; evaluate is unaffected."""
    import ctypes as C
    from tests.gridworld_env import GridWorld, TrackingGridWorld
    from tests.util import make_deep_policy_arrays
    from twisterl_amd import collector as col
    w, h, steps = 4, 3, 9
    n = w * h
    arrs = make_deep_policy_arrays(n, seed=21, emb=48, common=(64,), scale=2.0)
    gp, op = amd_policy(arrs), oracle_policy(oracle, arrs)
    oracle.set_det_exp(True)
    try:
        for ep in range(4):
            cur = TrackingGridWorld(w, h, steps); cur.seed_episode(11, ep); cur.reset(4)
            plain = GridWorld(w, h, steps); plain.steps_left, plain.agent, plain.goal, plain.trap = cur.steps_left, cur.agent, cur.goal, cur.trap
            for det, ns, S in ((True, 1, 0), (False, 4, 0), (False, 2, 5)):
                (gs, gr), gsol = tw.collector.solve(tw.env.PyEnv(cur), gp, det, ns, S, 1.41, 1, seed=9)
                (os_, or_), osol = oracle.solve_env(cur, op, det, ns, S, 1.41, 1, seed=9)
                assert (gs, f32_bits(gr)) == (os_, f32_bits(or_)) and gsol == list(osol), (ep, det, ns, S)
                (ps, pr), pact = tw.collector.solve(tw.env.PyEnv(plain), gp, det, ns, S, 1.41, 1, seed=9)
                assert (ps, f32_bits(pr)) == (gs, f32_bits(gr))                         # the same attempts ...
                assert all(a < 4 for a in pact) and [x % 1000 for x in gsol] == pact    # ... reported as played actions / as the env's record
                assert not gsol or max(gsol) >= 1000 or all(x // 1000 == 0 for x in gsol)
            assert cur.path == []                                                       # (the caller's object is not touched)
        # the 8-bit entry point cannot carry this record: TW_ERR_INVALID, never a truncated list
        cur = TrackingGridWorld(w, h, steps); cur.seed_episode(11, 0); cur.reset(4)
        br = col._PyEnvBridge(tw.env.PyEnv(cur), prototype=cur.copy())
        prm = col._solve_params(True, 1, 0, 1.41, 1, 9)
        acts = (C.c_uint8 * 64)(); s_, r_, n_ = C.c_float(), C.c_float(), C.c_uint32()
        rc = _lib.lib().tw_solve_env(C.byref(br.vt), gp._handle(), C.byref(prm), 64, C.byref(s_), C.byref(r_), acts, 64, C.byref(n_))
        assert rc == _lib.TW_ERR_INVALID and b"tw_solve_env32" in _lib.lib().tw_last_error()
        # evaluate runs the same single_solve (evaluate.rs:36-52): numbers equal to the oracle's with either class
        proto = TrackingGridWorld(w, h, steps)
        env = tw.env.PyEnv(proto); env.difficulty = 4
        ge = tw.collector.evaluate(env, gp, num_episodes=10, deterministic=False, num_searches=2, num_mcts_searches=0, seed=5, C=1.41, max_expand_depth=1, num_cores=4)
        oe = oracle.evaluate_env(proto, op, 10, False, 2, 0, 1.41, 1, seed=5, difficulty=4)
        assert f32_bits(ge[0]) == f32_bits(oe[0]) and f32_bits(ge[1]) == f32_bits(oe[1])
        # the other members of the trait the table now carries reach the Python object
        assert env.track_solution() is True and tw.env.PyEnv(GridWorld(w, h, steps)).track_solution() is False
        op_, ap_ = env.twists()
        assert len(op_) == 2 and sorted(op_[1]) == list(range(n * n)) and ap_[1] == [0, 1, 3, 2] and tw.env.PyEnv(GridWorld(w, h, steps)).twists() == ([], [])
        bridge = col._PyEnvBridge(env)
        assert bool(bridge.vt.track_solution) and bool(bridge.vt.solution) and bool(bridge.vt.set_state) and bool(bridge.vt.twists)
        obs_p, act_p = (C.c_int32 * (2 * n * n))(), (C.c_int32 * 8)()
        assert bridge.vt.twists(1, obs_p, act_p, 2) == 2 and list(obs_p[n * n:]) == op_[1] and list(act_p[4:]) == [0, 1, 3, 2]
        st = (C.c_int64 * n)(*([1] + [0] * (n - 3) + [2, 3]))
        bridge.vt.set_state(1, st, n)
        assert proto.agent == (0, 0) and proto.goal == ((n - 2) % w, (n - 2) // w) and proto.trap == ((n - 1) % w, (n - 1) // w)
        plainb = col._PyEnvBridge(tw.env.PyEnv(GridWorld(w, h, steps)))
        assert not bool(plainb.vt.track_solution) and not bool(plainb.vt.solution) and not bool(plainb.vt.twists) and bool(plainb.vt.set_state)
    finally:
        oracle.set_det_exp(False)


def test_errors(tw, oracle):
    gp, _ = _pair(oracle, 9, 8, 32, 32)
    with pytest.raises(RuntimeError, match="No data in collected data chunks to merge"):
        tw.collector.PPOCollector(0, 0.9, 0.9, 1).collect(tw.env.Puzzle(3, 3, 1, 2, 256), gp)
    with pytest.raises(TypeError, match="__extract_env__"):
        tw.collector.PPOCollector(4, 0.9, 0.9, 1).collect(object(), gp)
    with pytest.raises(ValueError):
        tw.collector.PPOCollector(4, 0.9, 0.9, 1).collect(tw.env.Puzzle(5, 5, 1, 2, 256), gp)   # (a 25-cell board collects; this policy is a 9-cell one)
    with pytest.raises(ValueError):
        tw.collector.AZCollector(4, 2, 1.41, 1, 1).collect(tw.env.Puzzle(5, 5, 1, 2, 256), gp)
    with pytest.raises(ValueError):
        tw.collector.PPOCollector(4, 0.9, 0.9, 1).collect(tw.env.Puzzle(4, 4, 1, 2, 256), gp)   # obs_size mismatch
    arrs = make_policy_arrays(9, emb=50, hidden=32)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        amd_policy(arrs)._handle()
    arrs = make_policy_arrays(9, emb=64, hidden=600)
    with pytest.raises(RuntimeError, match="widths up to 512"):
        amd_policy(arrs)._handle()


# ------------------------------------------------------------------------------ AlphaZero / MCTS collect
def _assert_same_az(g, o, n_cells):
    a = g.to_numpy()
    assert set(a) == {"obs", "logits", "perms", "remaining_values", "ep_len", "ep_start"}
    assert np.array_equal(a["ep_len"], o.ep_len)
    assert np.array_equal(a["obs"].astype(np.int64), o.obs)
    assert np.all(a["perms"] == -1)
    assert np.array_equal(f32_bits(a["logits"]), f32_bits(o.logits))              # MCTS probs, bit-exact
    assert np.array_equal(f32_bits(a["remaining_values"]), f32_bits(o.additional_data["remaining_values"]))


@pytest.mark.parametrize("w,h,diff,emb,hidden,E,S,med,twists", [
    (3, 3, 3, 32, 32, 70, 24, 1, False),     # Puzzle-8, tiny net
    (3, 3, 2, 64, 64, 33, 16, 2, True),      # max_expand_depth 2 + full_predict over twists
    (2, 2, 2, 32, 32, 20, 9, 1, False),      # 2x2 board
    (3, 3, 4, 32, 32, 16, 0, 1, False),      # zero searches -> uniform probs (search.rs:184-186)
    (3, 3, 0, 32, 32, 10, 5, 1, False),      # difficulty 0: root is final
    (4, 4, 3, 512, 256, 48, 12, 1, False),   # Puzzle-15 at the benchmark's network size
    (3, 3, 6, 64, 128, 40, 60, 1, False),    # hidden 128: four waves share 32 episodes, one row tile each
    (3, 3, 10, 32, 64, 12, 400, 1, False),   # deep trees: search paths longer than the 8 levels kept in LDS
    # the walker-per-wave kernel (hidden 128 / 256, up to CUs x 16 episodes; tw_mcts_deep.hip):
    (3, 3, 5, 64, 128, 30, 25, 2, True),     #   max_expand_depth 2 + twists, 9-cell boards
    (4, 4, 10, 64, 256, 9, 300, 1, True),    #   trees larger than the LDS table (statistics in the arena), output pool wraps, yields
    (2, 2, 2, 32, 128, 20, 9, 1, False),     #   2x2 board
    (3, 3, 2, 32, 128, 1500, 6, 1, False),   #   more episodes than walkers: the episode queue, arenas reused (32-column engine, hidden 128)
    (3, 3, 0, 32, 128, 10, 5, 1, False),     #   difficulty 0: every root is final
    (3, 3, 4, 32, 256, 16, 0, 1, False),     #   zero searches
    (3, 3, 3, 32, 128, 400, 10, 1, False),   #   one walker per workgroup and the episode queue (up to 2 episodes per CU), 16 columns
    (3, 3, 4, 64, 256, 700, 8, 1, True),     #   two walkers per workgroup on the 32-column engine (2 to 3.5 episodes per CU), 16 columns each
    (3, 3, 4, 64, 256, 1300, 5, 1, True),    #   four walkers per workgroup on the 32-column engine (more than 3.5 episodes per CU), 8 columns each
    (3, 3, 3, 32, 128, 3300, 24, 2, True),   #   eight walkers per workgroup (more than 10 episodes per CU, short searches), 4 columns each
])
def test_az_collect_bit_exact_vs_oracle(tw, oracle, w, h, diff, emb, hidden, E, S, med, twists):
    n2 = w * h
    gp, op = _pair(oracle, n2, 2, emb, hidden, twists=twists)
    genv, oenv = tw.env.Puzzle(w, h, diff, 2, 256), oracle.Puzzle(w, h, diff, 2, 256)
    for merge_order in (True, False):
        coll = tw.collector.AZCollector(num_episodes=E, num_mcts_searches=S, C=1.41, max_expand_depth=med, num_cores=32,
                                        merge_order=merge_order)
        g = coll.collect(genv, gp, seed=31)
        o = oracle.az_collect(oenv, op, E, S, 1.41, med, seed=31, arith=oracle.ARITH_CHAIN, merge_order=merge_order,
                              det_math=True)
        _assert_same_az(g, o, n2)
        assert g.stats["forward_evals"] >= len(o.obs)      # at least the root evaluation of every move


def test_az_output_reuse_below_the_search_path_kept_in_lds(tw, oracle):
    """The lane-per-episode kernel keeps PATH_DEPTH = 8 levels of the search path in LDS; a node whose move takes its parent's
    move back finds its grandparent (whose stored output it reuses) at level plen-3 of that path -- while the path fits.  Below
    8 levels `push` stops recording and level plen-3 is an ancestor further up: the grandparent then comes from the parent
    links.  (Round 2's first form read the path level at any depth -- TW_OPT_AZ_REUSE = 2 -- and so took the output of the wrong
    node in deep trees: the "schedule-dependent" results of docs/HISTORY.md 5.5.)  Deep trees (400 searches, max_expand_depth 2) with
    every way of finding the grandparent: the bytes of the oracle; the diagnostic form counts the disagreements."""
    import bench
    arrs = bench.synthetic_weights(16, seed=0)                   # the benchmark's 512 / 256 policy (as scripts/az_reuse_probe.py)
    gp, op = bench.build_policy(arrs, [], []), oracle.Policy(*arrs, [], [])
    E, S, MED = 70, 400, 2
    genv, oenv = tw.env.Puzzle(4, 4, 8, 2, 256), oracle.Puzzle(4, 4, 8, 2, 256)
    coll = tw.collector.AZCollector(num_episodes=E, num_mcts_searches=S, C=1.41, max_expand_depth=MED, num_cores=32)
    o = oracle.az_collect(oenv, op, E, S, 1.41, MED, seed=7, arith=oracle.ARITH_CHAIN, det_math=True, num_threads=8)
    with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 2):
        for mode in (0, 1, 3, 4):
            with _lib.launch_option(_lib.TW_OPT_AZ_REUSE, mode):
                g = coll.collect(genv, gp, seed=7)
                cnt = _lib.debug_counters(13)
            _assert_same_az(g, o, 16)
            assert (g.stats["reused_evals"] == 0) == (mode == 1)
            assert cnt[12] == 0
            if mode == 4:
                # candidates; below PATH_DEPTH; there -- and only there -- the path level is not the grandparent
                assert cnt[3] == g.stats["reused_evals"] > 0 and cnt[4] > 0 and cnt[5] == 0
                assert cnt[6] == 0 and cnt[7] > 0 and cnt[8] == 0 and cnt[10] == 0 and cnt[11] == 0


def test_az_reference_style_consumer(tw, oracle):
    """AZ.data_to_torch reads obs, logits (= MCTS probs) and additional_data['remaining_values']
    (src/twisterl/rl/az.py:30-34); values / rewards / actions stay empty (az.rs:97-104)."""
    gp, _ = _pair(oracle, 9, 3, 32, 32)
    data = tw.collector.AZCollector(8, 10, 1.41, 1, 32).collect(tw.env.Puzzle(3, 3, 3, 2, 256), gp)
    obs, probs, vals = data.obs, data.logits, data.additional_data["remaining_values"]
    assert len(obs) == len(probs) == len(vals) > 0
    assert data.values == [] and data.rewards == [] and data.actions == [] and set(data.perms) == {-1}
    assert all(abs(sum(p) - 1.0) < 1e-5 for p in probs)


def test_az_sharding_invariance_and_single_rank_gather(tw, oracle):
    """Two AZ shards (episode_offset, compact order) concatenate to the unsharded collect; collect_sharded on one rank
    returns the same tensors (the multi-GPU path of INTEGRATION.md §D for the AlphaZero collector)."""
    import torch
    import torch.distributed as dist
    from twisterl_amd.dist import collect_sharded
    gp, _ = _pair(oracle, 9, 5, 64, 128)
    env = tw.env.Puzzle(3, 3, 4, 2, 256)
    mk = lambda n, off=0: tw.collector.AZCollector(n, 12, 1.41, 1, 1, merge_order=False, episode_offset=off)
    full = mk(90).collect(env, gp, seed=5).to_numpy()
    a, b = mk(40).collect(env, gp, seed=5).to_numpy(), mk(50, 40).collect(env, gp, seed=5).to_numpy()
    keys = [k for k in full if k not in ("ep_start",) and len(full[k])]
    assert "obs" in keys and "logits" in keys and "ep_len" in keys
    for k in keys:
        assert np.array_equal(full[k], np.concatenate([a[k], b[k]])), k
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29547", rank=0, world_size=1)
    try:
        merged, _ = collect_sharded(tw.collector.AZCollector(90, 12, 1.41, 1, 1), env, gp, seed=5)
        ref = tw.collector.AZCollector(90, 12, 1.41, 1, 1).collect(env, gp, seed=5).to_torch()     # reference merge order
        for k, v in merged.items():
            assert torch.equal(v.cpu(), ref[k].cpu()), k
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------ Conv1dPolicy
def _conv_pair(tw, oracle, n, conv_dim, v, hidden, seed):
    """Conv1dPolicy.to_rust() (src/twisterl/nn/policy.py:258-266, nn/utils.py:68-75): EmbeddingBag(conv_w.squeeze(2).T,
    zeros, True, obs_shape, conv_dim) with obs_shape = [n_cells, n_cells] (cell, tile)."""
    rng = np.random.default_rng(seed)
    n2 = n * n
    emb = n2 * v
    arrs = make_policy_arrays(n2, seed=seed, emb=emb, hidden=hidden)
    conv_w = rng.uniform(-0.3, 0.3, size=(v, n2)).astype(np.float32)           # Conv1d.weight.squeeze(2): [out][in]
    vec, bias = np.ascontiguousarray(conv_w.T), np.zeros(emb, dtype=np.float32)
    _, _, common, action, value = arrs
    seq = lambda ls: tw.nn.Sequential([tw.nn.Linear(w.tolist(), b.tolist(), r) for (w, b, r) in ls])
    gp = tw.nn.Policy(tw.nn.EmbeddingBag(vec.tolist(), bias.tolist(), True, [n2, n2], conv_dim), seq(common), seq(action), seq(value), [], [])
    op = oracle.Policy(vec, bias, common, action, value, (), (), obs_shape=[n2, n2], conv_dim=conv_dim)
    return gp, op, conv_w, arrs


@pytest.mark.parametrize("n,conv_dim,v,hidden,E", [(3, 0, 32, 64, 150), (3, 1, 32, 256, 70), (2, 0, 8, 32, 40)])
def test_conv1d_policy_collect_bit_exact_vs_oracle(tw, oracle, n, conv_dim, v, hidden, E):
    """The conv1d mode of EmbeddingBag (rust/src/nn/layers.rs:63-77) behind Conv1dPolicy: PPO collect and the plain
    forward are bit-equal to the oracle's restatement of that mode."""
    gp, op, _, _ = _conv_pair(tw, oracle, n, conv_dim, v, hidden, 11)
    genv, oenv = tw.env.Puzzle(n, n, 6, 2, 256), oracle.Puzzle(n, n, 6, 2, 256)
    g = tw.collector.PPOCollector(E, 0.99, 0.95, 1).collect(genv, gp, seed=21)
    o = oracle.ppo_collect(oenv, op, E, 0.99, 0.95, seed=21, arith=oracle.ARITH_CHAIN, det_log=True)
    _assert_same_collect(g, o, n * n)
    from twisterl_amd import _lib
    n2 = n * n
    rng = np.random.default_rng(1)
    obs = np.arange(n2)[None, :] * n2 + np.stack([rng.permutation(n2) for _ in range(8)])
    masks = np.ones((8, 4), dtype=np.uint8)
    la, va = gp.evaluate_batch(_lib.TW_EVAL_FORWARD, obs, masks, np.full(8, -1, dtype=np.int32))
    for i in range(8):
        lo, vo = op.forward(obs[i].tolist(), masks[i].tolist(), perm=-1, arith=oracle.ARITH_CHAIN)
        assert np.array_equal(f32_bits(la[i]), f32_bits(lo)) and f32_bits(va[i]) == f32_bits(vo)


def test_conv1d_policy_update_from_torch(tw, oracle):
    """Policy.update_from_torch for the Conv1dPolicy layout (conv_layer.weight): same result as rebuilding the policy."""
    import torch
    n, conv_dim, v, hidden = 3, 1, 32, 64
    gp_old, _, _, _ = _conv_pair(tw, oracle, n, conv_dim, v, hidden, 3)
    gp_new, _, conv_w, arrs = _conv_pair(tw, oracle, n, conv_dim, v, hidden, 4)
    _, _, [(w1, b1, _)], [(wa, ba, _)], [(wv, bv, _)] = arrs
    emb = n * n * v
    state = {"conv_layer.weight": torch.tensor(conv_w).unsqueeze(2).cuda(),
             "common.0.weight": torch.tensor(w1.reshape(emb, hidden).T.copy()).cuda(), "common.0.bias": torch.tensor(b1).cuda(),
             "action.0.weight": torch.tensor(wa.reshape(hidden, 4).T.copy()).cuda(), "action.0.bias": torch.tensor(ba).cuda(),
             "value.0.weight": torch.tensor(wv.reshape(hidden, 1).T.copy()).cuda(), "value.0.bias": torch.tensor(bv).cuda()}
    env = tw.env.Puzzle(n, n, 5, 2, 256)
    want = tw.collector.PPOCollector(100, 0.99, 0.95, 1).collect(env, gp_new, seed=8).to_numpy()
    gp_old.update_from_torch(state)
    got = tw.collector.PPOCollector(100, 0.99, 0.95, 1).collect(env, gp_old, seed=8).to_numpy()
    for k in ("obs", "logits", "values", "actions", "advs", "rets"):
        assert np.array_equal(got[k], want[k]), k


# ------------------------------------------------------------------------------ evaluate / solve
@pytest.mark.parametrize("w,diff,emb,hidden,twists", [(3, 4, 32, 32, False), (3, 6, 64, 64, True), (4, 5, 512, 256, False)])
def test_evaluate_and_solve_match_oracle(tw, oracle, w, diff, emb, hidden, twists):
    """collector.evaluate / collector.solve (python_interface/env.rs:180-207 over rl/evaluate.rs, rl/solve.rs):
    same (success_rate, mean_reward) and the same best action list as the oracle, bit for bit."""
    n2 = w * w
    gp, op = _pair(oracle, n2, 4, emb, hidden, twists=twists, scale=3.0)
    genv, oenv = tw.env.Puzzle(w, w, diff, 2, 256), oracle.Puzzle(w, w, diff, 2, 256)
    for det, ns in ((True, 1), (False, 1), (False, 5)):
        g = tw.collector.evaluate(genv, gp, num_episodes=100, deterministic=det, num_searches=ns, num_mcts_searches=0,
                                  seed=7, C=1.41, max_expand_depth=1, num_cores=32)
        o = oracle.evaluate(oenv, op, 100, det, ns, seed=7, arith=oracle.ARITH_CHAIN, det_math=True)
        assert f32_bits(g[0]) == f32_bits(o[0]) and f32_bits(g[1]) == f32_bits(o[1]), (det, ns, g, o)
    # solve from a given state (Algorithm.solve: env.set_state(state) then solve, rl/algorithm.py:226-246)
    rng = np.random.default_rng(1)
    start = oracle.Puzzle(w, w, diff, 2, 256); start.reset(seed=9, episode=3)
    state = start.get_state()
    genv.set_state(state); oenv.set_state(state)
    before = genv.get_state()
    for det, ns in ((True, 1), (False, 8)):
        (gs, gr), gact = tw.collector.solve(genv, gp, det, ns, 0, 1.41, 1, seed=5)
        (os_, or_), oact = oracle.solve(oenv, op, det, ns, seed=5, arith=oracle.ARITH_CHAIN, det_math=True)
        assert (gs, f32_bits(gr)) == (os_, f32_bits(or_)) and gact == oact
    assert genv.get_state() == before                        # the env passed in is not mutated (solve.rs:85)


def test_reference_trained_policy_on_the_gpu(tw, oracle):
    """The reference's trained Puzzle-8 checkpoint (tests/golden/ppo_puzzle8_v1_weights.npz) through collector.evaluate /
    collector.solve / PPOCollector on the GPU: bit-equal to the oracle, and it SOLVES the puzzle (success >= 0.97 at the
    config's diff_max 32) -- obs encoding, weight layout and action semantics pinned end to end on the device as well."""
    from tests.util import trained_puzzle8_arrays
    arrs = trained_puzzle8_arrays()
    gp, op = amd_policy(arrs), oracle_policy(oracle, arrs)
    for diff in (8, 32):
        genv, oenv = tw.env.Puzzle(3, 3, diff, 2, 256), oracle.Puzzle(3, 3, diff, 2, 256)
        for det, ns in ((False, 1), (True, 1), (False, 10)):            # the config's evals: ppo_1, ppo_10 (+ deterministic)
            g = tw.collector.evaluate(genv, gp, num_episodes=200, deterministic=det, num_searches=ns, num_mcts_searches=0,
                                      seed=1, C=1.41, max_expand_depth=1, num_cores=32)
            o = oracle.evaluate(oenv, op, 200, det, ns, seed=1, arith=oracle.ARITH_CHAIN, det_math=True)
            assert f32_bits(g[0]) == f32_bits(o[0]) and f32_bits(g[1]) == f32_bits(o[1]), (diff, det, ns, g, o)
            assert g[0] >= 0.97, (diff, det, ns, g)
    # a scrambled board, solved greedily on the device: the action list replays to the solved board on the oracle's Puzzle
    start = oracle.Puzzle(3, 3, 32, 2, 256); start.reset(seed=11, episode=0)
    state = start.get_state()
    genv.set_state(state)
    (gs, gr), acts = tw.collector.solve(genv, gp, True, 1, 0, 1.41, 1, seed=5)
    assert gs == 1.0 and len(acts) > 0
    rp = oracle.Puzzle(3, 3, 32, 2, 256); rp.set_state(state)
    _, _, _, fin, boards = oracle.replay(rp, acts)
    assert boards[-1].tolist() == list(range(9)) and fin[-1]
    # trained policies end their episodes early: a ragged PPO collect, bit-exact, mostly solved
    E = 3000
    g = tw.collector.PPOCollector(E, 0.995, 0.995, 32).collect(tw.env.Puzzle(3, 3, 32, 2, 256), gp, seed=77)
    o = oracle.ppo_collect(oracle.Puzzle(3, 3, 32, 2, 256), op, E, 0.995, 0.995, seed=77, arith=oracle.ARITH_CHAIN, det_log=True, num_threads=8)
    _assert_same_collect(g, o, 9)
    a = g.to_numpy()
    last = a["ep_start"].astype(np.int64) + a["ep_len"].astype(np.int64) - 1
    assert (a["rewards"][last] == 1.0).mean() >= 0.95 and a["ep_len"].mean() < 40


@pytest.mark.parametrize("w,diff,emb,hidden,twists,S,med", [(3, 3, 32, 32, False, 12, 1), (3, 4, 64, 64, True, 8, 2), (4, 3, 512, 256, True, 6, 1)])
def test_mcts_guided_evaluate_and_solve_match_oracle(tw, oracle, w, diff, emb, hidden, twists, S, med):
    """num_mcts_searches > 0 (solve.rs:41-47): the action distribution comes from predict_probs_mcts; same numbers and the
    same action list as the oracle, bit for bit (deterministic exp in the spec)."""
    n2 = w * w
    gp, op = _pair(oracle, n2, 6, emb, hidden, twists=twists, scale=3.0)
    genv, oenv = tw.env.Puzzle(w, w, diff, 2, 256), oracle.Puzzle(w, w, diff, 2, 256)
    for det, ns in ((True, 1), (False, 3)):
        g = tw.collector.evaluate(genv, gp, num_episodes=40, deterministic=det, num_searches=ns, num_mcts_searches=S,
                                  seed=3, C=1.41, max_expand_depth=med, num_cores=32)
        o = oracle.evaluate(oenv, op, 40, det, ns, num_mcts_searches=S, seed=3, Cc=1.41, max_expand_depth=med,
                            arith=oracle.ARITH_CHAIN, det_math=True)
        assert f32_bits(g[0]) == f32_bits(o[0]) and f32_bits(g[1]) == f32_bits(o[1]), (det, ns, g, o)
    start = oracle.Puzzle(w, w, diff, 2, 256); start.reset(seed=2, episode=1)
    state = start.get_state()
    genv.set_state(state); oenv.set_state(state)
    for det, ns in ((True, 1), (False, 4)):
        (gs, gr), gact = tw.collector.solve(genv, gp, det, ns, S, 1.41, med, seed=5)
        (os_, or_), oact = oracle.solve(oenv, op, det, ns, num_mcts_searches=S, Cc=1.41, max_expand_depth=med, seed=5,
                                        arith=oracle.ARITH_CHAIN, det_math=True)
        assert (gs, f32_bits(gr)) == (os_, f32_bits(or_)) and gact == oact


def test_mcts_guided_evaluate_on_the_walker_kernel(tw, oracle):
    """MCTS-guided evaluate / solve of the MFMA policy shapes run the walker kernel in its solve mode (tw_mcts_deep.hip, SOLVE: one walker
    per attempt, persistent, the episode queue; the reference's `mcts_100` evaluation: 35.7 -> 6.0 ms): one, two and four walkers per
    workgroup with rounds through the queue, attempts that start in a final state (difficulty 1: a scramble move into the wall leaves the
    board solved), greedy and sampled with several attempts per episode, both expansion depths -- the oracle's numbers, and the same as
    the lane-per-episode kernel (TW_OPT_AZ_VARIANT 2); solve() from a state with its action list."""
    import twisterl_amd
    cus = twisterl_amd.device_info()["compute_units"]
    gp, op = _pair(oracle, 9, 6, 64, 128, twists=True, scale=3.0)
    for diff, n_ep, det, ns, S, med in ((1, 40, True, 1, 5, 1), (3, cus + 9, False, 2, 4, 1), (2, 2 * cus + 5, False, 3, 4, 2), (3, 3 * cus + 1, True, 5, 3, 1)):
        genv, oenv = tw.env.Puzzle(3, 3, diff, 2, 256), oracle.Puzzle(3, 3, diff, 2, 256)
        kw = dict(num_episodes=n_ep, deterministic=det, num_searches=ns, num_mcts_searches=S, seed=3, C=1.41, max_expand_depth=med, num_cores=32)
        g = tw.collector.evaluate(genv, gp, **kw)
        with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 2):
            l = tw.collector.evaluate(genv, gp, **kw)
        assert f32_bits(g[0]) == f32_bits(l[0]) and f32_bits(g[1]) == f32_bits(l[1]), (diff, n_ep, g, l)
        if n_ep <= 2 * cus + 5:
            o = oracle.evaluate(oenv, op, n_ep, det, ns, num_mcts_searches=S, seed=3, Cc=1.41, max_expand_depth=med, arith=oracle.ARITH_CHAIN, det_math=True)
            assert f32_bits(g[0]) == f32_bits(o[0]) and f32_bits(g[1]) == f32_bits(o[1]), (diff, n_ep, g, o)
    genv, oenv = tw.env.Puzzle(3, 3, 4, 2, 256), oracle.Puzzle(3, 3, 4, 2, 256)
    start = oracle.Puzzle(3, 3, 4, 2, 256); start.reset(seed=2, episode=5)
    state = start.get_state()
    genv.set_state(state); oenv.set_state(state)
    for det, ns, S in ((True, 1, 8), (False, 6, 5)):
        (gs, gr), gact = tw.collector.solve(genv, gp, det, ns, S, 1.41, 1, seed=5)
        (os_, or_), oact = oracle.solve(oenv, op, det, ns, num_mcts_searches=S, Cc=1.41, max_expand_depth=1, seed=5, arith=oracle.ARITH_CHAIN, det_math=True)
        assert (gs, f32_bits(gr)) == (os_, f32_bits(or_)) and gact == oact
    assert genv.get_state() == state


def test_walker_self_play_uses_as_few_walkers_per_workgroup_as_keep_the_chip_busy(tw, oracle):
    """tw_mcts_deep.hip, short searches: one walker x 16 columns per workgroup up to 2 episodes per CU, two x 8 up to 4.5, four x 4
    beyond, and beyond ten eight x 2 in workgroups of eight waves (from 400 searches on: one up to 3, two below 8) -- at most
    one workgroup per CU, the rest of the episodes comes off the queue.  The parity cases of test_az_collect_bit_exact_vs_oracle
    run all of them; this pins the launches the workgroup shape tells apart, and every pinned shape (TW_OPT_AZ_VARIANT: 4 / 3 /
    5 / 6 = one / two / four / eight walkers, + 16 / + 32 = 16- / 32-column engine, + 128 / + 256 = decoupled engine waves on / off)
    gives the same bytes as the automatic choice."""
    import twisterl_amd
    cus = twisterl_amd.device_info()["compute_units"]
    gp, _ = _pair(oracle, 9, 2, 32, 128)
    env = tw.env.Puzzle(3, 3, 2, 2, 256)
    # (from two walkers on, the decoupled shape: four engine-only waves + the walkers = 64 x (4 + walkers) threads; from eight episodes per CU
    #  on the SPLIT shape: half of the CUs run engine workgroups (a kernel of their own), the others two workgroups of twelve walker waves)
    for E, want in ((cus // 2, (cus // 2, 256)), (cus, (cus, 256)), (2 * cus, (cus, 256)), (3 * cus, (cus, 384)), (6 * cus, (cus, 512)), (13 * cus, (cus, 768))):
        d = tw.collector.AZCollector(E, 4 if E <= 8 * cus else 24, 1.41, 1, 1).collect(env, gp, seed=3)     # (very short searches: walker kernel up to 8 episodes per CU)
        if E >= 8 * cus:
            want = (min(-(-E // 12), (cus - cus // 2) * 2), 768)
        assert (d.stats["rollout_blocks"], d.stats["rollout_threads"]) == want, (E, d.stats["rollout_blocks"], d.stats["rollout_threads"])
    # outputs taken from a grandparent / the chosen child (same board) are part of forward_evals, and counted in reused_evals
    # (walker kernel; the lane-per-episode kernel -- hidden 32 here -- takes the grandparent's only)
    d = tw.collector.AZCollector(cus, 40, 1.41, 1, 1).collect(tw.env.Puzzle(3, 3, 6, 2, 256), gp, seed=3)
    assert 0 < d.stats["reused_evals"] < d.stats["forward_evals"] and d.stats["reused_evals"] > d.stats["forward_evals"] // 10
    gp32, _ = _pair(oracle, 9, 2, 32, 32)
    d = tw.collector.AZCollector(64, 40, 1.41, 1, 1).collect(tw.env.Puzzle(3, 3, 6, 2, 256), gp32, seed=3)
    assert 0 < d.stats["reused_evals"] < d.stats["forward_evals"]
    E = 3 * cus
    auto = tw.collector.AZCollector(E, 6, 1.41, 1, 1).collect(env, gp, seed=4).to_numpy()
    from twisterl_amd import _lib
    # ... + 128 / + 256: the decoupled shape (engine-only waves beside the walkers) pinned on / off
    # ... + 512 / + 1024: the split shape (walkers and engine as two kernels, mailboxes in device memory) pinned on / off
    for variant in (16 + 4, 16 + 3, 16 + 5, 16 + 6, 32 + 4, 32 + 3, 32 + 5, 32 + 6, 256 + 16 + 3, 256 + 16 + 5, 256 + 16 + 6, 128 + 16 + 3, 128 + 16 + 5, 128 + 16 + 6, 256, 512, 1024):
        with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, variant):
            pinned = tw.collector.AZCollector(E, 6, 1.41, 1, 1).collect(env, gp, seed=4).to_numpy()
        for k in auto:
            assert np.array_equal(auto[k], pinned[k]), (variant, k)


def test_episode_order_is_a_stable_sort_by_manhattan_distance(tw, oracle):
    """launch_episode_order (tw_rollout.hip) through the test hook tw_debug_episode_order: the start boards are the ones the
    collectors play (first record of every self-play episode), the order is a permutation sorted by decreasing sum of the tiles'
    Manhattan distances to their places, ties in index order -- for ragged sizes around the kernel's 16 x 64-lane spans."""
    gp, _ = _pair(oracle, 16, 2, 32, 128)
    for (w, h, diff), n in (((4, 4, 8), 4096), ((4, 4, 8), 1), ((4, 4, 8), 63), ((4, 4, 8), 1025), ((3, 3, 5), 777), ((4, 3, 30), 20_000), ((2, 2, 3), 130)):
        env = tw.env.Puzzle(w, h, diff, 2, 256)
        boards, order = _lib.debug_episode_order(env._desc(), 41, 5, n)
        assert sorted(order.tolist()) == list(range(n))
        cells = w * h
        tiles = np.stack([(boards >> np.uint64(4 * i)) & np.uint64(15) for i in range(cells)], axis=1).astype(np.int64)     # [n][cell]
        assert np.all(np.sort(tiles, axis=1) == np.arange(cells))
        px = np.arange(cells) % w; py = np.arange(cells) // w     # the solved board holds tile v at cell v (the blank, 0, at cell 0)
        ci = np.arange(cells)
        d = (np.abs(ci % w - px[tiles]) + np.abs(ci // w - py[tiles])) * (tiles != 0)
        key = np.minimum(d.sum(axis=1), 63)
        want = np.argsort(-key, kind="stable")
        assert np.array_equal(order.astype(np.int64), want), (w, h, n)
        if cells == 16 and n == 4096:                               # the boards are the collectors' start boards
            g = tw.collector.AZCollector(num_episodes=300, num_mcts_searches=2, C=1.41, max_expand_depth=1, num_cores=1, seed=41, episode_offset=5,
                                         merge_order=False).collect(env, gp, seed=41).to_numpy()
            first = g["obs"][g["ep_start"].astype(np.int64)].astype(np.int64)
            assert np.array_equal(first, tiles[:300] + 16 * ci)


def test_self_play_takes_the_longest_looking_episodes_first_and_gives_the_same_bytes(tw, oracle):
    """The walker self-play kernel hands the episodes to its walkers by decreasing distance of the start board from the solved one
    (launch_episode_order, tw_rollout.hip) -- a schedule, not a result: every episode is keyed by its own index.  Several rounds
    through the queue, a difficulty that leaves episodes of 1 .. depth-limit moves, one / four / eight walkers per workgroup:
    bit-identical to the oracle and to the order by index (TW_OPT_AZ_VARIANT + 64)."""
    import twisterl_amd
    cus = twisterl_amd.device_info()["compute_units"]
    gp, op = _pair(oracle, 9, 11, 64, 128, twists=True)
    genv, oenv = tw.env.Puzzle(3, 3, 4, 2, 256), oracle.Puzzle(3, 3, 4, 2, 256)
    for E, S, pin in ((5 * cus + 7, 12, 0), (2 * cus - 3, 12, 0), (3 * cus + 1, 12, 6)):
        coll = tw.collector.AZCollector(num_episodes=E, num_mcts_searches=S, C=1.41, max_expand_depth=1, num_cores=32, seed=31, merge_order=False)
        with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, pin):
            g = coll.collect(genv, gp, seed=31).to_numpy()
        with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, pin + 64):
            h = coll.collect(genv, gp, seed=31).to_numpy()
        for k in g:
            assert np.array_equal(g[k], h[k]), (E, k)
        assert g["ep_len"].min() < g["ep_len"].max()                  # ragged: the order matters for the schedule
        o = oracle.az_collect(oenv, op, E, S, 1.41, 1, seed=31, arith=oracle.ARITH_CHAIN, num_threads=16, merge_order=False, det_math=True)
        assert np.array_equal(g["ep_len"], o.ep_len) and np.array_equal(g["obs"].astype(np.int64), o.obs)
        assert np.array_equal(f32_bits(g["logits"]), f32_bits(o.logits))
        assert np.array_equal(f32_bits(g["remaining_values"]), f32_bits(o.additional_data["remaining_values"]))


# ------------------------------------------------------------------------------ full-size properties
def test_full_size_properties_puzzle8_65k(tw, oracle):
    """BASELINE config 2 size (65,536 envs): determinism, replay parity on a sample, GAE parity on
    a sample, record-count identities."""
    gp, op = _pair(oracle, 9, 0, 512, 256)
    env = tw.env.Puzzle(3, 3, 32, 2, 256)
    coll = tw.collector.PPOCollector(65536, 0.995, 0.995, 32, merge_order=False)
    a = coll.collect(env, gp, seed=21).to_numpy()
    b = coll.collect(env, gp, seed=21).to_numpy()
    for k in a:
        assert np.array_equal(a[k], b[k]), k                       # same seed twice -> identical buffers
    L = a["ep_len"].astype(np.int64)
    assert L.sum() == a["obs"].shape[0] and L.min() >= 1 and L.max() <= 65
    starts = np.concatenate([[0], np.cumsum(L)])
    assert np.array_equal(a["ep_start"].astype(np.int64), starts[:-1])
    rng = np.random.default_rng(0)
    for e in rng.choice(65536, size=48, replace=False):
        s, n = int(starts[e]), int(L[e])
        p = oracle.Puzzle(3, 3, 32, 2, 256)
        p.reset(seed=21, episode=int(e))
        obs, masks, rew, fin, _ = oracle.replay(p, a["actions"][s:s + n - 1].astype(np.int64))
        assert np.array_equal(obs, a["obs"][s:s + n].astype(np.int64))
        assert np.array_equal(f32_bits(rew), f32_bits(a["rewards"][s:s + n]))
        assert fin[-1] and not fin[:-1].any()
        assert np.all((a["logits"][s:s + n] == np.float32(-1e10)) == ~masks)
        advs, rets = oracle.gae(a["rewards"][s:s + n], a["values"][s:s + n], 0.995, 0.995)
        assert np.array_equal(f32_bits(advs), f32_bits(a["advs"][s:s + n]))
        assert np.array_equal(f32_bits(rets), f32_bits(a["rets"][s:s + n]))
        for t in (0, n - 1):
            lo, vo = op.forward(obs[t].tolist(), masks[t].tolist(), arith=oracle.ARITH_CHAIN)
            assert np.array_equal(f32_bits(lo), f32_bits(a["logits"][s + t])) and f32_bits(vo) == f32_bits(a["values"][s + t])
    # merge order = rotation of the index order by one episode (collector.rs:40-46)
    m = tw.collector.PPOCollector(65536, 0.995, 0.995, 32, merge_order=True).collect(env, gp, seed=21).to_numpy()
    tail = int(L[-1])
    for k in ("obs", "logits", "values", "actions", "advs", "rets"):
        assert np.array_equal(m[k][:tail], a[k][-tail:]) and np.array_equal(m[k][tail:], a[k][:-tail]), k


# ------------------------------------------------------------------------------ f16-input MFMA mode
def _check_f16_collect(oracle, a, op, w, h, diff, seed, n_perms, episodes, gamma=0.995, lam=0.995, arith=None, atol=1e-4):
    """Per-record parity of a precision="fp16" collect (merge_order=False) with the oracle, independent of
    sampling: env transitions / obs / masks / rewards / twist draws BIT-EXACT by replaying the GPU's
    actions; logits and values within 1e-4 of the oracle's ARITH_F16 forward on the same record; the
    action BIT-EXACT = the oracle's Gumbel-max on the GPU's own logits with the spec's uniforms; GAE
    bit-exact on the GPU's own values.  Returns the largest logit/value deviation seen."""
    n2 = w * h
    L = a["ep_len"].astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(L)])
    worst = 0.0
    for e in episodes:
        s, n = int(starts[e]), int(L[e])
        p = oracle.Puzzle(w, h, diff, 2, 256)
        p.reset(seed=seed, episode=int(e))
        obs, masks, rew, fin, _ = oracle.replay(p, a["actions"][s:s + n - 1].astype(np.int64))
        assert np.array_equal(obs, a["obs"][s:s + n].astype(np.int64))
        assert np.array_equal(f32_bits(rew), f32_bits(a["rewards"][s:s + n]))
        assert fin[-1] and not fin[:-1].any()
        assert np.all((a["logits"][s:s + n] == np.float32(-1e10)) == ~masks.astype(bool))
        advs, rets = oracle.gae(a["rewards"][s:s + n], a["values"][s:s + n], gamma, lam)
        assert np.array_equal(f32_bits(advs), f32_bits(a["advs"][s:s + n]))
        assert np.array_equal(f32_bits(rets), f32_bits(a["rets"][s:s + n]))
        for t in range(n):
            perm = -1
            if n_perms:
                perm = (oracle.philox4x32_10([e & 0xFFFFFFFF, e >> 32, t, 2], [seed & 0xFFFFFFFF, seed >> 32])[0] * n_perms) >> 32
            assert int(a["perms"][s + t]) == perm
            lo, vo = op.forward(obs[t].tolist(), masks[t].tolist(), perm=perm, arith=oracle.ARITH_F16 if arith is None else arith)
            lg = a["logits"][s + t]
            # typical deviation 1e-6 (f32 accumulation order inside the MFMA); when that last-bit difference
            # flips the f16 rounding of one hidden activation (2^-11 relative) a logit moves by up to ~3e-5
            np.testing.assert_allclose(lg, lo, atol=atol, rtol=1e-5)
            assert abs(float(a["values"][s + t]) - vo) <= atol * max(1.0, abs(vo))
            worst = max(worst, float(np.max(np.abs(lg - np.asarray(lo, np.float32)))), abs(float(a["values"][s + t]) - vo))
            u = [(x >> 8) / 16777216.0 for x in oracle.philox4x32_10([e & 0xFFFFFFFF, e >> 32, t, 1], [seed & 0xFFFFFFFF, seed >> 32])]
            assert int(a["actions"][s + t]) == oracle.sample_from_logits(lg, u, det_log=True)
    return worst


@pytest.mark.parametrize("w,h,diff,emb,hidden,E,twists", [
    (3, 3, 5, 32, 32, 300, False),      # Puzzle-8, one stage, ragged workgroup tail
    (3, 3, 12, 64, 64, 129, True),      # two stages, twists
    (2, 2, 3, 96, 32, 64, False),       # three stages (ring wrap), 2x2 board
    (3, 2, 4, 128, 128, 70, False),     # non-square board (6 cells padded to 9 chunks)
    (4, 4, 6, 512, 256, 256, True),     # Puzzle-15 at the benchmark's network size, with twists
])
def test_ppo_collect_f16_mode(tw, oracle, w, h, diff, emb, hidden, E, twists):
    """precision="fp16" (BASELINE config 2: 'MLP policy fp16, bit-exact state check vs CPU')."""
    n2 = w * h
    if twists and w != h:
        pytest.skip("transpose twist needs a square board")
    gp, op = _pair(oracle, n2, 1, emb, hidden, twists=twists)
    genv = tw.env.Puzzle(w, h, diff, 2, 256)
    oenv = oracle.Puzzle(w, h, diff, 2, 256)
    coll = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32},
                                     seed=13, merge_order=False, precision="fp16")
    a = coll.collect(genv, gp, seed=13).to_numpy()
    b = coll.collect(genv, gp, seed=13).to_numpy()
    for k in a:
        assert np.array_equal(a[k], b[k]), k                       # deterministic
    worst = _check_f16_collect(oracle, a, op, w, h, diff, 13, 2 if twists else 0, range(E))
    # whole-collect comparison with the oracle running the same spec: identical wherever no Gumbel near-tie
    # flipped an action (the f32 accumulation order inside the MFMA is the only difference)
    o = oracle.ppo_collect(oenv, op, E, 0.995, 0.995, seed=13, arith=oracle.ARITH_F16, det_log=True, merge_order=False)
    same = 0
    L = a["ep_len"].astype(np.int64); starts = np.concatenate([[0], np.cumsum(L)])
    Lo = o.ep_len.astype(np.int64); so = np.concatenate([[0], np.cumsum(Lo)])
    for e in range(E):
        if L[e] == Lo[e] and np.array_equal(a["actions"][starts[e]:starts[e + 1]].astype(np.int64), o.actions[so[e]:so[e + 1]]):
            same += 1
            np.testing.assert_allclose(a["rets"][starts[e]:starts[e + 1]], o.additional_data["rets"][so[e]:so[e + 1]], atol=1e-5, rtol=1e-5)
            np.testing.assert_allclose(a["advs"][starts[e]:starts[e + 1]], o.additional_data["advs"][so[e]:so[e + 1]], atol=2e-5, rtol=1e-5)
    assert same >= 0.98 * E, (same, E, worst)


def test_f16_mode_errors_and_full_size_sample(tw, oracle):
    """65,536 Puzzle-8 envs in the f16 mode (BASELINE config 2): record-count identities and the per-record
    replay parity on a sample; unsupported shapes fail loudly."""
    gp, op = _pair(oracle, 9, 0, 512, 256)
    env = tw.env.Puzzle(3, 3, 32, 2, 256)
    a = tw.collector.PPOCollector(65536, 0.995, 0.995, 32, merge_order=False, precision="fp16").collect(env, gp, seed=21).to_numpy()
    L = a["ep_len"].astype(np.int64)
    assert L.sum() == a["obs"].shape[0] and L.min() >= 1 and L.max() <= 65
    rng = np.random.default_rng(1)
    _check_f16_collect(oracle, a, op, 3, 3, 32, 21, 0, [int(e) for e in rng.choice(65536, size=24, replace=False)])
    # how far the f16-input mode is from the reference's f32 arithmetic on the same records (documented in DESIGN.md §2)
    L0 = np.concatenate([[0], np.cumsum(L)])
    worst32 = 0.0
    for e in rng.choice(65536, size=8, replace=False):
        for t in range(int(L[e])):
            r = int(L0[e]) + t
            obs = a["obs"][r].astype(np.int64)
            masks = (a["logits"][r] != np.float32(-1e10)).tolist()
            lr, vr = op.forward(obs.tolist(), masks, arith=oracle.ARITH_REF)
            worst32 = max(worst32, float(np.max(np.abs(a["logits"][r] - np.asarray(lr, np.float32)))), abs(float(a["values"][r]) - vr))
    assert 1e-6 < worst32 < 1e-2, worst32
    # a twist that does not map cells to cells has no f16 image
    arrs = make_policy_arrays(9, seed=0, emb=32, hidden=32)
    bad = list(range(81)); bad[0], bad[9] = bad[9], bad[0]
    pol = amd_policy(arrs, [bad], [[0, 1, 2, 3]])
    with pytest.raises(RuntimeError, match="f16"):
        tw.collector.PPOCollector(4, 0.9, 0.9, 1, precision="fp16").collect(tw.env.Puzzle(3, 3, 2, 2, 256), pol)


# ------------------------------------------------------------------------------ trainer hand-off
def test_trainer_handoff_matches_reference_formulas(tw, oracle):
    """twisterl_amd.trainer.ppo_data_to_torch / az_data_to_torch against a CPU restatement of the reference's
    PPO.data_to_torch / AZ.data_to_torch (src/twisterl/rl/ppo.py:25-61, rl/az.py:28-46) on the same collected data."""
    import torch
    from twisterl_amd import trainer
    gp, _ = _pair(oracle, 9, 2, 64, 32, twists=True)
    env = tw.env.Puzzle(3, 3, 6, 2, 256)
    data = tw.collector.PPOCollector(**{"num_episodes": 200, "gamma": 0.99, "lambda": 0.95, "num_cores": 1}, seed=4).collect(env, gp)
    a = data.to_numpy()
    n = len(data)
    # reference formulas on the CPU (ppo.py:37-59)
    np_obs = np.zeros((n, 81), dtype=float)
    for i, o in enumerate(a["obs"].astype(int)):
        np_obs[i, o] = 1.0
    t_logits = torch.tensor(a["logits"], dtype=torch.float)
    t_acts = torch.tensor(a["actions"].astype(np.int64), dtype=torch.long)
    t_advs = torch.tensor(a["advs"], dtype=torch.float)
    want_logp = torch.distributions.Categorical(logits=t_logits).log_prob(t_acts).numpy()
    want_norm = ((t_advs - t_advs.mean()) / (t_advs.std() + 1e-8)).numpy()
    for norm in (False, True):
        pt_obs, pt_logp, pt_acts, pt_advs, pt_rets, pt_perm = trainer.ppo_data_to_torch(data, 81, normalize_advantage=norm)
        assert pt_obs.is_cuda and pt_obs.dtype == torch.float32 and pt_acts.dtype == torch.int64 and pt_perm.dtype == torch.int64
        assert np.array_equal(pt_obs.cpu().numpy(), np_obs.astype(np.float32))
        assert np.array_equal(pt_acts.cpu().numpy(), a["actions"].astype(np.int64))
        assert np.array_equal(pt_perm.cpu().numpy(), a["perms"].astype(np.int64))
        assert np.array_equal(f32_bits(pt_rets.cpu().numpy()), f32_bits(a["rets"]))
        np.testing.assert_allclose(pt_logp.cpu().numpy(), want_logp, atol=2e-6, rtol=1e-6)
        if norm:
            np.testing.assert_allclose(pt_advs.cpu().numpy(), want_norm, atol=1e-5, rtol=1e-5)
        else:
            assert np.array_equal(f32_bits(pt_advs.cpu().numpy()), f32_bits(a["advs"]))
    # mini-batch of rows: same values, normalisation statistics still those of the whole collect
    lo, hi = 37, 37 + 300
    mb = trainer.ppo_data_to_torch(data, 81, normalize_advantage=True, rows=(lo, hi))
    assert np.array_equal(mb[0].cpu().numpy(), np_obs[lo:hi].astype(np.float32))
    np.testing.assert_allclose(mb[3].cpu().numpy(), want_norm[lo:hi], atol=1e-5, rtol=1e-5)
    m, sd = trainer.adv_stats(data)
    assert abs(m - float(a["advs"].astype(np.float64).mean())) < 1e-9 and abs(sd - float(a["advs"].astype(np.float64).std(ddof=1))) < 1e-9
    # AlphaZero data (az.py:28-46)
    az = tw.collector.AZCollector(32, 6, 1.41, 1, 1, seed=2).collect(env, gp)
    b = az.to_numpy()
    o2, p2, v2 = trainer.az_data_to_torch(az, 81)
    want = np.zeros((len(az), 81), np.float32)
    for i, o in enumerate(b["obs"].astype(int)):
        want[i, o] = 1.0
    assert np.array_equal(o2.cpu().numpy(), want) and v2.shape == (len(az), 1)
    assert np.array_equal(f32_bits(p2.cpu().numpy()), f32_bits(b["logits"]))
    assert np.array_equal(f32_bits(v2.cpu().numpy()[:, 0]), f32_bits(b["remaining_values"]))
    with pytest.raises(RuntimeError):
        trainer.ppo_data_to_torch(az, 81)


@pytest.mark.parametrize("w,emb,hidden,E,diff", [(4, 64, 32, 301, 9), (2, 32, 32, 77, 3)])
def test_trainer_one_hot_of_boards_whose_cells_own_multiples_of_four_ids(tw, oracle, w, emb, hidden, E, diff):
    """PPO.data_to_torch's dense one-hot (src/twisterl/rl/ppo.py:37-39) for Puzzle-15 (16 x 16 ids) and the 2 x 2 board (4 x 4): the
    form of the kernel that writes eight rows per wave and trip (tw_trainer.hip), whole collects and row ranges that start and end
    inside such a group of rows."""
    from twisterl_amd import trainer
    n2 = w * w
    gp, _ = _pair(oracle, n2, 6, emb, hidden, twists=True)
    env = tw.env.Puzzle(w, w, diff, 2, 256)
    data = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.99, "lambda": 0.95, "num_cores": 1}, seed=8).collect(env, gp)
    a = data.to_numpy()
    n = len(data)
    want = np.zeros((n, n2 * n2), np.float32)
    for i, o in enumerate(a["obs"].astype(int)):
        want[i, o] = 1.0
    assert np.array_equal(trainer.ppo_data_to_torch(data, n2 * n2)[0].cpu().numpy(), want)
    for lo, hi in ((0, 1), (5, 5 + 33), (n - 13, n), (17, n - 3)):
        got = trainer.ppo_data_to_torch(data, n2 * n2, rows=(lo, hi))[0].cpu().numpy()
        assert got.shape == (hi - lo, n2 * n2) and np.array_equal(got, want[lo:hi]), (lo, hi)


# ------------------------------------------------------------------------------ policy sync
@pytest.mark.parametrize("n2,emb,hidden,twists", [(9, 64, 32, True), (16, 512, 256, True), (4, 96, 128, False)])
def test_policy_update_from_torch_equals_rebuilding(tw, oracle, n2, emb, hidden, twists):
    """Policy.update_from_torch (device-to-device image rebuild) gives the same policy, bit for bit, as building a new
    Policy from the reference's host export (BasicPolicy.to_rust, nn/policy.py:191-199 + nn/utils.py:17-79): identical
    collects in both precisions."""
    import torch
    w = int(round(n2 ** 0.5))
    arrs_old = make_policy_arrays(n2, seed=1, emb=emb, hidden=hidden)
    arrs_new = make_policy_arrays(n2, seed=2, emb=emb, hidden=hidden)
    op_, ap_ = puzzle_transpose_twist(w) if twists else ((), ())
    pol = amd_policy(arrs_old, op_, ap_)           # built from weight set 1 ...
    ref = amd_policy(arrs_new, op_, ap_)           # ... reference: rebuilt from weight set 2 through the host
    we, be, (w1, b1, _), (wa, ba, _), (wv, bv, _) = arrs_new[0], arrs_new[1], arrs_new[2][0], arrs_new[3][0], arrs_new[4][0]
    state = {                                      # the trainer's tensors: torch layout [out][in], on the GPU
        "embeddings.weight": torch.tensor(we.T.copy()).cuda(), "embeddings.bias": torch.tensor(be).cuda(),
        "common.0.weight": torch.tensor(w1.reshape(emb, hidden).T.copy()).cuda(), "common.0.bias": torch.tensor(b1).cuda(),
        "action.0.weight": torch.tensor(wa.reshape(hidden, 4).T.copy()).cuda(), "action.0.bias": torch.tensor(ba).cuda(),
        "value.0.weight": torch.tensor(wv.reshape(hidden, 1).T.copy()).cuda(), "value.0.bias": torch.tensor(bv).cuda(),
    }
    env = tw.env.Puzzle(w, w, 5, 2, 256)
    before = tw.collector.PPOCollector(64, 0.99, 0.95, 1).collect(env, pol, seed=3).to_numpy()
    pol.update_from_torch(state)
    for prec in ("fp32", "fp16", "fp16x2"):
        coll = tw.collector.PPOCollector(**{"num_episodes": 200, "gamma": 0.99, "lambda": 0.95, "num_cores": 1}, seed=3, precision=prec)
        a, b = coll.collect(env, pol, seed=3).to_numpy(), coll.collect(env, ref, seed=3).to_numpy()
        for k in a:
            assert np.array_equal(a[k], b[k]), (prec, k)
    after = tw.collector.PPOCollector(64, 0.99, 0.95, 1).collect(env, pol, seed=3).to_numpy()
    assert not np.array_equal(before["logits"][:8], after["logits"][:8])          # the weights did change
    la, va = pol.evaluate_batch(0, np.arange(n2)[None, :] * n2 + np.arange(n2)[None, :], np.ones((1, 4), np.uint8))
    lb, vb = ref.evaluate_batch(0, np.arange(n2)[None, :] * n2 + np.arange(n2)[None, :], np.ones((1, 4), np.uint8))
    assert np.array_equal(f32_bits(la), f32_bits(lb)) and np.array_equal(f32_bits(va), f32_bits(vb))
    with pytest.raises(ValueError):
        pol.update_from_torch({**state, "common.0.bias": torch.zeros(3).cuda()})


@pytest.mark.parametrize("n2,emb,common,pl,vl", [(9, 64, (128, 64), (), ()), (16, 128, (96,), (32,), (16,)), (4, 32, (), (24,), ())])
def test_policy_update_from_torch_for_any_depth(tw, oracle, n2, emb, common, pl, vl):
    """Policy.update_from_torch for the stacks the generic engine runs (tw_policy_update_device_layers): the Linear layers of a
    BasicPolicy stack are the even entries of its torch Sequential (nn/utils.py:82-93).  Same policy, bit for bit, as one built
    from the host export: identical PPO and self-play collects and batched evaluations."""
    import torch
    from tests.util import make_deep_policy_arrays
    w = int(round(n2 ** 0.5))
    arrs_old = make_deep_policy_arrays(n2, seed=1, emb=emb, common=common, policy_layers=pl, value_layers=vl, scale=2.0)
    arrs_new = make_deep_policy_arrays(n2, seed=2, emb=emb, common=common, policy_layers=pl, value_layers=vl, scale=2.0)
    pol, ref = amd_policy(arrs_old), amd_policy(arrs_new)
    we, be, cs, as_, vs = arrs_new
    state = {"embeddings.weight": torch.tensor(we.T.copy()).cuda(), "embeddings.bias": torch.tensor(be).cuda()}
    for name, layers in (("common", cs), ("action", as_), ("value", vs)):
        for i, (wl, bl, _) in enumerate(layers):
            state[f"{name}.{2 * i}.weight"] = torch.tensor(np.asarray(wl).reshape(-1, len(bl)).T.copy()).cuda()     # export: weight.T.flatten()
            state[f"{name}.{2 * i}.bias"] = torch.tensor(np.asarray(bl)).cuda()
    env = tw.env.Puzzle(w, w, 4, 2, 256)
    before = tw.collector.PPOCollector(64, 0.99, 0.95, 1).collect(env, pol, seed=3).to_numpy()
    pol.update_from_torch(state)
    a, b = (tw.collector.PPOCollector(300, 0.99, 0.95, 1).collect(env, p_, seed=3).to_numpy() for p_ in (pol, ref))
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert not np.array_equal(before["logits"][:8], a["logits"][:8])
    za, zb = (tw.collector.AZCollector(30, 8, 1.41, 1, 1).collect(env, p_, seed=5).to_numpy() for p_ in (pol, ref))
    for k in za:
        assert np.array_equal(za[k], zb[k]), k
    obs = np.arange(n2)[None, :] * n2 + np.arange(n2)[None, :]
    la, va = pol.evaluate_batch(0, obs, np.ones((1, 4), np.uint8))
    lb, vb = ref.evaluate_batch(0, obs, np.ones((1, 4), np.uint8))
    assert np.array_equal(f32_bits(la), f32_bits(lb)) and np.array_equal(f32_bits(va), f32_bits(vb))
    with pytest.raises(KeyError):
        pol.update_from_torch({k: v for k, v in state.items() if k != "embeddings.bias"})


# ------------------------------------------------------------------------------ f32-equivalent split-f16 mode
@pytest.mark.parametrize("w,h,diff,emb,hidden,E,twists", [
    (3, 3, 5, 64, 32, 300, False),      # two embedding tiles (the minimum), ragged workgroup tail
    (3, 3, 12, 96, 64, 129, True),      # three tiles (odd), twists
    (2, 2, 3, 128, 32, 64, False),      # four tiles, 2x2 board
    (3, 2, 4, 160, 128, 70, False),     # five tiles, non-square board
    (4, 4, 6, 512, 256, 256, True),     # Puzzle-15 at the benchmark's network size, with twists
])
def test_ppo_collect_f16x2_mode(tw, oracle, w, h, diff, emb, hidden, E, twists):
    """precision="fp16x2": every f32 operand as two f16 terms on the f16 matrix core.  Per record: env transitions / obs /
    masks / rewards / twist draws / sampling bit-exact (replay), logits and values within 1e-5 of the REFERENCE f32
    arithmetic (BASELINE.json's tolerance), GAE bit-exact on the GPU's own values."""
    n2 = w * h
    if twists and w != h:
        pytest.skip("transpose twist needs a square board")
    gp, op = _pair(oracle, n2, 1, emb, hidden, twists=twists)
    genv = tw.env.Puzzle(w, h, diff, 2, 256)
    coll = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32},
                                     seed=13, merge_order=False, precision="fp16x2")
    a = coll.collect(genv, gp, seed=13).to_numpy()
    b = coll.collect(genv, gp, seed=13).to_numpy()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    worst = _check_f16_collect(oracle, a, op, w, h, diff, 13, 2 if twists else 0, range(E), arith=oracle.ARITH_REF, atol=1e-5)
    assert worst < 1e-5, worst
    # against the exact f32 mode of the same library: the same trajectories except where a Gumbel near-tie flips
    c = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32},
                                  seed=13, merge_order=False).collect(genv, gp, seed=13).to_numpy()
    La, Lc = a["ep_len"].astype(np.int64), c["ep_len"].astype(np.int64)
    sa, sc = np.concatenate([[0], np.cumsum(La)]), np.concatenate([[0], np.cumsum(Lc)])
    same = sum(1 for e in range(E) if La[e] == Lc[e] and np.array_equal(a["actions"][sa[e]:sa[e + 1]], c["actions"][sc[e]:sc[e + 1]]))
    assert same >= 0.99 * E, (same, E)


def test_f16x2_full_size_sample(tw, oracle):
    """65,536 Puzzle-8 envs (BASELINE config 2 size) in the split-f16 mode: record-count identities, per-record replay parity
    against the reference f32 arithmetic on a sample (1e-5), and identical episode lengths / actions to the exact mode."""
    gp, op = _pair(oracle, 9, 0, 512, 256)
    env = tw.env.Puzzle(3, 3, 32, 2, 256)
    a = tw.collector.PPOCollector(65536, 0.995, 0.995, 32, merge_order=False, precision="fp16x2").collect(env, gp, seed=21).to_numpy()
    c = tw.collector.PPOCollector(65536, 0.995, 0.995, 32, merge_order=False).collect(env, gp, seed=21).to_numpy()
    L = a["ep_len"].astype(np.int64)
    assert L.sum() == a["obs"].shape[0] and L.min() >= 1 and L.max() <= 65
    rng = np.random.default_rng(2)
    worst = _check_f16_collect(oracle, a, op, 3, 3, 32, 21, 0, [int(e) for e in rng.choice(65536, size=24, replace=False)],
                               arith=oracle.ARITH_REF, atol=1e-5)
    assert worst < 1e-5, worst
    same_len = float(np.mean(a["ep_len"] == c["ep_len"]))
    assert same_len >= 0.999, same_len                        # a Gumbel near-tie may flip an action once in a long while
    if np.array_equal(a["ep_len"], c["ep_len"]):
        assert float(np.mean(a["actions"] == c["actions"])) >= 0.999
        np.testing.assert_allclose(a["rets"], c["rets"], atol=1e-5, rtol=1e-5) if np.array_equal(a["actions"], c["actions"]) else None


def test_persistent_lane_mode_bit_exact(tw, oracle):
    """More episodes than resident lanes (65,536): the f32 kernel runs 256 persistent workgroups whose lanes take the next
    episode from a queue when theirs is over (ragged lengths: Puzzle-8, one scramble move, depth 24) -- longest-looking episodes
    first.  Which lane runs an episode must not matter: bit-identical to the oracle, to the non-persistent launch and to the queue in
    index order."""
    import os
    gp, op = _pair(oracle, 9, 5, 32, 32, twists=True)
    E = 70_000
    genv, oenv = tw.env.Puzzle(3, 3, 1, 24, 256), oracle.Puzzle(3, 3, 1, 24, 256)
    coll = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32}, seed=17, merge_order=True)
    g = coll.collect(genv, gp, seed=17)
    o = oracle.ppo_collect(oenv, op, E, 0.995, 0.995, seed=17, arith=oracle.ARITH_CHAIN, det_log=True, merge_order=True, num_threads=8)
    _assert_same_collect(g, o, 9)
    L = g.to_numpy()["ep_len"]
    assert L.min() < L.max() and L.mean() < 0.8 * L.max()      # the lengths are indeed ragged
    with _lib.launch_option(_lib.TW_OPT_NO_PERSIST, 1):
        h = coll.collect(genv, gp, seed=17).to_numpy()
    a = g.to_numpy()
    for k in a:
        assert np.array_equal(a[k], h[k]), k
    with _lib.launch_option(_lib.TW_OPT_AZ_VARIANT, 64):        # the queue in index order instead of longest-looking first: the same bytes
        h = coll.collect(genv, gp, seed=17).to_numpy()
    for k in a:
        assert np.array_equal(a[k], h[k]), k
    # the f16-matrix-core kernels have the same mode (their lane halves own one episode each): identical to their
    # non-persistent launches
    for prec in ("fp16", "fp16x2"):
        c16 = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32}, seed=17, precision=prec)
        p1 = c16.collect(genv, gp, seed=17).to_numpy() if prec == "fp16" else None
        if prec == "fp16x2":
            gp2, _ = _pair(oracle, 9, 5, 64, 32, twists=True)          # the split mode needs two embedding tiles
            p1 = c16.collect(genv, gp2, seed=17).to_numpy()
        with _lib.launch_option(_lib.TW_OPT_NO_PERSIST, 1):
            p2 = c16.collect(genv, gp2 if prec == "fp16x2" else gp, seed=17).to_numpy()
        for k in p1:
            assert np.array_equal(p1[k], p2[k]), (prec, k)


def test_az_persistent_lane_mode_bit_exact(tw, oracle):
    """AlphaZero self-play with more episodes than resident lanes: persistent lanes reuse their tree arena for every
    episode they take; bit-identical to the oracle and to the plain launch."""
    import os
    gp, op = _pair(oracle, 9, 7, 32, 32, twists=False, scale=3.0)
    E = 66_000
    genv, oenv = tw.env.Puzzle(3, 3, 1, 3, 256), oracle.Puzzle(3, 3, 1, 3, 256)
    coll = tw.collector.AZCollector(num_episodes=E, num_mcts_searches=3, C=1.41, max_expand_depth=1, num_cores=32, seed=19, merge_order=False)
    g = coll.collect(genv, gp, seed=19).to_numpy()
    o = oracle.az_collect(oenv, op, E, 3, 1.41, 1, seed=19, arith=oracle.ARITH_CHAIN, num_threads=8, merge_order=False, det_math=True)
    assert np.array_equal(g["ep_len"], o.ep_len)
    assert np.array_equal(g["obs"].astype(np.int64), o.obs)
    assert np.array_equal(f32_bits(g["logits"]), f32_bits(o.logits))
    assert np.array_equal(f32_bits(g["remaining_values"]), f32_bits(o.additional_data["remaining_values"]))
    with _lib.launch_option(_lib.TW_OPT_NO_PERSIST, 1):
        h = coll.collect(genv, gp, seed=19).to_numpy()
    for k in g:
        assert np.array_equal(g[k], h[k]), k


def test_mid_size_batches_use_the_queue_with_the_small_batch_geometry(tw, oracle):
    """Between CUs x 32 (8,192) and ~40k episodes (self-play: 49k) the small-batch geometry runs with the episode queue:
    four waves share 32 episodes and take the next one together.  PPO and self-play: bit-identical to the oracle and to the
    launch without the queue; the launch geometry is the persistent one (one workgroup per CU)."""
    import os
    gp, op = _pair(oracle, 9, 5, 64, 128, twists=True)
    E = 9_000
    genv, oenv = tw.env.Puzzle(3, 3, 1, 24, 256), oracle.Puzzle(3, 3, 1, 24, 256)
    coll = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32}, seed=23)
    g = coll.collect(genv, gp, seed=23)
    o = oracle.ppo_collect(oenv, op, E, 0.995, 0.995, seed=23, arith=oracle.ARITH_CHAIN, det_log=True, num_threads=8)
    _assert_same_collect(g, o, 9)
    assert g.stats["rollout_threads"] == 256 and g.stats["rollout_blocks"] * 32 < E      # fewer lanes than episodes
    a = g.to_numpy()
    gp2, op2 = _pair(oracle, 9, 7, 64, 128, twists=False, scale=3.0)
    aenv, aoenv = tw.env.Puzzle(3, 3, 2, 3, 256), oracle.Puzzle(3, 3, 2, 3, 256)
    acoll = tw.collector.AZCollector(num_episodes=E, num_mcts_searches=6, C=1.41, max_expand_depth=1, num_cores=32, seed=29, merge_order=False)
    z = acoll.collect(aenv, gp2, seed=29).to_numpy()
    zo = oracle.az_collect(aoenv, op2, E, 6, 1.41, 1, seed=29, arith=oracle.ARITH_CHAIN, num_threads=8, merge_order=False, det_math=True)
    assert np.array_equal(z["ep_len"], zo.ep_len) and np.array_equal(z["obs"].astype(np.int64), zo.obs)
    assert np.array_equal(f32_bits(z["logits"]), f32_bits(zo.logits))
    assert np.array_equal(f32_bits(z["remaining_values"]), f32_bits(zo.additional_data["remaining_values"]))
    with _lib.launch_option(_lib.TW_OPT_NO_PERSIST, 1):
        h = coll.collect(genv, gp, seed=23).to_numpy()
        zh = acoll.collect(aenv, gp2, seed=29).to_numpy()
    for k in a:
        assert np.array_equal(a[k], h[k]), k
    for k in z:
        assert np.array_equal(z[k], zh[k]), k


def test_compiled_host_over_the_c_abi_matches_the_python_mirror(tw, oracle):
    """examples/collect_from_c.c (gcc, no Python, no torch in the process) builds a policy from an LCG, collects through the
    C ABI and prints FNV-1a checksums of every field; the same weights through twisterl_amd.nn give the same bytes."""
    import os, subprocess
    from twisterl_amd import build as tb
    exe = tb.build_c_example()
    E, seed = 500, 7
    env = dict(os.environ)
    out = subprocess.run([exe, str(E), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    got = {}
    for line in out.stdout.splitlines():
        parts = line.split()
        if len(parts) == 3 and parts[0] in ("obs", "logits", "perms", "values", "rewards", "actions", "advs", "rets", "ep_len"):
            got[parts[0]] = (int(parts[1]), int(parts[2], 16))
    assert set(got) >= {"obs", "logits", "values", "actions", "advs", "rets", "ep_len"}

    state = 0x9e3779b97f4a7c15
    def filled(n, bound):                                  # the C program's LCG, bit for bit
        nonlocal state
        v = np.empty(n, dtype=np.float32)
        for i in range(n):
            state = (state * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
            r = np.float32(state >> 40)
            v[i] = (r / np.float32(16777216.0) * np.float32(2.0) - np.float32(1.0)) * np.float32(bound)
        return v
    OBS, EMB, HID, ACT = 81, 64, 64, 4
    emb, emb_b = filled(OBS * EMB, 0.111), filled(EMB, 0.111)
    w1, b1 = filled(EMB * HID, 0.125), filled(HID, 0.125)
    wa, ba = filled(HID * ACT, 0.125), filled(ACT, 0.125)
    wv, bv = filled(HID, 0.125), filled(1, 0.125)
    pol = tw.nn.Policy(tw.nn.EmbeddingBag(emb.reshape(OBS, EMB).tolist(), emb_b.tolist(), True, [OBS], 0),
                       tw.nn.Sequential([tw.nn.Linear(w1.tolist(), b1.tolist(), True)]),
                       tw.nn.Sequential([tw.nn.Linear(wa.tolist(), ba.tolist(), False)]),
                       tw.nn.Sequential([tw.nn.Linear(wv.tolist(), bv.tolist(), False)]), [], [])
    a = tw.collector.PPOCollector(E, 0.995, 0.995, 1).collect(tw.env.Puzzle(3, 3, 6, 2, 256), pol, seed=seed).to_numpy()

    def fnv1a(arr):
        h = 0xcbf29ce484222325
        for b in np.ascontiguousarray(arr).view(np.uint8).reshape(-1).tolist():
            h = ((h ^ b) * 0x100000001b3) & (2**64 - 1)
        return h
    for k, (nbytes, h) in got.items():
        assert a[k].nbytes == nbytes, (k, a[k].nbytes, nbytes)
        assert fnv1a(a[k]) == h, k


@pytest.mark.parametrize("geom", ["32", "8"])
def test_every_launch_shape_gives_the_same_bytes(tw, oracle, geom):
    """The launch shape is chosen from the batch size (16 / 32 episodes per workgroup shared by four waves, or 8 waves x 32);
    tw_set_launch_option(TW_OPT_FORCE_GEOM) pins it.  A Puzzle-15 collect and a self-play collect must not depend on it, bit for bit -- and the
    default shape is checked against the oracle by the other tests."""
    import os
    gp, _ = _pair(oracle, 16, 1, 512, 256, twists=True)
    env = tw.env.Puzzle(4, 4, 6, 2, 256)
    pc = tw.collector.PPOCollector(600, 0.995, 0.995, 1)
    gz, _ = _pair(oracle, 9, 2, 64, 128, twists=False, scale=3.0)
    zenv = tw.env.Puzzle(3, 3, 3, 2, 256)
    zc = tw.collector.AZCollector(300, 12, 1.41, 1, 1)
    a = pc.collect(env, gp, seed=31)
    z = zc.collect(zenv, gz, seed=37)
    with _lib.launch_option(_lib.TW_OPT_FORCE_GEOM, int(geom)):
        b = pc.collect(env, gp, seed=31)
        y = zc.collect(zenv, gz, seed=37)
    assert (a.stats["rollout_blocks"], a.stats["rollout_threads"]) != (b.stats["rollout_blocks"], b.stats["rollout_threads"])
    an, bn, zn, yn = a.to_numpy(), b.to_numpy(), z.to_numpy(), y.to_numpy()
    for k in an:
        assert np.array_equal(an[k], bn[k]), (geom, k)
    for k in zn:
        assert np.array_equal(zn[k], yn[k]), (geom, k)


def test_end_to_end_loop_sketch(tw, oracle):
    """examples/ppo_loop_sketch.py: collect -> data_to_torch -> torch PPO update -> device policy sync -> evaluate, three
    iterations with nothing going through host lists; the losses are finite and the synced policy is the trained one."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("ppo_loop_sketch", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "ppo_loop_sketch.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    hist = mod.run(iterations=3, episodes=2048, log=lambda *_: None)
    assert len(hist) == 3 and all(np.isfinite(h[0]) for h in hist) and all(0.0 <= h[1] <= 1.0 for h in hist)
