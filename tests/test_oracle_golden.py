"""Pins the CPU oracle against every known answer the reference's own tests / notebooks hold
for the hot path, plus published Philox known-answer vectors.

CPU only.  The oracle is the checker for the GPU parity tests, so it is pinned first.
"""
import json
import math
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


# ------------------------------------------------------------------ envs/puzzle.rs
def test_puzzle_fresh_is_solved(oracle):
    g = GOLD["puzzle_fresh_solved"]
    p = oracle.Puzzle(*g["ctor"])
    assert p.solved() is g["solved"]
    assert p.depth == 1                      # Puzzle::new sets depth 1 (puzzle.rs:41)


def test_puzzle_step_and_masks(oracle):
    g = GOLD["puzzle_2x2_step2"]
    p = oracle.Puzzle(*g["ctor"])
    p.step(g["action"])
    assert list(p.zero_location) == g["zero_location"]
    assert p.masks() == g["masks"]


def test_replay_123_moves_solves(oracle):
    """examples/hub_puzzle_model.ipynb: start board + the 123 actions printed by the
    reference's own solve() must end at the identity board."""
    g = GOLD["replay_123"]
    p = oracle.Puzzle(*g["ctor"])
    p.set_state(g["start"])
    assert p.depth == 256 and p.zero_location == (2, 1)
    obs, masks, rew, fin, board = oracle.replay(p, g["actions"])
    assert board[-1].tolist() == g["end"]
    assert fin[-1] and not fin[:-1].any()    # solved only at the very end
    assert rew[-1] == 1.0 and np.all(rew[:-1] == np.float32(-0.5 / 256))
    # observe: obs[i] = i*N^2 + state[i]   (puzzle.rs:183-185)
    assert np.array_equal(obs, np.arange(9)[None, :] * 9 + board)
    # every recorded action was legal under the mask of its state (the reference samples
    # masked actions only): pins action -> direction (0 left, 1 up, 2 right, 3 down)
    for t, a in enumerate(g["actions"]):
        assert masks[t, a], (t, a)
        zi = int(np.where(board[t] == 0)[0][0])
        zj = int(np.where(board[t + 1] == 0)[0][0])
        dx, dy = (zj % 3) - (zi % 3), (zj // 3) - (zi // 3)
        assert (dx, dy) == [(-1, 0), (0, -1), (1, 0), (0, 1)][a]


def test_reward_constants(oracle):
    c = GOLD["constants"]
    p = oracle.Puzzle(3, 3, 1, 2, 256)
    assert p.reward() == c["reward_solved"]
    p.step(2)                                # depth 1 -> 0, unsolved
    assert p.depth == 0 and p.reward() == c["reward_timeout"] and p.is_final()
    q = oracle.Puzzle(3, 3, 1, 2, 256)
    q.set_state([1, 0, 2, 3, 4, 5, 6, 7, 8])
    assert q.reward() == c["reward_step_max_depth_256"] and not q.is_final()


def test_step_illegal_is_noop_but_consumes_depth(oracle):
    p = oracle.Puzzle(4, 4, 3, 2, 256)
    p.set_state(list(range(16)))
    before = p.get_state()
    p.step(0)                                # blank at (0,0): left is a wall
    p.step(1)                                # up is a wall
    assert p.get_state() == before and p.depth == 254
    q = oracle.Puzzle(2, 2, 0, 1, 10)
    q.step(0); q.step(0)                     # depth 1 -> 0 -> saturates at 0 (puzzle.rs:159)
    assert q.depth == 0


def test_reset_depth_and_scramble_semantics(oracle):
    p = oracle.Puzzle(4, 4, 7, 2, 256)
    p.reset(seed=5, episode=11)
    assert p.depth == 14                     # depth_slope*difficulty (puzzle.rs:132)
    # the scramble is `difficulty` steps from the identity using stream 0 of the RNG spec
    q = oracle.Puzzle(4, 4, 7, 2, 256)
    q.set_state(list(range(16)))
    for d in range(7):
        w = oracle.philox4x32_10([11, 0, d, 0], [5, 0])
        q.step((w[0] * 4) >> 32)
    assert q.get_state() == p.get_state() and q.zero_location == p.zero_location
    z = oracle.Puzzle(3, 3, 0, 2, 256)
    z.reset()
    assert z.depth == 0 and z.solved() and z.is_final()


# ------------------------------------------------------------------ nn/layers.rs, nn/policy.rs
def _policy_from_linear(oracle, weights, bias, relu):
    # a Policy whose "common" stack is the Linear under test; embedding = identity passthrough
    n_in = len(weights) // len(bias)
    emb = np.eye(n_in, dtype=np.float32)
    return oracle.Policy(emb, np.zeros(n_in, np.float32), [(weights, bias, relu)],
                         [(np.eye(len(bias), dtype=np.float32).T.flatten(), np.zeros(len(bias)), False)],
                         [(np.zeros(len(bias)), np.zeros(1), False)], emb_relu=False)


@pytest.mark.parametrize("key", ["linear_forward", "linear_forward_relu"])
@pytest.mark.parametrize("arith", [0, 1])
def test_linear_known_answers(oracle, key, arith):
    """layers.rs:98-111.  Input [1,2] is fed as the multi-hot obs {0, 1, 1} through an identity
    embedding (so x = [1,2]); the column-major weight layout is what the test pins."""
    g = GOLD[key]
    pol = _policy_from_linear(oracle, g["weights"], g["bias"], g["relu"])
    logits, _ = pol.raw_predict([0, 1, 1], arith=arith)
    assert logits == g["out"]


def test_embedding_bag_known_answer(oracle):
    g = GOLD["embedding_bag"]
    pol = oracle.Policy(np.array(g["vectors"], np.float32), g["bias"], [],
                        [(np.eye(2, dtype=np.float32).flatten(), [0.0, 0.0], False)],
                        [(np.zeros(2), np.zeros(1), False)], emb_relu=g["relu"])
    logits, _ = pol.raw_predict(g["input"])
    assert logits == g["out"]


def test_argmax_known_answers(oracle):
    assert oracle.argmax(GOLD["argmax_basic"]["v"]) == GOLD["argmax_basic"]["idx"]
    assert oracle.argmax([float("nan"), 1.0, 0.5]) == GOLD["argmax_nan"]["idx"]
    assert oracle.argmax([]) == 0
    assert oracle.argmax([2.0, 2.0, 1.0]) == 0          # first max wins (strict >)


def test_sample_from_logits_in_range_and_masked(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        u = rng.random(4).astype(np.float32)
        a = oracle.sample_from_logits([0.1, -1e10, 0.3, -1e10], u)
        assert a in (0, 2)
    assert oracle.sample_from_logits([0.1, 2.0, 0.3], [0.0, 0.5, 0.5]) in (1, 2)  # u=0 -> -inf noise


def test_gumbel_is_categorical(oracle):
    """Gumbel-max == categorical(softmax(logits)) -- distributional pin of policy.rs:169-172."""
    logits = np.array([0.5, -0.2, 1.1, 0.0], np.float32)
    p = np.exp(logits) / np.exp(logits).sum()
    n = 40000
    cnt = np.zeros(4)
    for i in range(n):
        w = oracle.philox4x32_10([i, 0, 0, 1], [9, 0])
        u = [(x >> 8) / 16777216.0 for x in w]
        cnt[oracle.sample_from_logits(logits, u, det_log=True)] += 1
    assert np.abs(cnt / n - p).max() < 0.01


def test_dummy_env_gae_known_answer(oracle):
    g = GOLD["dummy_env_ppo"]
    advs, rets = oracle.gae(g["rewards"], g["values"], g["gamma"], g["lambda"])
    np.testing.assert_allclose(rets, g["derived_rets"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(advs, g["derived_advs"], rtol=0, atol=1e-6)
    assert len(rets) == g["records"]


def test_gae_matches_float64_recurrence(oracle):
    rng = np.random.default_rng(3)
    r = rng.normal(size=257).astype(np.float32)
    v = rng.normal(size=257).astype(np.float32)
    advs, rets = oracle.gae(r, v, 0.995, 0.995)
    a64 = np.zeros(257); t64 = np.zeros(257)
    a64[-1] = r[-1] - v[-1]; t64[-1] = r[-1]
    for t in range(255, -1, -1):
        t64[t] = r[t] + 0.995 * (v[t + 1] + 0.995 * a64[t + 1])
        a64[t] = t64[t] - v[t]
    np.testing.assert_allclose(advs, a64, atol=1e-4)
    np.testing.assert_allclose(rets, t64, atol=1e-4)


# ------------------------------------------------------------------ collector/*.rs
def _tiny_policy(oracle, n2=9, seed=0, perms=False):
    rng = np.random.default_rng(seed)
    obs_size, E, H = n2 * n2, 16, 32
    emb = rng.normal(scale=0.3, size=(obs_size, E)).astype(np.float32)
    eb = rng.normal(scale=0.1, size=E).astype(np.float32)
    w1 = rng.normal(scale=0.3, size=(E, H)).astype(np.float32)
    b1 = rng.normal(scale=0.1, size=H).astype(np.float32)
    wa = rng.normal(scale=0.3, size=(H, 4)).astype(np.float32)
    ba = rng.normal(scale=0.1, size=4).astype(np.float32)
    wv = rng.normal(scale=0.3, size=(H, 1)).astype(np.float32)
    bv = rng.normal(scale=0.1, size=1).astype(np.float32)
    op, ap = (), ()
    if perms:
        from tests.util import puzzle_transpose_twist
        op, ap = puzzle_transpose_twist(int(math.isqrt(n2)))
    return oracle.Policy(emb, eb, [(w1.flatten(), b1, True)], [(wa.flatten(), ba, False)],
                         [(wv.flatten(), bv, False)], op, ap)


def test_merge_order_last_chunk_first(oracle):
    """collector.rs:40-46 / :101-126: merged = [chunk E-1, chunk 0, ..., chunk E-2]."""
    env = oracle.Puzzle(3, 3, 4, 2, 256)
    pol = _tiny_policy(oracle)
    a = oracle.ppo_collect(env, pol, 5, 0.9, 0.95, seed=1, merge_order=False)
    b = oracle.ppo_collect(env, pol, 5, 0.9, 0.95, seed=1, merge_order=True)
    L = a.ep_len.astype(int)
    assert np.array_equal(a.ep_len, b.ep_len)
    starts = np.concatenate([[0], np.cumsum(L)])
    order = [4, 0, 1, 2, 3]
    idx = np.concatenate([np.arange(starts[e], starts[e + 1]) for e in order])
    for f in ("obs", "logits", "perms", "values", "rewards", "actions"):
        assert np.array_equal(getattr(b, f), getattr(a, f)[idx]), f
    for k in ("advs", "rets"):
        assert np.array_equal(b.additional_data[k], a.additional_data[k][idx])
    g = GOLD["merge_order"]
    assert g["merged_actions"] == [g["chunk_actions"][1][0], g["chunk_actions"][0][0]]


def test_collect_records_terminal_state_and_keys(oracle):
    """ppo.rs:69-80,173-182: the terminal state IS recorded; keys 'advs','rets' exist."""
    env = oracle.Puzzle(3, 3, 3, 2, 256)
    pol = _tiny_policy(oracle)
    d = oracle.ppo_collect(env, pol, 64, 0.995, 0.995, seed=7)
    assert set(d.additional_data) == {"advs", "rets"}
    assert d.obs.shape[0] == d.ep_len.sum() == len(d.values) == len(d.additional_data["rets"])
    assert d.ep_len.min() >= 1 and d.ep_len.max() <= 3 * 2 + 1
    assert np.all(d.perms == -1)              # Puzzle has no twists -> perm None (env.rs:59)
    # difficulty 0 => born final => exactly one record, reward 1.0, adv = r - v
    z = oracle.ppo_collect(oracle.Puzzle(3, 3, 0, 2, 256), pol, 3, 0.9, 0.9, seed=1)
    assert z.ep_len.tolist() == [1, 1, 1] and np.all(z.rewards == 1.0)
    np.testing.assert_array_equal(z.additional_data["advs"], z.rewards - z.values)


def test_collect_is_replayable(oracle):
    """Every collected episode replays through Puzzle::step to the recorded obs/rewards."""
    env = oracle.Puzzle(4, 4, 6, 2, 256)
    pol = _tiny_policy(oracle, n2=16, seed=2)
    d = oracle.ppo_collect(env, pol, 16, 0.995, 0.995, seed=3, merge_order=False)
    pos = 0
    for e, n in enumerate(d.ep_len.astype(int)):
        p = oracle.Puzzle(4, 4, 6, 2, 256)
        p.reset(seed=3, episode=e)
        obs, masks, rew, fin, _ = oracle.replay(p, d.actions[pos:pos + n - 1])
        assert np.array_equal(obs, d.obs[pos:pos + n])
        assert np.array_equal(rew, d.rewards[pos:pos + n])
        assert fin[-1] and not fin[:-1].any()
        # masked logits: illegal actions carry exactly -1e10 (policy.rs:62)
        assert np.all((d.logits[pos:pos + n] == np.float32(-1e10)) == ~masks)
        pos += n


def test_threads_and_offsets_do_not_change_results(oracle):
    env = oracle.Puzzle(3, 3, 5, 2, 256)
    pol = _tiny_policy(oracle)
    a = oracle.ppo_collect(env, pol, 24, 0.995, 0.995, seed=4, num_threads=1, merge_order=False)
    b = oracle.ppo_collect(env, pol, 24, 0.995, 0.995, seed=4, num_threads=4, merge_order=False)
    c0 = oracle.ppo_collect(env, pol, 10, 0.995, 0.995, seed=4, episode_offset=0, merge_order=False)
    c1 = oracle.ppo_collect(env, pol, 14, 0.995, 0.995, seed=4, episode_offset=10, merge_order=False)
    for f in ("obs", "logits", "values", "rewards", "actions"):
        assert np.array_equal(getattr(a, f), getattr(b, f))
        assert np.array_equal(getattr(a, f), np.concatenate([getattr(c0, f), getattr(c1, f)]))


def test_arith_modes_agree_within_tolerance(oracle):
    """The fma-chain order (what the HIP exact mode computes) stays within 1e-5 of the
    reference's un-fused order on a full-size synthetic Puzzle-15 policy."""
    from tests.util import make_policy_arrays
    arrs = make_policy_arrays(16, seed=0)
    pol = oracle.Policy(*arrs)
    rng = np.random.default_rng(1)
    worst = 0.0
    for _ in range(50):
        board = rng.permutation(16)
        obs = (np.arange(16) * 16 + board).tolist()
        l0, v0 = pol.raw_predict(obs, arith=0)
        l1, v1 = pol.raw_predict(obs, arith=1)
        worst = max(worst, np.abs(np.array(l0) - np.array(l1)).max(), abs(v0 - v1))
        # independent float64 forward
        emb, eb, (w1, b1, _), (wa, ba, _), (wv, bv, _) = arrs[0], arrs[1], arrs[2][0], arrs[3][0], arrs[4][0]
        h0 = np.maximum(eb.astype(np.float64) + emb[obs].astype(np.float64).sum(0), 0)
        h1 = np.maximum(h0 @ w1.reshape(512, 256).astype(np.float64) + b1, 0)
        la = h1 @ wa.reshape(256, 4).astype(np.float64) + ba
        np.testing.assert_allclose(l0, la, atol=1e-5)
        np.testing.assert_allclose(v0, float((h1 @ wv.reshape(256, 1).astype(np.float64) + bv)[0]), atol=1e-5)
    assert worst < 1e-5


def test_twists_match_torch_twin_semantics(oracle):
    """policy.rs:81-83,95-97 vs src/twisterl/nn/policy.py:143-154: permuting the obs ids by
    obs_perm and gathering logits by act_perm; for the board-transpose symmetry a transposed
    board under perm 1 must see exactly the un-permuted forward of the original board."""
    from tests.util import puzzle_transpose_twist
    pol = _tiny_policy(oracle, n2=9, seed=5, perms=True)
    plain = _tiny_policy(oracle, n2=9, seed=5, perms=False)
    op, ap = puzzle_transpose_twist(3)
    rng = np.random.default_rng(2)
    board = rng.permutation(9)
    obs = (np.arange(9) * 9 + board).tolist()
    l_id, v_id = pol.raw_predict(obs, perm=0)
    l_pl, v_pl = plain.raw_predict(obs)
    assert l_id == l_pl and v_id == v_pl      # perm 0 is the identity
    l_t, v_t = pol.raw_predict(obs, perm=1)
    obs_t = sorted(op[1][o] for o in obs)     # ids of the transposed board (EmbeddingBag sums: order-free up to fp)
    l_ref, v_ref = plain.raw_predict([op[1][o] for o in obs])
    assert v_t == v_ref and l_t == [l_ref[a] for a in ap[1]]
    assert len(obs_t) == 9


# ------------------------------------------------------------------ RNG spec
def test_philox_known_answer_vectors(oracle):
    """The three philox4x32-10 lines of Random123's published known-answer file (tests/kat_vectors: counter, key -> output)."""
    assert oracle.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_logf_det_tracks_libm(oracle):
    """The deterministic log stays within 2 ulp of libm's logf over the sampler's domain."""
    rng = np.random.default_rng(0)
    u = (rng.integers(1, 1 << 24, size=20000).astype(np.float32)) / np.float32(16777216.0)
    u = np.concatenate([u, np.float32([2.0 ** -24, 1 - 2.0 ** -24, 0.5, 0.25, 0.75])])
    a = oracle.logf_det_array(u)
    ref = np.log(u.astype(np.float64))
    ulp = np.spacing(np.abs(ref).astype(np.float32))
    assert np.max(np.abs(a - ref) / ulp) <= 2.0
    b = oracle.logf_det_array(np.abs(a))
    ref2 = np.log(np.abs(a).astype(np.float64))
    ulp2 = np.maximum(np.spacing(np.abs(ref2).astype(np.float32)), np.float32(1e-9))
    assert np.max(np.abs(b - ref2) / ulp2) <= 2.5
    assert oracle.logf_det(0.0) == -math.inf and oracle.logf_det(math.inf) == math.inf


def test_expf_det_tracks_libm(oracle):
    """The deterministic exp of the masked softmax stays within 1 ulp of libm's expf."""
    xs = np.concatenate([np.linspace(-87, 88, 4001), np.random.default_rng(1).normal(size=4000) * 4]).astype(np.float32)
    a = np.array([oracle.expf_det(float(x)) for x in xs], np.float32)
    ref = np.exp(xs.astype(np.float64))
    ulp = np.spacing(ref.astype(np.float32))
    assert np.max(np.abs(a - ref) / ulp) <= 1.0
    assert oracle.expf_det(0.0) == 1.0 and oracle.expf_det(-200.0) == 0.0 and oracle.expf_det(100.0) == math.inf
    # det-exp softmax vs libm softmax: same probabilities within 1e-6
    pol = _tiny_policy(oracle)
    obs = [i * 9 + i for i in range(9)]
    p0, _ = pol.predict(obs, [True, False, True, True])
    oracle.set_det_exp(True)
    try:
        p1, _ = pol.predict(obs, [True, False, True, True])
    finally:
        oracle.set_det_exp(False)
    np.testing.assert_allclose(p0, p1, atol=1e-6)
    assert p1[1] == 0.0


# ------------------------------------------------------------------ MCTS / AZ (search.rs, az.rs)
def test_az_collect_shapes_and_remaining_values(oracle):
    """az.rs:97-106,177-186: logits slot holds MCTS probs, values/rewards/actions empty,
    key 'remaining_values' = sum of rewards from t to the end."""
    env = oracle.Puzzle(3, 3, 2, 2, 256)
    pol = _tiny_policy(oracle)
    d = oracle.az_collect(env, pol, 6, 20, 1.41, 1, seed=2, merge_order=False)
    assert set(d.additional_data) == {"remaining_values"}
    assert len(d.values) == len(d.rewards) == len(d.actions) == 0
    assert np.all(d.perms == -1)
    np.testing.assert_allclose(d.logits.sum(1), 1.0, atol=1e-6)
    pos = 0
    for n in d.ep_len.astype(int):
        boards = d.obs[pos:pos + n] - np.arange(9)[None, :] * 9
        solved = np.all(boards == np.arange(9)[None, :], axis=1)
        # reward per record: solved 1.0; else depth-0 -> -0.5 (only possible at the last record)
        r = np.where(solved, 1.0, -0.5 / 256).astype(np.float32)
        if not solved[-1]:
            r[-1] = -0.5
        rem = np.float32(0)
        tot = np.float32(0)
        tv = []
        for x in r:
            tv.append(tot); tot = np.float32(tot + x)
        np.testing.assert_array_equal(d.additional_data["remaining_values"][pos:pos + n],
                                      np.float32(tot) - np.float32(tv))
        pos += n


def test_mcts_visit_counts_sum_to_searches(oracle):
    env = oracle.Puzzle(3, 3, 4, 2, 256)
    env.reset(seed=1, episode=0)
    pol = _tiny_policy(oracle)
    probs = oracle.mcts_probs(env, pol, 50, 1.41, 1)
    legal = env.masks()
    assert abs(sum(probs) - 1.0) < 1e-6
    assert all((p == 0.0) for p, m in zip(probs, legal) if not m)
    # counts are multiples of 1/50: every search back-propagates through exactly one root child
    assert all(abs(p * 50 - round(p * 50)) < 1e-4 for p in probs)


# ------------------------------------------------------------------ solve / evaluate (rl/solve.rs, rl/evaluate.rs)
def test_solve_and_evaluate_semantics(oracle):
    pol = _tiny_policy(oracle)
    # an already-final env: no step is taken, total = reward of the final state (solve.rs:30,65-66)
    done = oracle.Puzzle(3, 3, 0, 2, 256); done.reset()
    (s, r), acts = oracle.solve(done, pol, True, 1)
    assert (s, r, acts) == (1.0, 1.0, [])
    # one move from solved with depth left: best-of-N keeps the better (success, reward) tuple
    env = oracle.Puzzle(3, 3, 1, 2, 256); env.set_state([1, 0, 2, 3, 4, 5, 6, 7, 8])
    (s1, r1), a1 = oracle.solve(env, pol, False, 1, seed=2)
    (s8, r8), a8 = oracle.solve(env, pol, False, 16, seed=2)
    assert (s8, r8) >= (s1, r1) and env.get_state() == [1, 0, 2, 3, 4, 5, 6, 7, 8]
    if s8 == 1.0:      # replay the returned actions: they must solve the board
        q = env.clone()
        for a in a8:
            q.step(a)
        assert q.solved() and abs(r8 - (1.0 - 0.5 / 256 * len(a8))) < 1e-5
    # evaluate = mean over episodes of reset + solve; difficulty 0 => always solved, reward 1
    assert oracle.evaluate(oracle.Puzzle(3, 3, 0, 2, 256), pol, 10, True, 1) == (1.0, 1.0)
    sr, mr = oracle.evaluate(oracle.Puzzle(3, 3, 3, 2, 256), pol, 40, False, 4, seed=1)
    assert 0.0 <= sr <= 1.0 and -0.6 <= mr <= 1.0


def test_round_f16_matches_numpy_float16(oracle):
    """The ARITH_F16 rounding is IEEE binary16 round-to-nearest-even (numpy's float16 cast)."""
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.normal(size=2000).astype(np.float32), (rng.normal(size=500) * 1e-6).astype(np.float32),
                         (rng.normal(size=500) * 3e4).astype(np.float32),
                         np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 6.1e-5, 1.0009765625, 1.00048828125], np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).astype(np.float32)
    got = np.array([oracle.round_f16(float(x)) for x in xs], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_f16_forward_close_to_f32_forward(oracle):
    from tests.util import make_policy_arrays, oracle_policy
    op = oracle_policy(oracle, make_policy_arrays(9, seed=4, emb=64, hidden=32))
    obs = [i * 9 + v for i, v in enumerate([3, 1, 2, 0, 4, 5, 6, 7, 8])]
    l32, v32 = op.forward(obs, [True] * 4, arith=oracle.ARITH_REF)
    l16, v16 = op.forward(obs, [True] * 4, arith=oracle.ARITH_F16)
    assert np.max(np.abs(np.asarray(l32) - np.asarray(l16))) < 5e-3 and abs(v32 - v16) < 5e-3
    assert not np.array_equal(np.asarray(l32, np.float32), np.asarray(l16, np.float32))


@pytest.mark.parametrize("conv_dim", [0, 1])
def test_conv1d_embedding_mode_matches_torch_conv1d(oracle, conv_dim):
    """The oracle's conv1d EmbeddingBag mode (rust/src/nn/layers.rs:63-77) against what Conv1dPolicy computes in torch
    (src/twisterl/nn/policy.py:233-247: [Transpose if conv_dim == 1] -> Conv1d(kernel 1, no bias) -> Transpose -> Flatten)
    on the one-hot observation the ids stand for -- an independent statement of the same layer."""
    import torch
    rng = np.random.default_rng(5)
    R, Cc, v, hidden = 4, 5, 6, 8
    shape = [R, Cc]
    n_in, n_slices = shape[conv_dim], shape[1 - conv_dim]
    conv_w = rng.standard_normal((v, n_in)).astype(np.float32)              # Conv1d.weight.squeeze(2)
    emb = n_slices * v
    w1 = rng.standard_normal((hidden, emb)).astype(np.float32) * 0.2; b1 = rng.standard_normal(hidden).astype(np.float32)
    wa = rng.standard_normal((3, hidden)).astype(np.float32) * 0.3; ba = rng.standard_normal(3).astype(np.float32)
    wv = rng.standard_normal((1, hidden)).astype(np.float32) * 0.3; bv = rng.standard_normal(1).astype(np.float32)
    pol = oracle.Policy(np.ascontiguousarray(conv_w.T), np.zeros(emb, dtype=np.float32),
                        [(np.ascontiguousarray(w1.T).reshape(-1), b1, True)], [(np.ascontiguousarray(wa.T).reshape(-1), ba, False)],
                        [(np.ascontiguousarray(wv.T).reshape(-1), bv, False)], (), (), obs_shape=shape, conv_dim=conv_dim)
    for trial in range(6):
        ids = sorted(rng.choice(R * Cc, size=5, replace=False).tolist())
        x = torch.zeros(1, R, Cc)
        for i in ids:
            x[0, i // Cc, i % Cc] = 1.0
        y = x.transpose(1, 2) if conv_dim == 1 else x
        y = torch.nn.functional.conv1d(y, torch.tensor(conv_w).unsqueeze(2)).transpose(1, 2).flatten(1)
        hdn = torch.relu(torch.relu(y) @ torch.tensor(w1).T + torch.tensor(b1))
        lg = (hdn @ torch.tensor(wa).T + torch.tensor(ba))[0].numpy()
        val = float((hdn @ torch.tensor(wv).T + torch.tensor(bv))[0, 0])
        lo, vo = pol.raw_predict(ids)
        np.testing.assert_allclose(lo, lg, atol=1e-5, rtol=1e-5)
        assert abs(vo - val) <= 1e-5


# ------------------------------------------------------------------ the reference's own trained policy
def test_reference_trained_policy_solves_the_oracle_puzzle(oracle):
    """examples/ppo_puzzle8_v1.pt was trained by the reference against the reference's Puzzle.  Exported the way to_rust()
    exports it and run through the oracle's evaluate (rl/evaluate.rs:22-89 restated), it solves every scrambled 3x3 board
    up to the config's diff_max = 32 -- which it only can if obs encoding (puzzle.rs:183-185), weight layout
    (nn/utils.py:17-59, layers.rs:26) and action -> direction map (puzzle.rs:135-160) ALL agree with the reference.
    Controls: an untrained policy, and the trained one behind a swapped action map, fail."""
    from tests.util import make_policy_arrays, trained_puzzle8_arrays
    arrs = trained_puzzle8_arrays()
    pol = oracle.Policy(*arrs)
    for diff in (8, 32):
        env = oracle.Puzzle(3, 3, diff, 2, 256)
        for det in (False, True):
            for arith in (oracle.ARITH_REF, oracle.ARITH_CHAIN):
                rate, rew = oracle.evaluate(env, pol, 200, det, 1, seed=1, arith=arith)
                assert rate >= 0.97 and rew > 0.9, (diff, det, arith, rate, rew)
    env = oracle.Puzzle(3, 3, 32, 2, 256)
    untrained = oracle.Policy(*make_policy_arrays(9, seed=0))
    assert oracle.evaluate(env, untrained, 200, False, 1, seed=1)[0] < 0.3
    # the same weights with left<->right swapped in the action head: the policy walks the blank the wrong way
    emb, eb, common, action, value = arrs
    wa = action[0][0].reshape(256, 4)[:, [2, 1, 0, 3]]
    swapped = oracle.Policy(emb, eb, common, [(np.ascontiguousarray(wa).reshape(-1), action[0][1][[2, 1, 0, 3]], False)], value)
    assert oracle.evaluate(env, swapped, 200, False, 1, seed=1)[0] < 0.5
    # and with the table rows of (cell, tile) read as (tile, cell): a transposed obs encoding
    T = np.arange(81).reshape(9, 9).T.reshape(-1)
    assert oracle.evaluate(env, oracle.Policy(np.ascontiguousarray(emb[T]), eb, common, action, value), 200, False, 1, seed=1)[0] < 0.5


def test_generic_env_collect_dummy_env_known_answer(oracle):
    """The reference's own collector test (rust/src/collector/ppo.rs:128-183): DummyEnv (final after one step, reward 1.0) and a
    one-weight policy: 1 episode => 2 records; with gamma 0.9, lambda 0.95, values [1, 1], rewards [1, 1]: rets [1.9, 1.0],
    advs [0.9, 0.0].  Through the oracle's collector for arbitrary environments (ppo_collect_env)."""
    class DummyEnv:
        def __init__(self):
            self.steps = 0
        def copy(self):
            c = DummyEnv(); c.steps = self.steps; return c
        def num_actions(self): return 1
        def obs_shape(self): return [1]
        def reset(self, difficulty): self.steps = 0
        def next(self, action): self.steps += 1
        def masks(self): return [True]
        def is_final(self): return self.steps >= 1
        def value(self): return 1.0
        def observe(self): return [0]
    pol = oracle.Policy(np.zeros((1, 1), np.float32), np.ones(1, np.float32), [], [(np.ones(1, np.float32), np.zeros(1, np.float32), False)],
                        [(np.ones(1, np.float32), np.zeros(1, np.float32), False)], emb_relu=False)
    d = oracle.ppo_collect_env(DummyEnv(), pol, 1, 0.9, 0.95, seed=0)
    assert d.obs.shape[0] == 2 and d.values.tolist() == [1.0, 1.0] and d.rewards.tolist() == [1.0, 1.0]
    assert np.allclose(d.additional_data["rets"], [1.9, 1.0]) and np.allclose(d.additional_data["advs"], [0.9, 0.0], atol=1e-7)
    assert d.actions.tolist() == [0, 0] and d.ep_len.tolist() == [2]


def test_generic_env_self_play_dummy_env_known_answer(oracle):
    """The reference's AZCollector test (rust/src/collector/az.rs:132-186): DummyEnv (one action, final after one step, reward
    1.0), a one-weight policy, 1 episode, 1 search => 2 records with `remaining_values`.  Through the oracle's self-play
    collector for arbitrary environments (az_collect_env); with one action every MCTS prob is 1 and the remaining values are
    [2, 1].  And on a Puzzle the restatement must agree with the oracle's native self-play collector, bit for bit."""
    class DummyEnv:
        def __init__(self):
            self.steps = 0
        def copy(self):
            c = DummyEnv(); c.steps = self.steps; return c
        def num_actions(self): return 1
        def obs_shape(self): return [1]
        def reset(self, difficulty): self.steps = 0
        def next(self, action): self.steps += 1
        def masks(self): return [True]
        def is_final(self): return self.steps >= 1
        def value(self): return 1.0
        def observe(self): return [0]
    pol = oracle.Policy(np.zeros((1, 1), np.float32), np.ones(1, np.float32), [], [(np.ones(1, np.float32), np.zeros(1, np.float32), False)],
                        [(np.ones(1, np.float32), np.zeros(1, np.float32), False)], emb_relu=False)
    oracle.set_det_exp(True)
    try:
        d = oracle.az_collect_env(DummyEnv(), pol, 1, 1, 1.0, 1, seed=0)
        assert d.obs.shape[0] == 2 and "remaining_values" in d.additional_data and d.ep_len.tolist() == [2]
        assert d.logits.reshape(-1).tolist() == [1.0, 1.0] and d.additional_data["remaining_values"].tolist() == [2.0, 1.0]

        # the same algorithm, two implementations: a Puzzle behind the Python env protocol against two_az_collect
        class PyPuzzle:
            def __init__(self, w, h, diff):
                self.p = oracle.Puzzle(w, h, diff, 2, 256); self.seed = (0, 0)
            def copy(self):
                c = PyPuzzle.__new__(PyPuzzle); c.p = self.p.clone(); c.seed = self.seed; return c
            def seed_episode(self, seed, episode): self.seed = (seed, episode)
            def num_actions(self): return 4
            def obs_shape(self): return self.p.obs_shape()
            def reset(self, difficulty): self.p.reset(seed=self.seed[0], episode=self.seed[1])
            def next(self, action): self.p.step(action)
            def masks(self): return self.p.masks()
            def is_final(self): return self.p.is_final()
            def value(self): return self.p.reward()
            def observe(self): return self.p.observe()
        from tests.util import make_policy_arrays, oracle_policy
        op = oracle_policy(oracle, make_policy_arrays(9, seed=3, emb=32, hidden=32, scale=3.0))
        for S, med in ((7, 1), (5, 2)):
            a = oracle.az_collect_env(PyPuzzle(3, 3, 3), op, 6, S, 1.41, med, seed=23)
            b = oracle.az_collect(oracle.Puzzle(3, 3, 3, 2, 256), op, 6, S, 1.41, med, seed=23, arith=oracle.ARITH_CHAIN, num_threads=2, det_math=True)
            assert np.array_equal(a.obs, b.obs) and np.array_equal(a.ep_len, b.ep_len)
            assert np.array_equal(a.logits.view(np.uint32), b.logits.view(np.uint32))
            assert np.array_equal(a.additional_data["remaining_values"].view(np.uint32), b.additional_data["remaining_values"].view(np.uint32))
        # evaluate / solve of the same Puzzle through the restatements for arbitrary environments against two_evaluate / two_solve
        PyPuzzle.success = lambda self: self.p.solved()
        f = lambda x: np.float32(x).view(np.uint32)
        for det, ns, S in ((True, 1, 0), (False, 3, 0), (False, 2, 4)):
            a = oracle.evaluate_env(PyPuzzle(3, 3, 4), op, 12, det, ns, S, 1.41, 1, seed=7)
            b = oracle.evaluate(oracle.Puzzle(3, 3, 4, 2, 256), op, 12, det, ns, num_mcts_searches=S, seed=7, Cc=1.41, max_expand_depth=1,
                                arith=oracle.ARITH_CHAIN, det_math=True)
            assert (f(a[0]), f(a[1])) == (f(b[0]), f(b[1])), (det, ns, S, a, b)
            start = PyPuzzle(3, 3, 4); start.seed_episode(9, 3); start.reset(0)
            nat = oracle.Puzzle(3, 3, 4, 2, 256); nat.set_state(start.p.get_state())
            start.p.set_state(start.p.get_state())                       # (set_state: depth = max_depth, as the native side)
            (s1, r1), a1 = oracle.solve_env(start, op, det, ns, S, 1.41, 1, seed=5)
            (s2, r2), a2 = oracle.solve(nat, op, det, ns, num_mcts_searches=S, Cc=1.41, max_expand_depth=1, seed=5, arith=oracle.ARITH_CHAIN, det_math=True)
            assert (s1, f(r1)) == (s2, f(r2)) and list(a1) == list(a2), (det, ns, S)
    finally:
        oracle.set_det_exp(False)
