"""GPU: the configurations the published numbers are quoted on, AT FULL SIZE, against the oracle.

* BASELINE config 3 exactly as bench.py runs it (Puzzle-15, 262,144 envs, difficulty 128, twists {identity,
  transpose}, BasicPolicy 256->512->256->4|1 with torch-default-init weights seed 0, merge order): the launch is the
  8-wave x 32-episode shape with persistent lanes and the episode queue -- the kernel the bench times.
* BASELINE config 5 at the reference's per-GPU batch (4,096 episodes x 100 and x 1,000 searches, Puzzle-15, the
  same policy).

The RNG is keyed by the GLOBAL episode index, so `oracle.*_collect(num_episodes=1, episode_offset=e)` is the oracle's
version of episode e of the big batch: sampled episodes are compared bit for bit on every field; size-independent
properties (determinism, record-count identities, the [E-1, 0, .., E-2] rotation) cover the rest.  The big buffers stay
on the device (torch views); only the sampled slices and the per-episode tables are copied to the host.
"""
import numpy as np
import pytest
import torch

from tests.util import f32_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tw():
    import twisterl_amd
    assert twisterl_amd.device_count() >= 1, "no GPU visible: the -m gpu tests need the MI355X box"
    return twisterl_amd.twisterl


def _bench_policy(oracle, twists):
    import bench
    arrs = bench.synthetic_weights(16, seed=0)
    op, ap = bench.transpose_twist(4) if twists else ([], [])
    return bench.build_policy(arrs, op, ap), oracle.Policy(*arrs, op, ap)


def _sample_episodes(E, n, seed):
    rng = np.random.default_rng(seed)
    return sorted(set([0, 1, E - 2, E - 1] + [int(x) for x in rng.choice(E, size=n, replace=False)]))


def test_config3_as_benched_full_size_vs_oracle(tw, oracle):
    import twisterl_amd
    gp, op = _bench_policy(oracle, twists=True)
    E, D = 262_144, 128
    env, oenv = tw.env.Puzzle(4, 4, D, 2, 256), oracle.Puzzle(4, 4, D, 2, 256)
    coll = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32})
    g = coll.collect(env, gp, seed=1000)
    # the launch shape of the bench: 8 waves x 32 episodes per workgroup, one persistent workgroup per CU
    assert g.stats["rollout_threads"] == 512
    assert g.stats["rollout_blocks"] == twisterl_amd.device_info()["compute_units"]
    t = g.to_torch()
    n = len(g)
    L = t["ep_len"].cpu().numpy().astype(np.int64)
    S = t["ep_start"].cpu().numpy().astype(np.int64)
    # record-count identities
    assert L.shape == (E,) and L.sum() == n == g.stats["records"] and L.min() >= 1 and L.max() <= 2 * D + 1
    for k, w in (("obs", 16), ("logits", 4)):
        assert tuple(t[k].shape) == (n, w)
    for k in ("perms", "values", "rewards", "actions", "advs", "rets"):
        assert tuple(t[k].shape) == (n,)
    # merge order [E-1, 0, 1, .., E-2] (collector.rs:40-46): episode E-1 first, then the index order
    order = np.concatenate([[E - 1], np.arange(E - 1)])
    assert np.array_equal(S[order], np.concatenate([[0], np.cumsum(L[order])[:-1]]))
    # integer sanity of the whole buffers on the device: actions < 4, twists in {0, 1}, obs id of cell c in [16c, 16c+16)
    assert int(t["actions"].max()) <= 3 and int(t["perms"].min()) >= 0 and int(t["perms"].max()) <= 1
    cell = torch.arange(16, device=t["obs"].device, dtype=torch.int32)[None, :]
    obs_i = t["obs"].to(torch.int32)
    assert bool(((obs_i >> 4) == cell).all())
    assert bool((obs_i & 15).sum(dim=1).eq(120).all())                    # every board is a permutation of 0..15
    # the terminal record of every episode: reward 1.0 (solved) or -0.5 (out of depth); every other record -0.5/256
    last = torch.from_numpy(S + L - 1).to(t["rewards"].device)
    r_last = t["rewards"][last]
    assert bool(((r_last == 1.0) | (r_last == -0.5)).all())
    assert int((t["rewards"] == np.float32(-0.5 / 256)).sum()) == n - E
    # determinism: the same seed again gives the same bytes (atomic episode queue: which lane runs an episode must not matter)
    h = coll.collect(env, gp, seed=1000)
    th = h.to_torch()
    for k in t:
        assert torch.equal(t[k], th[k]), k
    del h, th
    # sampled episodes against the oracle, every field, bit for bit
    for e in _sample_episodes(E, 36, seed=3):
        s, ln = int(S[e]), int(L[e])
        o = oracle.ppo_collect(oenv, op, 1, 0.995, 0.995, seed=1000, episode_offset=e, arith=oracle.ARITH_CHAIN, det_log=True,
                               merge_order=False)
        assert ln == int(o.ep_len[0]) == o.obs.shape[0], e
        sl = lambda k: t[k][s:s + ln].cpu().numpy()
        assert np.array_equal(sl("obs").astype(np.int64), o.obs), e
        assert np.array_equal(sl("actions").astype(np.int64), o.actions), e
        assert np.array_equal(sl("perms").astype(np.int32), o.perms), e
        assert np.array_equal(f32_bits(sl("rewards")), f32_bits(o.rewards)), e
        assert np.array_equal(f32_bits(sl("logits")), f32_bits(o.logits)), e
        assert np.array_equal(f32_bits(sl("values")), f32_bits(o.values)), e
        assert np.array_equal(f32_bits(sl("advs")), f32_bits(o.additional_data["advs"])), e
        assert np.array_equal(f32_bits(sl("rets")), f32_bits(o.additional_data["rets"])), e


@pytest.mark.parametrize("searches,n_sample", [(100, 28), (1000, 6)])
def test_config5_selfplay_full_size_vs_oracle(tw, oracle, searches, n_sample):
    """AZCollector at the reference's per-GPU batch: 4,096 episodes x {100, 1,000} MCTS searches per move on Puzzle-15 with the
    512/256 policy and the transpose twist (full_predict averages both twists), C = 1.41, max_expand_depth = 1
    (src/twisterl/defaults.py:84-91)."""
    gp, op = _bench_policy(oracle, twists=True)
    E, D = 4096, 8
    env, oenv = tw.env.Puzzle(4, 4, D, 2, 256), oracle.Puzzle(4, 4, D, 2, 256)
    coll = tw.collector.AZCollector(E, searches, 1.41, 1, 32)
    g = coll.collect(env, gp, seed=500)
    t = g.to_torch()
    n = len(g)
    L = t["ep_len"].cpu().numpy().astype(np.int64)
    S = t["ep_start"].cpu().numpy().astype(np.int64)
    assert L.sum() == n and L.min() >= 1 and L.max() <= 2 * D + 1
    order = np.concatenate([[E - 1], np.arange(E - 1)])
    assert np.array_equal(S[order], np.concatenate([[0], np.cumsum(L[order])[:-1]]))
    probs = t["logits"]
    assert bool((probs >= 0).all()) and bool(((probs.sum(dim=1) - 1.0).abs() < 1e-5).all())     # visit counts / sum
    assert bool((t["perms"] == -1).all())                                                     # az.rs:95
    import twisterl_amd
    cus = twisterl_amd.device_info()["compute_units"]
    assert g.stats["rollout_blocks"] == (cus - cus // 2) * (2 if searches < 400 else 1)
    # (the split shape: half of the CUs run engine workgroups; the walker kernel has two workgroups of twelve walker waves on each of the
    #  other CUs below 400 searches per move, one of sixteen from there on)
    assert g.stats["rollout_threads"] == (768 if searches < 400 else 1024)
    # one root evaluation + at most `searches` leaf evaluations per record, two twists each
    assert 2 * n <= g.stats["forward_evals"] <= 2 * n * (searches + 1)
    h = coll.collect(env, gp, seed=500).to_torch()
    for k in t:
        assert torch.equal(t[k], h[k]), k
    del h
    for e in _sample_episodes(E, n_sample, seed=5):
        s, ln = int(S[e]), int(L[e])
        o = oracle.az_collect(oenv, op, 1, searches, 1.41, 1, seed=500, episode_offset=e, arith=oracle.ARITH_CHAIN, merge_order=False,
                              det_math=True)
        assert ln == int(o.ep_len[0]), e
        sl = lambda k: t[k][s:s + ln].cpu().numpy()
        assert np.array_equal(sl("obs").astype(np.int64), o.obs), e
        assert np.array_equal(f32_bits(sl("logits")), f32_bits(o.logits)), e
        assert np.array_equal(f32_bits(sl("remaining_values")), f32_bits(o.additional_data["remaining_values"])), e


def test_reference_default_selfplay_512x1000_vs_oracle(tw, oracle):
    """The reference's own default (src/twisterl/defaults.py:84-88: 512 episodes x 1,000 searches) as bench.py's side entry runs it -- no
    twists, one walker per workgroup on long streaks of searches served from its table, two episodes per CU taken longest-looking
    first: determinism, the record identities, sampled episodes (among them the longest ones) bit-equal to the oracle."""
    import twisterl_amd
    gp, op = _bench_policy(oracle, twists=False)
    E, S_, D = 512, 1000, 8
    env, oenv = tw.env.Puzzle(4, 4, D, 2, 256), oracle.Puzzle(4, 4, D, 2, 256)
    coll = tw.collector.AZCollector(E, S_, 1.41, 1, 32)
    g = coll.collect(env, gp, seed=2)
    t = g.to_torch()
    L = t["ep_len"].cpu().numpy().astype(np.int64)
    S = t["ep_start"].cpu().numpy().astype(np.int64)
    assert L.sum() == len(g) and L.min() >= 1 and L.max() == 2 * D + 1
    cus = twisterl_amd.device_info()["compute_units"]
    assert (g.stats["rollout_blocks"], g.stats["rollout_threads"]) == (min(cus, E), 256)
    assert len(g) <= g.stats["forward_evals"] <= len(g) * (S_ + 1) and g.stats["reused_evals"] > g.stats["forward_evals"] // 2
    h = coll.collect(env, gp, seed=2).to_torch()
    for k in t:
        assert torch.equal(t[k], h[k]), k
    del h
    longest = [int(e) for e in np.argsort(-L, kind="stable")[:2]]
    for e in sorted(set(longest + _sample_episodes(E, 4, seed=9))):
        s, ln = int(S[e]), int(L[e])
        o = oracle.az_collect(oenv, op, 1, S_, 1.41, 1, seed=2, episode_offset=e, arith=oracle.ARITH_CHAIN, merge_order=False, det_math=True)
        assert ln == int(o.ep_len[0]), e
        sl = lambda k: t[k][s:s + ln].cpu().numpy()
        assert np.array_equal(sl("obs").astype(np.int64), o.obs), e
        assert np.array_equal(f32_bits(sl("logits")), f32_bits(o.logits)), e
        assert np.array_equal(f32_bits(sl("remaining_values")), f32_bits(o.additional_data["remaining_values"])), e


@pytest.mark.parametrize("E,searches,n_sample,threads", [(65_536, 32, 26, 512), (16_384, 100, 24, 256)])
def test_selfplay_lane_per_episode_kernel_full_size_vs_oracle(tw, oracle, request, E, searches, n_sample, threads):
    """`mcts_f32_kernel` (one lane pair per episode, tw_mcts.hip) at the sizes its numbers are quoted on: 65,536 x 32 (every
    episode resident: 256 workgroups of 8 waves x 32 episodes) and 16,384 x 100 (four waves share 32 episodes, one persistent
    workgroup per CU, episode queue), Puzzle-15, 512/256 policy, transpose twist.  A node whose move takes its parent's move
    back takes its grandparent's stored output (reused_evals): same bits as the evaluation the reference repeats
    (rust/src/rl/search.rs:132-164, collector/az.rs:51-109)."""
    import twisterl_amd
    from twisterl_amd import _lib
    gp, op = _bench_policy(oracle, twists=True)
    D = 8
    env, oenv = tw.env.Puzzle(4, 4, D, 2, 256), oracle.Puzzle(4, 4, D, 2, 256)
    coll = tw.collector.AZCollector(E, searches, 1.41, 1, 32)
    _lib.check(_lib.lib().tw_set_launch_option(_lib.TW_OPT_AZ_VARIANT, 2))       # this kernel, whatever the automatic choice is at this size
    request.addfinalizer(lambda: _lib.check(_lib.lib().tw_set_launch_option(_lib.TW_OPT_AZ_VARIANT, 0)))
    g = coll.collect(env, gp, seed=900)
    assert g.stats["rollout_threads"] == threads
    assert g.stats["rollout_blocks"] == twisterl_amd.device_info()["compute_units"]
    assert g.stats["speculative_evals"] == 0                      # (the walker kernel would report its look-ahead here)
    assert 0 < g.stats["reused_evals"] < g.stats["forward_evals"]
    assert _lib.debug_counters(13)[12] == 0                       # no stored output was ever offered for another board
    t = g.to_torch()
    n = len(g)
    L = t["ep_len"].cpu().numpy().astype(np.int64)
    S = t["ep_start"].cpu().numpy().astype(np.int64)
    assert L.sum() == n and L.min() >= 1 and L.max() <= 2 * D + 1
    order = np.concatenate([[E - 1], np.arange(E - 1)])
    assert np.array_equal(S[order], np.concatenate([[0], np.cumsum(L[order])[:-1]]))
    probs = t["logits"]
    assert bool((probs >= 0).all()) and bool(((probs.sum(dim=1) - 1.0).abs() < 1e-5).all())
    assert 2 * n <= g.stats["forward_evals"] <= 2 * n * (searches + 1)
    h = coll.collect(env, gp, seed=900).to_torch()                # determinism (the queue: which lanes run an episode must not matter)
    for k in t:
        assert torch.equal(t[k], h[k]), k
    del h
    with _lib.launch_option(_lib.TW_OPT_AZ_REUSE, 1):             # and the same bytes when every output comes out of a forward
        h = coll.collect(env, gp, seed=900)
        assert h.stats["reused_evals"] == 0
        h = h.to_torch()
    for k in t:
        assert torch.equal(t[k], h[k]), k
    del h
    for e in _sample_episodes(E, n_sample - 4, seed=11):
        s, ln = int(S[e]), int(L[e])
        o = oracle.az_collect(oenv, op, 1, searches, 1.41, 1, seed=900, episode_offset=e, arith=oracle.ARITH_CHAIN, merge_order=False,
                              det_math=True)
        assert ln == int(o.ep_len[0]), e
        sl = lambda k: t[k][s:s + ln].cpu().numpy()
        assert np.array_equal(sl("obs").astype(np.int64), o.obs), e
        assert np.array_equal(f32_bits(sl("logits")), f32_bits(o.logits)), e
        assert np.array_equal(f32_bits(sl("remaining_values")), f32_bits(o.additional_data["remaining_values"])), e


@pytest.mark.parametrize("transport", ["torch", "cabi"])
def test_config4_rank_share_full_size(tw, oracle, transport):
    """BASELINE config 4's PER-RANK code path exactly as `bench.py --gpus 8` runs it on each GPU, at world 1: 262,144 envs of
    Puzzle-15 (difficulty 128, twists) collected by `collect_sharded(step_episodes=(CUs-8)*256, reserve_cus=8,
    max_episode_records=257)` -- five pipeline steps, the persistent grid on CUs-8 compute units in all but the last step, each
    step's chunk received at its final offset in the pre-allocated result (the torch.distributed transport bench.py defaults to,
    and the library's own tw_gather_* over RCCL that TW_GATHER=cabi selects).  The merged result must be BYTE-EQUAL to the
    un-sharded collect of the same seed, and sampled episodes bit-equal to the oracle (reference merge order, collector.rs:40-46)."""
    import os
    import torch.distributed as dist
    import twisterl_amd
    from twisterl_amd.dist import Comm, collect_sharded, pipeline_steps
    gp, op = _bench_policy(oracle, twists=True)
    E, D = 262_144, 128
    t_max = 2 * D + 1
    env, oenv = tw.env.Puzzle(4, 4, D, 2, 256), oracle.Puzzle(4, 4, D, 2, 256)
    coll = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32})
    cus = twisterl_amd.device_info()["compute_units"]
    reserve, step_eps = 8, (cus - 8) * 256
    assert pipeline_steps(E, 1, 1, step_eps) == 5
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        comm = Comm() if transport == "cabi" else None
        merged, parts = collect_sharded(coll, env, gp, seed=1000, dst=0, max_episode_records=t_max, reserve_cus=reserve,
                                        step_episodes=step_eps, comm=comm)
        # five steps; all but the last leave 8 CUs to the transfer kernels
        assert [p.stats["episodes"] for p in parts] == [step_eps] * 4 + [E - 4 * step_eps]
        assert [p.stats["rollout_blocks"] for p in parts] == [cus - reserve] * 4 + [cus]
        # 248 x 256 episodes = every lane of the 8-wave shape resident once; the 8,192 left over: 256 workgroups of 32 episodes
        assert [p.stats["rollout_threads"] for p in parts] == [512] * 4 + [256]
        n = sum(len(p) for p in parts)
        assert merged["obs"].shape[0] == n
        del parts
        whole = coll.collect(env, gp, seed=1000)
        w = whole.to_torch()
        assert len(whole) == n
        for k in ("obs", "logits", "perms", "values", "rewards", "actions", "advs", "rets"):
            assert torch.equal(merged[k], w[k]), k
        L = w["ep_len"].cpu().numpy().astype(np.int64)
        S = w["ep_start"].cpu().numpy().astype(np.int64)
        if transport == "cabi":                      # the library's gather also rebuilds the per-episode tables of the merged order
            assert torch.equal(merged["ep_len"].to(w["ep_len"].dtype), w["ep_len"]) and torch.equal(merged["ep_start"].to(w["ep_start"].dtype), w["ep_start"])
        order = np.concatenate([[E - 1], np.arange(E - 1)])
        assert np.array_equal(S[order], np.concatenate([[0], np.cumsum(L[order])[:-1]]))
        del whole, w
        # sampled episodes (among them the first and last of every step) against the oracle, every field, bit for bit
        edges = [e for s in range(1, 5) for e in (s * step_eps - 1, s * step_eps)]
        for e in sorted(set(edges + _sample_episodes(E, 22, seed=4))):
            s, ln = int(S[e]), int(L[e])
            o = oracle.ppo_collect(oenv, op, 1, 0.995, 0.995, seed=1000, episode_offset=e, arith=oracle.ARITH_CHAIN, det_log=True,
                                   merge_order=False)
            assert ln == int(o.ep_len[0]) == o.obs.shape[0], e
            sl = lambda k: merged[k][s:s + ln].cpu().numpy()
            assert np.array_equal(sl("obs").astype(np.int64), o.obs), e
            assert np.array_equal(sl("actions").astype(np.int64), o.actions), e
            assert np.array_equal(sl("perms").astype(np.int32), o.perms), e
            assert np.array_equal(f32_bits(sl("rewards")), f32_bits(o.rewards)), e
            assert np.array_equal(f32_bits(sl("logits")), f32_bits(o.logits)), e
            assert np.array_equal(f32_bits(sl("values")), f32_bits(o.values)), e
            assert np.array_equal(f32_bits(sl("advs")), f32_bits(o.additional_data["advs"])), e
            assert np.array_equal(f32_bits(sl("rets")), f32_bits(o.additional_data["rets"])), e
        if comm is not None:
            comm.close()
    finally:
        dist.destroy_process_group()


# ----------------------------------------------------------------- bench.py's side entries, at exactly the shapes it times
def test_bench_side_entry_config1_puzzle8_1k_whole_collect_vs_oracle(tw, oracle):
    """BASELINE config 1's batch on the GPU as bench.side_configs times it ("config1_puzzle8_1k_f32"): Puzzle-8
    (examples/ppo_puzzle8_v1.json:3-9,21-26: 3x3, depth_slope 2, max_depth 256, gamma = lambda = 0.995) at difficulty 32, 1,024
    envs, BasicPolicy 81->512->256->4|1 with torch-default-init weights seed 0, no twists, f32.  The launch is the 16-episode
    workgroup shape (64 workgroups of four waves, Engine3T); the WHOLE collect (every record, every field) is bit-equal to the
    oracle, for each of the seeds the bench times."""
    import bench
    arrs = bench.synthetic_weights(9, seed=0)
    gp, op = bench.build_policy(arrs, [], []), oracle.Policy(*arrs, [], [])
    E, D = 1024, 32
    env, oenv = tw.env.Puzzle(3, 3, D, 2, 256), oracle.Puzzle(3, 3, D, 2, 256)
    coll = tw.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32}, precision="fp32")
    for seed in (2, 3, 4):
        g = coll.collect(env, gp, seed=seed)
        assert (g.stats["rollout_blocks"], g.stats["rollout_threads"]) == (64, 256)
        o = oracle.ppo_collect(oenv, op, E, 0.995, 0.995, seed=seed, arith=oracle.ARITH_CHAIN, det_log=True, num_threads=8, merge_order=True)
        a = g.to_numpy()
        assert len(g) == o.obs.shape[0] and 40_000 < len(g) <= E * (2 * D + 1)
        assert np.array_equal(a["ep_len"], o.ep_len)
        assert np.array_equal(a["obs"].astype(np.int64), o.obs)
        assert np.array_equal(a["actions"].astype(np.int64), o.actions)
        assert np.array_equal(a["perms"].astype(np.int32), o.perms) and np.all(a["perms"] == -1)
        for k, ref in (("rewards", o.rewards), ("logits", o.logits), ("values", o.values), ("advs", o.additional_data["advs"]),
                       ("rets", o.additional_data["rets"])):
            assert np.array_equal(f32_bits(a[k]), f32_bits(ref)), (seed, k)


def test_bench_side_entry_default_evaluations_vs_oracle(tw, oracle):
    """The four `evaluations_100_episodes` numbers bench.side_configs prints (the reference's default evaluations,
    src/twisterl/defaults.py:27-57, run beside every collect by learn_step, src/twisterl/rl/algorithm.py:117-121) on the
    Puzzle-15 512/256 seed-0 policy at difficulty 8, 100 episodes, seed 0: (success_rate, mean_reward) bit-equal to the oracle's
    evaluate (rust/src/rl/evaluate.rs:22-89) for the same draws."""
    import bench
    arrs = bench.synthetic_weights(16, seed=0)
    gp, op = bench.build_policy(arrs, [], []), oracle.Policy(*arrs, [], [])
    env, oenv = tw.env.Puzzle(4, 4, 8, 2, 256), oracle.Puzzle(4, 4, 8, 2, 256)
    seen = {}
    for name, det, ns, S in (("ppo_deterministic", True, 1, 0), ("ppo_1", False, 1, 0), ("ppo_10", False, 10, 0), ("mcts_100", True, 1, 100)):
        g = tw.collector.evaluate(env, gp, num_episodes=100, deterministic=det, num_searches=ns, num_mcts_searches=S, seed=0, C=1.41,
                                  max_expand_depth=1, num_cores=32)
        o = oracle.evaluate(oenv, op, 100, det, ns, num_mcts_searches=S, seed=0, Cc=1.41, max_expand_depth=1, arith=oracle.ARITH_CHAIN, det_math=True)
        assert f32_bits(g[0]) == f32_bits(o[0]) and f32_bits(g[1]) == f32_bits(o[1]), (name, g, o)
        assert 0.0 <= g[0] <= 1.0
        seen[name] = g
    # the searches help on this policy (what the bench line shows): best-of-10 and MCTS-guided solve more than one greedy attempt
    assert seen["ppo_10"][0] > seen["ppo_1"][0] and seen["mcts_100"][0] > seen["ppo_deterministic"][0]
