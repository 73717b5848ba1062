"""CPU: host-side logic of the product (the Puzzle host object behind the C ABI, the
CollectedData container, collector constructors) against the reference's known answers and the
oracle.  Nothing here launches a kernel."""
import json
import os

import numpy as np
import pytest

from twisterl_amd import twisterl

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


# ------------------------------------------------------------------ env surface (python_interface/env.rs)
def test_puzzle_known_answers():
    g = GOLD["puzzle_2x2_step2"]
    p = twisterl.env.Puzzle(*g["ctor"])
    assert p.solved() and p.depth == 1
    p.step(g["action"])
    assert p.masks() == g["masks"] and p.get_state() == [1, 0, 2, 3]
    assert p.num_actions() == 4 and p.obs_shape() == [4, 4] and p.twists() == ([], [])
    assert isinstance(p.__extract_env__(), int)


def test_puzzle_replay_123(oracle):
    g = GOLD["replay_123"]
    p = twisterl.env.Puzzle(*g["ctor"])
    p.set_state(g["start"])
    assert p.depth == 256
    for a in g["actions"]:
        assert not p.is_final() and p.masks()[a] and p.reward() == -0.5 / 256
        p.step(a)
    assert p.get_state() == g["end"] and p.solved() and p.is_final() and p.reward() == 1.0
    assert p.observe() == [i * 9 + i for i in range(9)]


@pytest.mark.parametrize("w,h", [(3, 3), (4, 4), (2, 3), (5, 5)])
def test_puzzle_matches_oracle_on_random_walks(oracle, w, h):
    rng = np.random.default_rng(w * 10 + h)
    a = twisterl.env.Puzzle(w, h, 9, 2, 64)
    b = oracle.Puzzle(w, h, 9, 2, 64)
    a.reset(seed=3, episode=17)
    b.reset(seed=3, episode=17)
    assert a.get_state() == b.get_state() and a.depth == b.depth == 18
    for _ in range(80):
        act = int(rng.integers(4))
        a.step(act); b.step(act)
        assert a.get_state() == b.get_state() and a.masks() == b.masks() and a.observe() == b.observe()
        assert a.reward() == b.reward() and a.is_final() == b.is_final() and a.depth == b.depth
    a.difficulty = 3
    assert a.difficulty == 3
    a.set_position(0, 0, 7)
    assert a.get_position(0, 0) == 7
    with pytest.raises(OverflowError):
        twisterl.env.Puzzle(3, 3, -1, 2, 256)


# ------------------------------------------------------------------ CollectedData (python_interface/collector.rs:24-137)
def test_collected_data_ctor_and_defaults():
    d = twisterl.collector.CollectedData([[0, 5]], [[0.1, 0.2]], [0.5], [1.0], [3])      # (obs, logits, values, rewards, actions)
    assert d.obs == [[0, 5]] and d.values == [0.5] and d.rewards == [1.0] and d.actions == [3]
    assert d.perms == [-1] and d.additional_data == {}                                    # perms default: None -> -1
    d2 = twisterl.collector.CollectedData([[1]], [[0.4]], [0.5], [0.6], [0], perms=[2])
    assert d2.perms == [2]
    d2.perms = [-5]
    assert d2.perms == [-1]


def test_collected_data_merge_known_answer():
    """collector.rs:101-126: merge appends; merging chunk d1 into d2 gives actions [0, 1]."""
    d1 = twisterl.collector.CollectedData([[0]], [[0.1]], [0.2], [0.3], [1], perms=[0])
    d2 = twisterl.collector.CollectedData([[1]], [[0.4]], [0.5], [0.6], [0])
    d2.set_additional_data_item("rets", [1.0])
    d1.set_additional_data_item("rets", [2.0])
    d1.set_additional_data_item("advs", [3.0])
    d2.merge(d1)
    assert d2.actions == GOLD["merge_order"]["merged_actions"] and d2.obs == [[1], [0]] and d2.perms == [-1, 0]
    assert d2.additional_data == {"rets": [1.0, 2.0], "advs": [3.0]}
    assert d2.get_additional_data_item("advs") == [3.0] and d2.get_additional_data_item("nope") is None
    # getters clone: mutating the returned list must not change the container
    d2.obs.append([9])
    assert len(d2.obs) == 2
    d2.values = [1, 2]
    assert d2.values == [1.0, 2.0]


def test_collector_constructors():
    cfg = {"num_cores": 32, "num_episodes": 1024, "lambda": 0.995, "gamma": 0.995}      # examples/ppo_puzzle8_v1.json:21-26
    c = twisterl.collector.PPOCollector(**cfg)
    assert (c.num_episodes, c.gamma, c.lambda_, c.num_cores) == (1024, 0.995, 0.995, 32)
    c = twisterl.collector.PPOCollector(8, 0.9, 0.95, 1)
    assert c.lambda_ == 0.95
    with pytest.raises(TypeError):
        twisterl.collector.PPOCollector(8, 0.9, 0.95)
    with pytest.raises(TypeError):
        twisterl.collector.PPOCollector(8, 0.9, 0.95, 1, bogus=1)
    with pytest.raises(OverflowError):
        twisterl.collector.PPOCollector(-1, 0.9, 0.95, 1)
    a = twisterl.collector.AZCollector(num_episodes=512, num_mcts_searches=1000, C=1.41, max_expand_depth=1, num_cores=32)
    assert a.num_mcts_searches == 1000
    with pytest.raises(TypeError):      # AZ_CONFIG's stray "seed" positional-style misuse is still a TypeError for unknown kwargs
        twisterl.collector.AZCollector(512, 1000, 1.41, 1, 32, bogus=2)


def test_collect_rejects_foreign_envs_and_policies():
    from tests.util import amd_policy, make_policy_arrays
    pol = amd_policy(make_policy_arrays(9, emb=32, hidden=32))
    c = twisterl.collector.PPOCollector(4, 0.9, 0.9, 1)
    with pytest.raises(TypeError, match="Object must implement __extract_env__ method"):   # env.rs:168-170
        c.collect(object(), pol)

    class Fake:
        def __extract_env__(self):
            return 1234
    with pytest.raises(TypeError, match="Expected environment of type"):
        c.collect(Fake(), pol)
    with pytest.raises(TypeError):
        c.collect(twisterl.env.Puzzle(3, 3, 1, 2, 256), object())


def test_policy_constructor_validation():
    with pytest.raises(ValueError):
        twisterl.nn.Linear([1.0, 2.0, 3.0], [0.0, 0.0], False)
    lin = twisterl.nn.Linear([1.0, 2.0, 3.0, 4.0], [1.0, 1.0], False)      # layers.rs:98-103 layout
    assert (lin.in_features, lin.out_features) == (2, 2)
    with pytest.raises(TypeError):
        twisterl.nn.Sequential([object()])
    emb = twisterl.nn.EmbeddingBag([[1.0, 2.0], [3.0, 4.0]], [0.0, 0.0], False, [2], 0)
    with pytest.raises(ValueError):
        twisterl.nn.Policy(emb, twisterl.nn.Sequential([]), twisterl.nn.Sequential([lin]), twisterl.nn.Sequential([lin]),
                           [[0, 1]], [])


def test_precision_kwarg_values():
    """precision= is a build extension of the collector constructors: the three rollout modes are accepted by name,
    anything else is a ValueError at construction (no GPU needed)."""
    from twisterl_amd import twisterl, _lib
    for name, code in (("fp32", 0), ("fp16", 1), ("fp16x2", 2)):
        c = twisterl.collector.PPOCollector(**{"num_episodes": 4, "gamma": 0.9, "lambda": 0.9, "num_cores": 1}, precision=name)
        assert _lib.PRECISIONS[c.precision] == code
    import pytest
    with pytest.raises(ValueError):
        twisterl.collector.PPOCollector(4, 0.9, 0.9, 1, precision="bf16")


def test_conv1d_embeddingbag_dense_table_matches_reference_rule():
    """EmbeddingBag conv1d mode (rust/src/nn/layers.rs:63-77): id i -> (row, col) of obs_shape (swapped for conv_dim 1);
    out[col*v:(col+1)*v] += vectors[row].  The dense table the kernels gather from must reproduce that for every id."""
    import numpy as np
    from twisterl_amd import twisterl
    rng = np.random.default_rng(0)
    for conv_dim, shape in ((0, [3, 5]), (1, [3, 5]), (0, [4, 4])):
        v = 6
        vec = rng.standard_normal((shape[conv_dim], v)).astype(np.float32)
        n_slices = shape[1 - conv_dim]
        eb = twisterl.nn.EmbeddingBag(vec.tolist(), [0.0] * (n_slices * v), True, shape, conv_dim)
        t = eb.dense_table()
        assert t.shape == (shape[0] * shape[1], n_slices * v)
        for i in range(shape[0] * shape[1]):
            row, col = divmod(i, shape[1])
            if conv_dim == 1:
                row, col = col, row
            want = np.zeros(n_slices * v, dtype=np.float32)
            want[col * v:(col + 1) * v] = vec[row]
            assert np.array_equal(t[i], want)
    import pytest
    with pytest.raises(ValueError):
        twisterl.nn.EmbeddingBag(vec.tolist(), [0.0] * 5, True, [4, 4], 0).dense_table()
    with pytest.raises(ValueError):
        twisterl.nn.EmbeddingBag(vec.tolist(), [0.0], True, [4, 4, 4], 0)


def test_pyenv_bridge_refuses_observations_that_do_not_fit_the_buffers():
    """The C side hands the env callbacks buffers of n_obs ids / num_actions flags (tw_env_vtable): an observe() of another
    length, an id outside obs_shape or a masks() of another length must raise instead of writing past the buffer or leaving
    stale ids in it (the callbacks are exercised directly: no device needed)."""
    import ctypes as C
    from tests.gridworld_env import GridWorld
    from twisterl_amd.collector import _PyEnvBridge
    from twisterl_amd.env import PyEnv

    class Odd(GridWorld):
        mode = "ok"

        def copy(self):
            c = Odd(self.width, self.height, self.max_steps)
            c.steps_left, c.agent, c.goal, c.trap, c.mode = self.steps_left, self.agent, self.goal, self.trap, self.mode
            return c

        def observe(self):
            o = super().observe()
            return {"long": o + [0], "short": o[:-1], "range": o[:-1] + [10 ** 6]}.get(self.mode, o)

        def masks(self):
            m = super().masks()
            return m + [True] if self.mode == "masks" else m

    for mode, exc in (("ok", None), ("long", ValueError), ("short", ValueError), ("range", IndexError), ("masks", ValueError)):
        env = Odd(3, 3, 5)
        br = _PyEnvBridge(PyEnv(env))
        n_obs = br.vt.n_obs
        env.mode = mode                                             # the prototype (handle 1) now misbehaves
        obs = (C.c_int32 * (n_obs + 4))(*([-7] * (n_obs + 4)))
        msk = (C.c_uint8 * 8)(*([9] * 8))
        br.vt.observe(1, obs)
        br.vt.masks(1, msk)
        assert list(obs[n_obs:]) == [-7] * 4 and list(msk[4:]) == [9] * 4      # nothing was written past the buffers
        if exc is None:
            assert not br.err and all(0 <= v < 81 for v in obs[:n_obs])
            br.finish(0)
        else:
            assert br.err and isinstance(br.err[0], exc)
            with pytest.raises(exc):
                br.finish(0)
