"""ctypes front-end of the CPU ORACLE (oracle/tw_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), and the
cpu_baseline leg of bench.py.  The product package (twisterl_amd/) must never import this.

The oracle restates the reference algorithm (file:line citations live in tw_oracle.c); this
module only marshals numpy arrays in and out of it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libtw_oracle.so")

ARITH_REF = 0    # reference order: un-fused multiply/add, k-ordered, bias last
ARITH_CHAIN = 1  # fused-multiply-add chain (what an f32 MFMA computes), bias last
ARITH_F16 = 2    # f16-rounded weights/activations, wide accumulation, f32 biases (the HIP library's precision="fp16" mode)

MAX_CELLS = 64
MAX_LAYERS = 8


def build(force: bool = False) -> str:
    """Compile the oracle with gcc via oracle/Makefile (idempotent)."""
    src = [os.path.join(_HERE, f) for f in ("tw_oracle.c", "tw_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


class _Puzzle(C.Structure):
    _fields_ = [("state", C.c_int64 * MAX_CELLS), ("zx", C.c_int64), ("zy", C.c_int64),
                ("depth", C.c_int64), ("width", C.c_int64), ("height", C.c_int64),
                ("difficulty", C.c_int64), ("depth_slope", C.c_int64), ("max_depth", C.c_int64)]


class _Linear(C.Structure):
    _fields_ = [("in_", C.c_int), ("out", C.c_int), ("w", C.POINTER(C.c_float)),
                ("b", C.POINTER(C.c_float)), ("relu", C.c_int)]


class _EmbBag(C.Structure):
    _fields_ = [("n_vectors", C.c_int), ("vec_len", C.c_int), ("vectors", C.POINTER(C.c_float)),
                ("bias", C.POINTER(C.c_float)), ("bias_len", C.c_int), ("relu", C.c_int),
                ("obs_shape", C.c_int * 2), ("obs_ndim", C.c_int), ("conv_dim", C.c_int)]


class _Policy(C.Structure):
    _fields_ = [("emb", _EmbBag),
                ("common", _Linear * MAX_LAYERS), ("n_common", C.c_int),
                ("action", _Linear * MAX_LAYERS), ("n_action", C.c_int),
                ("value", _Linear * MAX_LAYERS), ("n_value", C.c_int),
                ("n_perms", C.c_int), ("obs_size", C.c_int), ("n_actions", C.c_int),
                ("obs_perms", C.POINTER(C.c_int32)), ("act_perms", C.POINTER(C.c_int32))]


class _Collected(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_cells", C.c_int), ("n_actions", C.c_int),
                ("obs", C.POINTER(C.c_int64)), ("logits", C.POINTER(C.c_float)),
                ("perms", C.POINTER(C.c_int32)), ("values", C.POINTER(C.c_float)),
                ("rewards", C.POINTER(C.c_float)), ("actions", C.POINTER(C.c_int64)),
                ("advs", C.POINTER(C.c_float)), ("rets", C.POINTER(C.c_float)),
                ("remaining", C.POINTER(C.c_float)), ("ep_len", C.POINTER(C.c_uint32)),
                ("n_episodes", C.c_uint64), ("has_ppo", C.c_int)]


class _PPOParams(C.Structure):
    _fields_ = [("num_episodes", C.c_uint64), ("episode_offset", C.c_uint64),
                ("gamma", C.c_float), ("lambda_", C.c_float), ("seed", C.c_uint64),
                ("arith", C.c_int), ("det_log", C.c_int), ("num_threads", C.c_int),
                ("merge_order", C.c_int)]


class _AZParams(C.Structure):
    _fields_ = [("num_episodes", C.c_uint64), ("episode_offset", C.c_uint64),
                ("num_mcts_searches", C.c_uint32), ("C", C.c_float),
                ("max_expand_depth", C.c_uint32), ("seed", C.c_uint64),
                ("arith", C.c_int), ("num_threads", C.c_int), ("merge_order", C.c_int),
                ("det_math", C.c_int)]


class _SolveParams(C.Structure):
    _fields_ = [("deterministic", C.c_int), ("num_searches", C.c_uint32), ("num_mcts_searches", C.c_uint32),
                ("C", C.c_float), ("max_expand_depth", C.c_uint32), ("seed", C.c_uint64), ("arith", C.c_int),
                ("det_math", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.two_logf_det.restype = C.c_float
        L.two_logf_det.argtypes = [C.c_float]
        L.two_expf_det.restype = C.c_float
        L.two_expf_det.argtypes = [C.c_float]
        L.two_puzzle_reward.restype = C.c_float
        L.two_puzzle_solved.restype = C.c_int
        L.two_puzzle_is_final.restype = C.c_int
        L.two_argmax.restype = C.c_int
        L.two_sample_from_logits.restype = C.c_int
        L.two_sample_weighted.restype = C.c_int
        L.two_sample_weighted.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float]
        L.two_ppo_collect.restype = C.c_int
        L.two_az_collect.restype = C.c_int
        L.two_gae.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_float,
                              C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.two_mcts_probs.argtypes = [C.POINTER(_Puzzle), C.POINTER(_Policy), C.c_uint32, C.c_float,
                                     C.c_uint32, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32,
                                     C.POINTER(C.c_float)]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ------------------------------------------------------------------------------------- RNG
def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
    o = (C.c_uint32 * 4)()
    lib().two_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def round_f16(x: float) -> float:
    """f32 -> binary16 (RNE) -> f32, as used by the ARITH_F16 forward"""
    f = lib().two_round_f16
    f.restype = C.c_float
    f.argtypes = [C.c_float]
    return float(f(C.c_float(x)))


def logf_det(x: float) -> float:
    return float(lib().two_logf_det(C.c_float(x)))


def expf_det(x: float) -> float:
    return float(lib().two_expf_det(C.c_float(x)))


def set_det_exp(on: bool) -> None:
    """masked softmax (predict/full_predict/MCTS priors) uses the deterministic exp when on"""
    lib().two_set_det_exp(C.c_int(int(on)))


def logf_det_array(x: np.ndarray) -> np.ndarray:
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    f = L.two_logf_det
    for i in range(x.size):
        out.flat[i] = f(C.c_float(float(x.flat[i])))
    return out


# ------------------------------------------------------------------------------------- env
class Puzzle:
    """Oracle twin of twisterl.env.Puzzle (python_interface/env.rs:117-160 over envs/puzzle.rs)."""

    def __init__(self, width, height, difficulty, depth_slope, max_depth):
        self.p = _Puzzle()
        lib().two_puzzle_new(C.byref(self.p), C.c_int64(width), C.c_int64(height),
                             C.c_int64(difficulty), C.c_int64(depth_slope), C.c_int64(max_depth))

    def clone(self):
        q = Puzzle.__new__(Puzzle)
        q.p = _Puzzle()
        C.memmove(C.byref(q.p), C.byref(self.p), C.sizeof(_Puzzle))
        return q

    @property
    def n_cells(self):
        return int(self.p.width * self.p.height)

    @property
    def difficulty(self):
        return int(self.p.difficulty)

    @difficulty.setter
    def difficulty(self, d):
        self.p.difficulty = int(d)

    @property
    def depth(self):
        return int(self.p.depth)

    @property
    def zero_location(self):
        return (int(self.p.zx), int(self.p.zy))

    def num_actions(self):
        return 4

    def obs_shape(self):
        return [self.n_cells, self.n_cells]

    def get_state(self):
        return [int(self.p.state[i]) for i in range(self.n_cells)]

    def solved(self):
        return bool(lib().two_puzzle_solved(C.byref(self.p)))

    def set_state(self, state):
        arr = (C.c_int64 * len(state))(*[int(s) for s in state])
        lib().two_puzzle_set_state(C.byref(self.p), arr, C.c_size_t(len(state)))

    def reset(self, seed=0, episode=0):
        lib().two_puzzle_reset(C.byref(self.p), C.c_uint64(seed), C.c_uint64(episode))

    def step(self, action):
        lib().two_puzzle_step(C.byref(self.p), C.c_int64(action))

    def masks(self):
        m = (C.c_uint8 * 4)()
        lib().two_puzzle_masks(C.byref(self.p), m)
        return [bool(x) for x in m]

    def is_final(self):
        return bool(lib().two_puzzle_is_final(C.byref(self.p)))

    def reward(self):
        return float(lib().two_puzzle_reward(C.byref(self.p)))

    def observe(self):
        o = (C.c_int64 * self.n_cells)()
        lib().two_puzzle_observe(C.byref(self.p), o)
        return [int(x) for x in o]


# ------------------------------------------------------------------------------------- nn
class Policy:
    """Oracle twin of twisterl.nn.Policy.  Weights use the reference's export layout
    (src/twisterl/nn/utils.py:17-79): Linear weights = torch_weight.T.flatten()
    ([in][out] row-major == column-major DMatrix(out,in)); EmbeddingBag vectors =
    torch_weight.T ([obs_size][emb])."""

    def __init__(self, emb_vectors, emb_bias, common, action, value, obs_perms=(), act_perms=(),
                 emb_relu=True, obs_shape=None, conv_dim=0):
        self._keep = []
        pol = _Policy()
        vec = np.ascontiguousarray(emb_vectors, dtype=np.float32)
        bias = np.ascontiguousarray(emb_bias, dtype=np.float32)
        self._keep += [vec, bias]
        pol.emb.n_vectors, pol.emb.vec_len = vec.shape
        pol.emb.vectors = _fp(vec)
        pol.emb.bias = _fp(bias)
        pol.emb.bias_len = bias.size
        pol.emb.relu = int(emb_relu)
        shape = list(obs_shape) if obs_shape is not None else [vec.shape[0]]
        pol.emb.obs_ndim = len(shape)
        for i, s in enumerate(shape[:2]):
            pol.emb.obs_shape[i] = int(s)
        pol.emb.conv_dim = int(conv_dim)

        def fill(dst, layers):
            for i, (w, b, relu) in enumerate(layers):
                w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1)
                b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
                self._keep += [w, b]
                dst[i].out = b.size
                dst[i].in_ = w.size // b.size
                dst[i].w = _fp(w)
                dst[i].b = _fp(b)
                dst[i].relu = int(relu)
            return len(layers)

        pol.n_common = fill(pol.common, common)
        pol.n_action = fill(pol.action, action)
        pol.n_value = fill(pol.value, value)
        pol.n_actions = int(np.asarray(action[-1][1]).size if len(action) else (len(act_perms[0]) if len(act_perms) else 0))
        pol.obs_size = int(vec.shape[0]) if len(shape) == 1 else int(np.prod(shape))
        pol.n_perms = len(obs_perms)
        if pol.n_perms:
            op = np.ascontiguousarray(obs_perms, dtype=np.int32)
            ap = np.ascontiguousarray(act_perms, dtype=np.int32)
            self._keep += [op, ap]
            pol.obs_perms = op.ctypes.data_as(C.POINTER(C.c_int32))
            pol.act_perms = ap.ctypes.data_as(C.POINTER(C.c_int32))
        self.pol = pol
        self.n_actions = pol.n_actions
        self.n_perms = pol.n_perms

    def _obs(self, obs):
        return (C.c_int64 * len(obs))(*[int(o) for o in obs]), len(obs)

    def _masks(self, masks):
        return (C.c_uint8 * len(masks))(*[1 if m else 0 for m in masks])

    def raw_predict(self, obs, perm=-1, arith=ARITH_REF):
        o, n = self._obs(obs)
        lg = (C.c_float * max(self.n_actions, 1))()
        v = C.c_float()
        lib().two_policy_raw_predict(C.byref(self.pol), o, C.c_int(n), C.c_int(perm), C.c_int(arith),
                                     lg, C.byref(v))
        return [float(x) for x in lg][: self.n_actions], float(v.value)

    def forward(self, obs, masks, perm=-1, arith=ARITH_REF):
        o, n = self._obs(obs)
        lg = (C.c_float * max(self.n_actions, 1))()
        v = C.c_float()
        lib().two_policy_forward(C.byref(self.pol), o, C.c_int(n), self._masks(masks), C.c_int(perm),
                                 C.c_int(arith), lg, C.byref(v))
        return [float(x) for x in lg][: self.n_actions], float(v.value)

    def predict(self, obs, masks, perm=-1, arith=ARITH_REF):
        o, n = self._obs(obs)
        pr = (C.c_float * max(self.n_actions, 1))()
        v = C.c_float()
        lib().two_policy_predict(C.byref(self.pol), o, C.c_int(n), self._masks(masks), C.c_int(perm),
                                 C.c_int(arith), pr, C.byref(v))
        return [float(x) for x in pr][: self.n_actions], float(v.value)

    def full_predict(self, obs, masks, arith=ARITH_REF):
        o, n = self._obs(obs)
        pr = (C.c_float * max(self.n_actions, 1))()
        v = C.c_float()
        lib().two_policy_full_predict(C.byref(self.pol), o, C.c_int(n), self._masks(masks),
                                      C.c_int(arith), pr, C.byref(v))
        return [float(x) for x in pr][: self.n_actions], float(v.value)


def argmax(values):
    a = np.ascontiguousarray(values, dtype=np.float32)
    return int(lib().two_argmax(_fp(a), C.c_int(a.size)))


def sample_from_logits(logits, u, det_log=False):
    a = np.ascontiguousarray(logits, dtype=np.float32)
    uu = np.ascontiguousarray(u, dtype=np.float32)
    return int(lib().two_sample_from_logits(_fp(a), C.c_int(a.size), _fp(uu), C.c_int(int(det_log))))


def sample_weighted(probs, u):
    a = np.ascontiguousarray(probs, dtype=np.float32)
    return int(lib().two_sample_weighted(_fp(a), C.c_int(a.size), C.c_float(u)))


def gae(rews, vals, gamma, lam):
    r = np.ascontiguousarray(rews, dtype=np.float32)
    v = np.ascontiguousarray(vals, dtype=np.float32)
    advs = np.empty_like(r)
    rets = np.empty_like(r)
    lib().two_gae(_fp(r), _fp(v), C.c_int(r.size), C.c_float(gamma), C.c_float(lam), _fp(advs), _fp(rets))
    return advs, rets


# ------------------------------------------------------------------------------------- collectors
@dataclass
class Collected:
    obs: np.ndarray        # [n, n_cells] int64
    logits: np.ndarray     # [n, A] f32
    perms: np.ndarray      # [n] int32 (-1 = None)
    values: np.ndarray     # [n] f32 (empty for AZ)
    rewards: np.ndarray
    actions: np.ndarray    # [n] int64
    additional_data: dict  # "advs","rets" (PPO) | "remaining_values" (AZ)
    ep_len: np.ndarray     # [E] uint32 in episode-index order


def _take(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros((0,), dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def _unpack(c: _Collected) -> Collected:
    n, nc, A, E = int(c.n), int(c.n_cells), int(c.n_actions), int(c.n_episodes)
    obs = _take(c.obs, n * nc, np.int64).reshape(n, nc)
    logits = _take(c.logits, n * A, np.float32).reshape(n, A)
    perms = _take(c.perms, n, np.int32)
    ep_len = _take(c.ep_len, E, np.uint32)
    if c.has_ppo:
        out = Collected(obs, logits, perms, _take(c.values, n, np.float32),
                        _take(c.rewards, n, np.float32), _take(c.actions, n, np.int64),
                        {"advs": _take(c.advs, n, np.float32), "rets": _take(c.rets, n, np.float32)},
                        ep_len)
    else:
        e = np.zeros((0,), np.float32)
        out = Collected(obs, logits, perms, e, e.copy(), np.zeros((0,), np.int64),
                        {"remaining_values": _take(c.remaining, n, np.float32)}, ep_len)
    lib().two_collected_free(C.byref(c))
    return out


def ppo_collect(env: Puzzle, policy: Policy, num_episodes, gamma, lam, seed=0, episode_offset=0,
                arith=ARITH_REF, det_log=False, num_threads=1, merge_order=True) -> Collected:
    prm = _PPOParams(num_episodes, episode_offset, gamma, lam, seed, arith, int(det_log),
                     num_threads, int(merge_order))
    out = _Collected()
    rc = lib().two_ppo_collect(C.byref(env.p), C.byref(policy.pol), C.byref(prm), C.byref(out))
    if rc != 0:
        raise RuntimeError("Something went wrong. No data in collected data chunks to merge. ")
    return _unpack(out)


def az_collect(env: Puzzle, policy: Policy, num_episodes, num_mcts_searches, Cc, max_expand_depth,
               seed=0, episode_offset=0, arith=ARITH_REF, num_threads=1, merge_order=True,
               det_math=False) -> Collected:
    prm = _AZParams(num_episodes, episode_offset, num_mcts_searches, Cc, max_expand_depth, seed,
                    arith, num_threads, int(merge_order), int(det_math))
    out = _Collected()
    rc = lib().two_az_collect(C.byref(env.p), C.byref(policy.pol), C.byref(prm), C.byref(out))
    if rc != 0:
        raise RuntimeError("Something went wrong. No data in collected data chunks to merge. ")
    return _unpack(out)


def mcts_probs(env: Puzzle, policy: Policy, num_mcts_searches, Cc, max_expand_depth,
               arith=ARITH_REF, seed=0, episode=0, t=0):
    pr = (C.c_float * policy.n_actions)()
    lib().two_mcts_probs(C.byref(env.p), C.byref(policy.pol), num_mcts_searches, Cc, max_expand_depth,
                         arith, seed, episode, t, pr)
    return [float(x) for x in pr]


def solve(env: Puzzle, policy: Policy, deterministic, num_searches, num_mcts_searches=0, Cc=1.41, max_expand_depth=1,
          seed=0, episode=0, arith=ARITH_REF, det_math=False):
    """rl/solve.rs:73-101 from the env's current state -> ((success, reward), actions)"""
    prm = _SolveParams(int(deterministic), num_searches, num_mcts_searches, Cc, max_expand_depth, seed, arith, int(det_math))
    acts = (C.c_int64 * (env.depth + 2))()
    s, r = C.c_float(), C.c_float()
    lib().two_solve.restype = C.c_int
    n = lib().two_solve(C.byref(env.p), C.byref(policy.pol), C.byref(prm), C.c_uint64(episode), C.byref(s), C.byref(r), acts)
    return (float(s.value), float(r.value)), [int(acts[i]) for i in range(n)]


def evaluate(env: Puzzle, policy: Policy, num_episodes, deterministic, num_searches, num_mcts_searches=0, seed=0, Cc=1.41,
             max_expand_depth=1, arith=ARITH_REF, det_math=False):
    """rl/evaluate.rs:22-89 -> (success_rate, mean_reward)"""
    prm = _SolveParams(int(deterministic), num_searches, num_mcts_searches, Cc, max_expand_depth, seed, arith, int(det_math))
    s, r = C.c_float(), C.c_float()
    lib().two_evaluate(C.byref(env.p), C.byref(policy.pol), C.byref(prm), C.c_uint64(num_episodes), C.byref(s), C.byref(r))
    return float(s.value), float(r.value)


def replay(env: Puzzle, actions):
    n = len(actions)
    nc = env.n_cells
    acts = np.ascontiguousarray(actions, dtype=np.int64)
    obs = np.empty((n + 1, nc), np.int64)
    masks = np.empty((n + 1, 4), np.uint8)
    rew = np.empty((n + 1,), np.float32)
    fin = np.empty((n + 1,), np.uint8)
    board = np.empty((n + 1, nc), np.int64)
    lib().two_replay(C.byref(env.p), acts.ctypes.data_as(C.POINTER(C.c_int64)), C.c_size_t(n),
                     obs.ctypes.data_as(C.POINTER(C.c_int64)), masks.ctypes.data_as(C.POINTER(C.c_uint8)),
                     _fp(rew), fin.ctypes.data_as(C.POINTER(C.c_uint8)),
                     board.ctypes.data_as(C.POINTER(C.c_int64)))
    return obs, masks.astype(bool), rew, fin.astype(bool), board


# ------------------------------------------------------------------------------------- any environment
def ppo_collect_env(proto, policy: Policy, num_episodes, gamma, lam, seed=0, episode_offset=0, difficulty=1, arith=ARITH_CHAIN,
                    merge_order=True) -> Collected:
    """PPOCollector::collect (rust/src/collector/ppo.rs:41-126) over ANY environment object with the reference's Python env
    protocol (python_interface/pyenv.rs: copy, reset(difficulty), next, masks, is_final, value, observe): clone + reset per
    episode (:59-60), per record observe / masks / reward of the current state, forward_with_perm, sample_from_logits, push,
    `if is_final break`, step (:69-80), GAE (:82-92), merge (collector.rs:40-46).  Randomness per the build's RNG spec (same
    streams as two_ppo_collect).  A pure-Python loop: small cases only."""
    A = policy.n_actions
    eps = []
    for i in range(num_episodes):
        e = proto.copy()
        ep = episode_offset + i
        if hasattr(e, "seed_episode"):
            e.seed_episode(seed, ep)
        e.reset(difficulty)
        obs_l, lg_l, perm_l, val_l, rew_l, act_l = [], [], [], [], [], []
        t = 0
        while True:
            obs = [int(x) for x in e.observe()]
            masks = [bool(m) for m in e.masks()]
            rew = np.float32(e.value())
            perm = -1
            if policy.n_perms > 0:
                w = philox4x32_10([ep & 0xFFFFFFFF, ep >> 32, t, 2], [seed & 0xFFFFFFFF, seed >> 32])
                perm = (w[0] * policy.n_perms) >> 32
            lg, v = policy.forward(obs, masks, perm=perm, arith=arith)
            u = []
            for a in range(A):
                w = philox4x32_10([ep & 0xFFFFFFFF, ep >> 32, t | ((a >> 2) << 24), 1], [seed & 0xFFFFFFFF, seed >> 32])
                u.append(np.float32(w[a & 3] >> 8) * np.float32(1.0 / 16777216.0))
            act = sample_from_logits(lg, u, det_log=True)
            obs_l.append(obs); lg_l.append(lg); perm_l.append(perm); val_l.append(v); rew_l.append(rew); act_l.append(act)
            if e.is_final():
                break
            e.next(act)
            t += 1
        advs, rets = gae(rew_l, val_l, gamma, lam)
        eps.append((obs_l, lg_l, perm_l, val_l, rew_l, act_l, advs, rets))
    order = ([num_episodes - 1] + list(range(num_episodes - 1))) if merge_order else list(range(num_episodes))
    cat = lambda k, dt: np.concatenate([np.asarray(eps[i][k], dtype=dt).reshape(len(eps[i][3]), -1) for i in order])
    return Collected(cat(0, np.int64), cat(1, np.float32), cat(2, np.int32).reshape(-1), cat(3, np.float32).reshape(-1),
                     cat(4, np.float32).reshape(-1), cat(5, np.int64).reshape(-1),
                     {"advs": cat(6, np.float32).reshape(-1), "rets": cat(7, np.float32).reshape(-1)},
                     np.asarray([len(e[3]) for e in eps], dtype=np.uint32))


def mcts_probs_env(env, policy: Policy, num_mcts_searches, C_, max_expand_depth, seed, key, t, arith=ARITH_CHAIN):
    """predict_probs_mcts (rust/src/rl/search.rs:104-189) over an environment object with the reference's Python env protocol:
    root full_predict, expand (a child per action with prior > 0: copy + next), `num_mcts_searches` times: descend by UCB
    (:29-39, first maximum), up to `max_expand_depth` times reward / is_final / full_predict / expand / next_sample,
    backpropagate; visit counts -> probs.  next_sample's draw: stream 4 | t << 8 of (seed, key), index it * max_expand_depth +
    expanded (the build's RNG spec).  Pure Python."""
    f32 = np.float32
    A = policy.n_actions
    root_state = env.copy()
    probs, _ = policy.full_predict([int(x) for x in root_state.observe()], [bool(m) for m in root_state.masks()], arith=arith)
    nodes = [dict(state=root_state, parent=-1, action=-1, prior=f32(0), visit=1, vsum=f32(0), children=[])]

    def expand(idx, pri):
        for a in range(A):
            if not (pri[a] > 0):
                continue
            st = nodes[idx]["state"].copy()
            st.next(a)
            nodes.append(dict(state=st, parent=idx, action=a, prior=f32(pri[a]), visit=0, vsum=f32(0), children=[]))
            nodes[idx]["children"].append(len(nodes) - 1)
    expand(0, probs)
    for it in range(num_mcts_searches):
        idx = 0
        while nodes[idx]["children"]:
            par, best, best_ucb = nodes[idx], None, f32(-np.inf)
            for c in par["children"]:
                ch = nodes[c]
                q = f32(0) if ch["visit"] == 0 else f32(ch["vsum"] / f32(ch["visit"]))
                d = f32(np.sqrt(f32(par["visit"])) / f32(f32(ch["visit"]) + f32(1)))
                d = f32(f32(C_) * d)
                d = f32(d * ch["prior"])
                ucb = f32(q + d)
                if ucb > best_ucb:
                    best, best_ucb = c, ucb
            idx = best
        value, expanded = f32(0), 0
        while expanded < max_expand_depth:
            st = nodes[idx]["state"]
            value = f32(st.value())
            if st.is_final():
                break
            pr, nv = policy.full_predict([int(x) for x in st.observe()], [bool(m) for m in st.masks()], arith=arith)
            expand(idx, pr)
            ch = nodes[idx]["children"]
            w = philox4x32_10([key & 0xFFFFFFFF, key >> 32, it * max_expand_depth + expanded, 4 | (t << 8)], [seed & 0xFFFFFFFF, seed >> 32])
            u = float(f32(w[0] >> 8) * f32(1.0 / 16777216.0))
            idx = ch[sample_weighted([nodes[c]["prior"] for c in ch], u)]
            value = f32(nv)
            expanded += 1
        j = idx
        while j >= 0:
            nodes[j]["vsum"] = f32(nodes[j]["vsum"] + value)
            nodes[j]["visit"] += 1
            j = nodes[j]["parent"]
    mp = np.zeros(A, dtype=np.float32)
    for c in nodes[0]["children"]:
        mp[nodes[c]["action"]] = f32(nodes[c]["visit"])
    sm = f32(0)
    for a in range(A):
        sm = f32(sm + mp[a])
    return (mp / sm).astype(np.float32) if sm > 0 else np.full(A, f32(1.0) / f32(A), dtype=np.float32)


def solve_env(env, policy: Policy, deterministic, num_searches, num_mcts_searches=0, C_=1.41, max_expand_depth=1, seed=0, episode=0,
              arith=ARITH_CHAIN):
    """solve (rust/src/rl/solve.rs:73-101) over single_solve (:17-71) from the CURRENT state of `env` (not modified): best of
    num_searches attempts by `(success, total) >` on the tuples.  Attempt k's draws are keyed episode * num_searches + k: twist of
    Policy::predict stream 2, action sample stream 5, index = the move (the build's RNG spec).  Pure Python."""
    f32 = np.float32
    best = ((0.0, float("-inf")), [])
    for k in range(num_searches):
        e = env.copy()
        key = episode * num_searches + k
        total, acts, t = f32(0), [], 0
        track = bool(e.track_solution()) if hasattr(e, "track_solution") else False     # solve.rs:28 (Env::track_solution, env.rs:62)
        while not e.is_final():
            total = f32(total + f32(e.value()))
            if num_mcts_searches == 0:
                perm = -1
                if policy.n_perms > 0:
                    w = philox4x32_10([key & 0xFFFFFFFF, key >> 32, t, 2], [seed & 0xFFFFFFFF, seed >> 32])
                    perm = (w[0] * policy.n_perms) >> 32
                probs = np.asarray(policy.predict([int(x) for x in e.observe()], [bool(m) for m in e.masks()], perm=perm, arith=arith)[0], dtype=np.float32)
            else:
                probs = mcts_probs_env(e, policy, num_mcts_searches, C_, max_expand_depth, seed, key, t, arith=arith)
            if deterministic:
                action, bv = 0, probs[0]
                for i in range(1, len(probs)):
                    if probs[i] > bv:
                        action, bv = i, probs[i]
            else:
                w = philox4x32_10([key & 0xFFFFFFFF, key >> 32, t, 5], [seed & 0xFFFFFFFF, seed >> 32])
                action = sample_weighted(probs, float(f32(w[0] >> 8) * f32(1.0 / 16777216.0)))
            e.next(action)
            if not track:                                   # solve.rs:57-59
                acts.append(action)
            t += 1
        if track:                                           # solve.rs:62-64: the environment's own record replaces the played actions
            acts = [int(x) for x in e.solution()]
        total = f32(total + f32(e.value()))
        val = ((1.0 if e.success() else 0.0, float(total)), acts)
        if val[0] > best[0]:
            best = val
    return best


def evaluate_env(proto, policy: Policy, num_episodes, deterministic, num_searches, num_mcts_searches=0, C_=1.41, max_expand_depth=1, seed=0,
                 difficulty=1, arith=ARITH_CHAIN):
    """evaluate (rust/src/rl/evaluate.rs:22-89): per episode reset a clone, solve, accumulate successes and rewards in episode
    order (f32), divide by the episode count.  Pure Python."""
    f32 = np.float32
    succ, rew = f32(0), f32(0)
    for ep in range(num_episodes):
        e = proto.copy()
        if hasattr(e, "seed_episode"):
            e.seed_episode(seed, ep)
        e.reset(difficulty)
        (s_, r_), _ = solve_env(e, policy, deterministic, num_searches, num_mcts_searches, C_, max_expand_depth, seed=seed, episode=ep, arith=arith)
        succ, rew = f32(succ + f32(s_)), f32(rew + f32(r_))
    return float(f32(succ / f32(num_episodes))), float(f32(rew / f32(num_episodes)))


def az_collect_env(proto, policy: Policy, num_episodes, num_mcts_searches, C_, max_expand_depth, seed=0, episode_offset=0,
                   difficulty=1, arith=ARITH_CHAIN, merge_order=True) -> Collected:
    """AZCollector::collect (rust/src/collector/az.rs:51-109) with predict_probs_mcts (rust/src/rl/search.rs:104-189) over ANY
    environment object with the reference's Python env protocol: per move root full_predict, expand (a child per action with
    prior > 0: copy + next), `num_mcts_searches` times: descend by UCB (:29-39, first maximum), up to `max_expand_depth` times
    reward / is_final / full_predict / expand / next_sample, backpropagate; visit counts -> probs; sample the action; record.
    Randomness per the build's RNG spec (streams of two_az_collect), exp of the soft-max deterministic (set_det_exp).  A pure-
    Python loop: small cases only.  Caller: set_det_exp(True) around the call."""
    f32 = np.float32
    A = policy.n_actions
    eps = []
    for i in range(num_episodes):
        env = proto.copy()
        ep = episode_offset + i
        if hasattr(env, "seed_episode"):
            env.seed_episode(seed, ep)
        env.reset(difficulty)
        obs_l, prob_l, val_l = [], [], []
        t = 0
        while True:
            mp = mcts_probs_env(env, policy, num_mcts_searches, C_, max_expand_depth, seed, ep, t, arith=arith)
            # az.rs:72-89
            w = philox4x32_10([ep & 0xFFFFFFFF, ep >> 32, t, 3], [seed & 0xFFFFFFFF, seed >> 32])
            action = sample_weighted(mp, float(f32(w[0] >> 8) * f32(1.0 / 16777216.0)))
            obs_l.append([int(x) for x in env.observe()]); prob_l.append(mp); val_l.append(f32(env.value()))
            if env.is_final():
                break
            env.next(action)
            t += 1
        total, before = f32(0), []
        for v in val_l:
            before.append(total)
            total = f32(total + v)
        eps.append((obs_l, prob_l, [f32(total - b) for b in before]))
    order = ([num_episodes - 1] + list(range(num_episodes - 1))) if merge_order else list(range(num_episodes))
    n_of = [len(e[2]) for e in eps]
    cat = lambda k, dt: np.concatenate([np.asarray(eps[i][k], dtype=dt).reshape(n_of[i], -1) for i in order])
    n = sum(n_of)
    return Collected(cat(0, np.int64), cat(1, np.float32), np.full(n, -1, dtype=np.int32), np.zeros(0, np.float32), np.zeros(0, np.float32),
                     np.zeros(0, np.int64), {"remaining_values": cat(2, np.float32).reshape(-1)}, np.asarray(n_of, dtype=np.uint32))
