"""`twisterl.collector` surface: CollectedData, PPOCollector, AZCollector.

Mirrors reference rust/src/python_interface/collector.rs:24-187.  `collect(env, policy)` runs
the whole episode collection on the GPU (tw_ppo_collect / tw_az_collect); the result stays
device-resident and is only turned into Python lists when a reference-style attribute
(`.obs`, `.logits`, ...) is read.  `.device_arrays()` / `.to_torch()` expose the same buffers
zero-copy for trainers that can take tensors.
"""
from __future__ import annotations

import ctypes as C
import ctypes as _c

import numpy as np

from . import _lib
from .env import PyEnv, get_env_desc
from .nn import Policy

ctypes_u8 = _c.c_uint8

_FIELD_DTYPES = {
    _lib.TW_F_OBS: np.uint8, _lib.TW_F_LOGITS: np.float32, _lib.TW_F_PERMS: np.int8,
    _lib.TW_F_VALUES: np.float32, _lib.TW_F_REWARDS: np.float32, _lib.TW_F_ACTIONS: np.uint8,
    _lib.TW_F_ADVS: np.float32, _lib.TW_F_RETS: np.float32, _lib.TW_F_REMAINING: np.float32,
    _lib.TW_F_EP_LEN: np.uint32, _lib.TW_F_EP_START: np.uint64,
}
_FIELD_NAMES = {
    "obs": _lib.TW_F_OBS, "logits": _lib.TW_F_LOGITS, "perms": _lib.TW_F_PERMS, "values": _lib.TW_F_VALUES,
    "rewards": _lib.TW_F_REWARDS, "actions": _lib.TW_F_ACTIONS, "advs": _lib.TW_F_ADVS, "rets": _lib.TW_F_RETS,
    "remaining_values": _lib.TW_F_REMAINING, "ep_len": _lib.TW_F_EP_LEN, "ep_start": _lib.TW_F_EP_START,
}


class _DeviceResult:
    """Owner of a tw_collected handle."""

    def __init__(self, handle):
        self.h = handle
        L = _lib.lib()
        self.n = int(L.tw_collected_num_records(handle))
        self.n_episodes = int(L.tw_collected_num_episodes(handle))
        self.n_cells = int(L.tw_collected_num_cells(handle))
        self.n_actions = int(L.tw_collected_num_actions(handle))
        self.is_ppo = bool(L.tw_collected_is_ppo(handle))
        self.obs_width = int(L.tw_collected_obs_width(handle))      # 2: an environment with more than 256 obs ids
        st = _lib.CollectStats()
        _lib.check(L.tw_collected_stats(handle, C.byref(st)))
        self.stats = {k: getattr(st, k) for k, _ in st._fields_}

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                _lib.lib().tw_collected_free(h)
            except Exception:
                pass

    def shape(self, field):
        if field == _lib.TW_F_OBS:
            return (self.n, self.n_cells)
        if field == _lib.TW_F_LOGITS:
            return (self.n, self.n_actions)
        if field in (_lib.TW_F_EP_LEN, _lib.TW_F_EP_START):
            return (self.n_episodes,)
        return (self.n,)

    def dtype(self, field):
        return np.uint16 if (field == _lib.TW_F_OBS and self.obs_width == 2) else _FIELD_DTYPES[field]

    def ptr(self, field):
        nbytes = C.c_size_t()
        p = _lib.lib().tw_collected_device_ptr(self.h, field, C.byref(nbytes))
        return (int(p) if p else 0), int(nbytes.value)

    def host(self, field) -> np.ndarray:
        p, nbytes = self.ptr(field)
        dt = np.dtype(self.dtype(field))
        if nbytes == 0:
            return np.zeros((0,), dt)
        out = np.empty(self.shape(field), dt)
        assert out.nbytes == nbytes, (field, out.nbytes, nbytes)
        _lib.check(_lib.lib().tw_collected_copy_to_host(self.h, field, out.ctypes.data_as(C.c_void_p), nbytes))
        return out


class DeviceArray:
    """Zero-copy view of one result field: implements __cuda_array_interface__ (which PyTorch-ROCm
    accepts: torch.as_tensor(view, device='cuda'))."""

    def __init__(self, owner: _DeviceResult, field: int):
        self._owner = owner
        ptr, nbytes = owner.ptr(field)
        self.shape = owner.shape(field)
        self.dtype = np.dtype(owner.dtype(field))
        self.nbytes = nbytes
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": self.dtype.str, "data": (ptr, False),
                                         "version": 2, "strides": None}


class CollectedData:
    """CollectedData(obs, logits, values, rewards, actions, perms=None) with get/set attributes
    obs, logits, perms, values, rewards, actions, additional_data and the methods merge,
    get_additional_data_item, set_additional_data_item (python_interface/collector.rs:24-137)."""

    def __init__(self, obs, logits, values, rewards, actions, perms=None):
        self._dev = None
        self._obs = [list(map(int, o)) for o in obs]
        self._logits = [list(map(float, l)) for l in logits]
        self._values = [float(v) for v in values]
        self._rewards = [float(v) for v in rewards]
        self._actions = [int(a) for a in actions]
        # perms.unwrap_or_else(|| vec![None; obs.len()])  (collector.rs:42)
        self._perms = [-1] * len(self._obs) if perms is None else [(-1 if p is None or p < 0 else int(p)) for p in perms]
        self._additional = {}

    # ---- construction from a device result ----------------------------------------------------
    @classmethod
    def _from_device(cls, dev: _DeviceResult) -> "CollectedData":
        self = cls.__new__(cls)
        self._dev = dev
        self._obs = self._logits = self._values = self._rewards = self._actions = self._perms = None
        self._additional = None
        return self

    def _materialize(self):
        """device buffers -> the reference's list-of-lists representation (each PyO3 getter clones
        into fresh Python lists, collector.rs:55-131)."""
        if self._dev is None or self._obs is not None:
            return
        d = self._dev
        self._obs = d.host(_lib.TW_F_OBS).astype(np.int64).tolist()
        self._logits = d.host(_lib.TW_F_LOGITS).tolist()
        self._perms = d.host(_lib.TW_F_PERMS).astype(np.int64).tolist()
        if d.is_ppo:
            self._values = d.host(_lib.TW_F_VALUES).tolist()
            self._rewards = d.host(_lib.TW_F_REWARDS).tolist()
            self._actions = d.host(_lib.TW_F_ACTIONS).astype(np.int64).tolist()
            self._additional = {"advs": d.host(_lib.TW_F_ADVS).tolist(), "rets": d.host(_lib.TW_F_RETS).tolist()}
        else:   # AZ: values / rewards / actions stay empty (collector/az.rs:97-104)
            self._values, self._rewards, self._actions = [], [], []
            self._additional = {"remaining_values": d.host(_lib.TW_F_REMAINING).tolist()}

    def _detach(self):
        self._materialize()
        self._dev = None

    # ---- reference attributes ------------------------------------------------------------------
    def _get(self, name):
        self._materialize()
        return getattr(self, name)

    obs = property(lambda s: [list(o) for o in s._get("_obs")], lambda s, v: s._set("_obs", [list(map(int, o)) for o in v]))
    logits = property(lambda s: [list(l) for l in s._get("_logits")], lambda s, v: s._set("_logits", [list(map(float, l)) for l in v]))
    perms = property(lambda s: list(s._get("_perms")), lambda s, v: s._set("_perms", [(-1 if p < 0 else int(p)) for p in v]))
    values = property(lambda s: list(s._get("_values")), lambda s, v: s._set("_values", [float(x) for x in v]))
    rewards = property(lambda s: list(s._get("_rewards")), lambda s, v: s._set("_rewards", [float(x) for x in v]))
    actions = property(lambda s: list(s._get("_actions")), lambda s, v: s._set("_actions", [int(x) for x in v]))
    additional_data = property(lambda s: {k: list(v) for k, v in s._get("_additional").items()},
                               lambda s, v: s._set("_additional", {str(k): [float(x) for x in vv] for k, vv in v.items()}))

    def _set(self, name, value):
        self._detach()
        setattr(self, name, value)

    def merge(self, other: "CollectedData") -> None:
        """Append `other` (collector/collector.rs:70-88)."""
        self._detach()
        other._materialize()
        self._obs.extend([list(o) for o in other._obs])
        self._logits.extend([list(l) for l in other._logits])
        self._perms.extend(other._perms)
        self._values.extend(other._values)
        self._rewards.extend(other._rewards)
        self._actions.extend(other._actions)
        for k, v in other._additional.items():
            self._additional.setdefault(k, [])
            self._additional[k].extend(v)

    def get_additional_data_item(self, key: str):
        self._materialize()
        v = self._additional.get(key)
        return None if v is None else list(v)

    def set_additional_data_item(self, key: str, value) -> None:
        self._detach()
        self._additional[str(key)] = [float(x) for x in value]

    # ---- build extensions: zero-copy access -----------------------------------------------------
    def __len__(self):
        return self._dev.n if (self._dev is not None and self._obs is None) else len(self._get("_obs"))

    @property
    def on_device(self) -> bool:
        return self._dev is not None

    @property
    def stats(self) -> dict:
        """Per-kernel HIP-event timings and counts of the collect() call that produced this."""
        return dict(self._dev.stats) if self._dev is not None else {}

    def device_arrays(self) -> dict:
        """name -> DeviceArray (zero-copy, __cuda_array_interface__) for every field present."""
        if self._dev is None:
            raise RuntimeError("this CollectedData no longer aliases device buffers (it was modified or built on the host)")
        out = {}
        for name, f in _FIELD_NAMES.items():
            if self._dev.ptr(f)[1] > 0:
                out[name] = DeviceArray(self._dev, f)
        return out

    def to_numpy(self) -> dict:
        if self._dev is None:
            raise RuntimeError("this CollectedData no longer aliases device buffers")
        return {name: self._dev.host(f) for name, f in _FIELD_NAMES.items() if self._dev.ptr(f)[1] > 0}

    def to_torch(self) -> dict:
        """name -> torch tensor on the current GPU aliasing the result buffers (no copy)."""
        import torch
        out = {}
        for name, arr in self.device_arrays().items():
            t = torch.as_tensor(arr, device="cuda")
            t._tw_owner = arr   # keep the owner alive as long as the tensor
            out[name] = t
        return out


class PyBaseCollector:
    """collect(py_env, policy) -> CollectedData (python_interface/collector.rs:139-152)."""

    _IS_PPO = True

    def collect(self, py_env, policy: Policy) -> CollectedData:
        raise NotImplementedError

    def empty_fields(self, py_env) -> dict:
        """Zero-record device tensors in the layout to_torch() gives (a rank without episodes in a sharded collect still
        takes part in the gather's collectives, twisterl_amd.dist)."""
        import torch
        desc = get_env_desc(py_env)
        nc = int(desc.width) * int(desc.height)
        z = lambda shape, dt: torch.empty(shape, dtype=dt, device="cuda")
        out = {"obs": z((0, nc), torch.uint8 if nc * nc <= 256 else torch.int16), "logits": z((0, 4), torch.float32), "perms": z((0,), torch.int8)}
        if self._IS_PPO:
            out.update(values=z((0,), torch.float32), rewards=z((0,), torch.float32), actions=z((0,), torch.uint8),
                       advs=z((0,), torch.float32), rets=z((0,), torch.float32))
        else:
            out["remaining_values"] = z((0,), torch.float32)
        return out

    @staticmethod
    def _check(py_env, policy):
        desc = get_env_desc(py_env)
        if not isinstance(policy, Policy):
            raise TypeError("argument 'policy': expected twisterl_amd.nn.Policy")
        return desc


def _u(name, v):
    if int(v) < 0:
        raise OverflowError(f"{name}: can't convert negative int to unsigned")
    return int(v)


class PPOCollector(PyBaseCollector):
    """PPOCollector(num_episodes, gamma, lambda, num_cores) (collector.rs:154-170).

    `lambda` is a Python keyword, so -- exactly as with the reference -- it is reachable through
    `PPOCollector(**config["collecting"])` (src/twisterl/rl/ppo.py:23).  `num_cores` is accepted
    for compatibility; the episodes run as GPU lanes, not rayon tasks.  Build extensions (all
    keyword-only, defaulted): seed, precision ("fp32" exact | "fp16" | "fp16x2"), episode_offset /
    merge_order / reserve_cus for sharded collection (twisterl_amd.dist).
    """

    def __init__(self, *args, **kwargs):
        names = ["num_episodes", "gamma", "lambda", "num_cores"]
        vals = dict(zip(names, args))
        if len(args) > 4:
            raise TypeError("PPOCollector() takes 4 positional arguments")
        for k in list(kwargs):
            if k in names:
                if k in vals:
                    raise TypeError(f"PPOCollector() got multiple values for argument '{k}'")
                vals[k] = kwargs.pop(k)
        missing = [n for n in names if n not in vals]
        if missing:
            raise TypeError(f"PPOCollector() missing required argument: '{missing[0]}'")
        self.num_episodes = _u("num_episodes", vals["num_episodes"])
        self.gamma = float(vals["gamma"])
        self.lambda_ = float(vals["lambda"])
        self.num_cores = _u("num_cores", vals["num_cores"])
        self.seed = kwargs.pop("seed", None)
        self.precision = kwargs.pop("precision", "fp32")
        self.episode_offset = int(kwargs.pop("episode_offset", 0))
        self.merge_order = bool(kwargs.pop("merge_order", True))
        self.reserve_cus = int(kwargs.pop("reserve_cus", 0))
        if kwargs:
            raise TypeError(f"PPOCollector() got an unexpected keyword argument '{next(iter(kwargs))}'")
        if self.precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        self._calls = 0

    def _next_seed(self) -> int:
        # The reference draws fresh OS entropy per call (rand::thread_rng()).  With seed=None we do
        # the same; with a fixed seed every call advances a counter so successive learn_steps see
        # different episodes while the whole run stays reproducible.
        if self.seed is None:
            import os
            return int.from_bytes(os.urandom(8), "little")
        s = (int(self.seed) + 0x9E3779B97F4A7C15 * self._calls) & (2**64 - 1)
        self._calls += 1
        return s

    def collect(self, py_env, policy: Policy, *, seed=None) -> CollectedData:
        if isinstance(py_env, PyEnv):
            return self._collect_foreign(py_env, policy, seed)
        desc = self._check(py_env, policy)
        prm = _lib.PPOParams(self.num_episodes, self.episode_offset, self.gamma, self.lambda_,
                             (int(seed) & (2**64 - 1)) if seed is not None else self._next_seed(),
                             _lib.PRECISIONS[self.precision], int(self.merge_order), max(0, int(self.reserve_cus)))
        out = C.c_void_p()
        _lib.check(_lib.lib().tw_ppo_collect(C.byref(desc), policy._handle(), C.byref(prm), C.byref(out)))
        return CollectedData._from_device(_DeviceResult(out.value))


class _PyEnvBridge:
    """A Python environment (reference protocol, python_interface/pyenv.rs) as the C function table of `trait Env`
    (tw_env_vtable): clones live here, keyed by small integer handles; an exception raised by the environment's code is kept
    and re-raised by the caller once the library call has returned (it must not unwind through the C frames)."""

    def __init__(self, py_env: PyEnv, prototype=None):
        proto = py_env._env if prototype is None else prototype
        self.proto, self.err = proto, []
        envs, nxt, err = {1: proto}, [2], self.err
        V = _lib.EnvVTable

        def guard(default=None):
            def deco(fn):
                def wrapped(*a):
                    if err:
                        return default
                    try:
                        return fn(*a)
                    except BaseException as e:
                        err.append(e)
                        return default
                return wrapped
            return deco

        @guard(0)
        def f_clone(h):
            envs[nxt[0]] = envs[h].copy()
            nxt[0] += 1
            return nxt[0] - 1

        @guard()
        def f_destroy(h):
            envs.pop(h, None)

        @guard()
        def f_reset(h, sd, ep):
            e = envs[h]
            if hasattr(e, "seed_episode"):
                e.seed_episode(int(sd), int(ep))
            e.reset(py_env.difficulty)                  # PyEnvImpl::reset (pyenv.rs:93-100)

        @guard()
        def f_step(h, a):
            envs[h].next(int(a))

        # The C side hands out buffers of exactly n_obs ids / num_actions flags per state (tw_env_vtable): an environment whose
        # observe() changes length (legal for the reference's EmbeddingBag, layers.rs:56-62) is refused here instead of writing
        # past the buffer or leaving stale ids in it -- see PyEnv's docstring.
        @guard()
        def f_observe(h, out_p):
            obs = envs[h].observe()
            if len(obs) != n_obs:
                raise ValueError(f"PyEnv: observe() returned {len(obs)} ids, the prototype returned {n_obs}; this collector needs a "
                                 "fixed number of obs ids per state")
            for i, v in enumerate(obs):
                v = int(v)
                if not 0 <= v < obs_size:
                    raise IndexError(f"PyEnv: obs id {v} outside the obs_shape's {obs_size} ids")      # (the reference panics: layers.rs:61)
                out_p[i] = v

        @guard()
        def f_masks(h, out_p):
            masks = envs[h].masks()
            if len(masks) != n_actions:
                raise ValueError(f"PyEnv: masks() returned {len(masks)} flags for {n_actions} actions")
            for i, v in enumerate(masks):
                out_p[i] = 1 if v else 0

        @guard(0.0)
        def f_reward(h):
            return float(envs[h].value())

        @guard(1)
        def f_final(h):
            return 1 if envs[h].is_final() else 0

        @guard(0)
        def f_success(h):
            return 1 if envs[h].success() else 0         # PyEnvImpl::success (pyenv.rs:141-149)

        # The rest of `trait Env` (rust/src/rl/env.rs:30,58-66).  The reference's PyEnvImpl leaves track_solution / solution /
        # twists at the trait's defaults (python_interface/pyenv.rs implements none of them), so these slots stay NULL unless the
        # Python object defines the method -- a build extension that gives Python environments what Rust ones have.
        @guard(0)
        def f_track(h):
            return 1 if envs[h].track_solution() else 0

        @guard(0)
        def f_solution(h, out_p, cap):
            sol = [int(x) for x in envs[h].solution()]
            for i in range(min(len(sol), int(cap))):
                out_p[i] = sol[i]
            return len(sol)

        @guard()
        def f_set_state(h, st_p, n):
            envs[h].set_state([int(st_p[i]) for i in range(int(n))])

        n_obs, n_actions = len(proto.observe()), int(proto.num_actions())
        obs_size = 1
        for x in proto.obs_shape():
            obs_size *= int(x)

        @guard(0)
        def f_twists(h, obs_p, act_p, cap):
            op, ap = envs[h].twists()
            for k in range(min(len(op), int(cap))):
                for i, v in enumerate(op[k]):
                    obs_p[k * obs_size + i] = int(v)
                for i, v in enumerate(ap[k]):
                    act_p[k * n_actions + i] = int(v)
            return len(op)

        fields = dict(V._fields_)
        opt = lambda name, fn: fields[name](fn) if callable(getattr(proto, name, None)) else fields[name]()
        self._keep = (f_clone, f_destroy, f_reset, f_step, f_observe, f_masks, f_reward, f_final, f_success, f_track, f_solution, f_set_state, f_twists)
        self.vt = V(1, n_actions, n_obs, obs_size, fields["clone"](f_clone), fields["destroy"](f_destroy), fields["reset"](f_reset),
                    fields["step"](f_step), fields["observe"](f_observe), fields["masks"](f_masks), fields["reward"](f_reward),
                    fields["is_final"](f_final), fields["success"](f_success),
                    opt("track_solution", f_track), opt("solution", f_solution), opt("set_state", f_set_state), opt("twists", f_twists))
        self.max_records = int(getattr(proto, "max_records", 1 << 16))

    def finish(self, rc):
        if self.err:
            raise self.err[0]
        _lib.check(rc)


def _collect_foreign(self, py_env: PyEnv, policy: Policy, seed) -> CollectedData:
    """PPOCollector / AZCollector.collect for an environment implemented in Python (tw_ppo_collect_env / tw_az_collect_env)."""
    if not isinstance(policy, Policy):
        raise TypeError("argument 'policy': expected twisterl_amd.nn.Policy")
    if self.precision not in ("fp32", "f32", "exact"):
        raise RuntimeError("environments implemented in Python are collected in f32")
    br = _PyEnvBridge(py_env)
    sd = (int(seed) & (2**64 - 1)) if seed is not None else self._next_seed()
    out = C.c_void_p()
    if self._IS_PPO:
        prm = _lib.PPOParams(self.num_episodes, self.episode_offset, self.gamma, self.lambda_, sd, _lib.TW_PREC_F32_EXACT, int(self.merge_order), 0)
        rc = _lib.lib().tw_ppo_collect_env(C.byref(br.vt), policy._handle(), C.byref(prm), br.max_records, C.byref(out))
    else:
        prm = _lib.AZParams(self.num_episodes, self.episode_offset, self.num_mcts_searches, self.C, self.max_expand_depth, sd,
                            _lib.TW_PREC_F32_EXACT, int(self.merge_order), 0)
        rc = _lib.lib().tw_az_collect_env(C.byref(br.vt), policy._handle(), C.byref(prm), br.max_records, C.byref(out))
    if br.err and out.value:
        _lib.lib().tw_collected_free(out)
    br.finish(rc)
    return CollectedData._from_device(_DeviceResult(out.value))


PPOCollector._collect_foreign = _collect_foreign


class AZCollector(PyBaseCollector):
    """AZCollector(num_episodes, num_mcts_searches, C, max_expand_depth, num_cores) (collector.rs:172-187)."""

    _IS_PPO = False

    def __init__(self, num_episodes, num_mcts_searches, C, max_expand_depth, num_cores, *, seed=None,
                 precision="fp32", episode_offset=0, merge_order=True, reserve_cus=0):
        self.num_episodes = _u("num_episodes", num_episodes)
        self.num_mcts_searches = _u("num_mcts_searches", num_mcts_searches)
        self.C = float(C)
        self.max_expand_depth = _u("max_expand_depth", max_expand_depth)
        self.num_cores = _u("num_cores", num_cores)
        self.seed, self.precision = seed, precision
        self.episode_offset, self.merge_order = int(episode_offset), bool(merge_order)
        self.reserve_cus = int(reserve_cus)
        if self.precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        self._calls = 0

    _next_seed = PPOCollector._next_seed
    _collect_foreign = _collect_foreign

    def collect(self, py_env, policy: Policy, *, seed=None) -> CollectedData:
        if isinstance(py_env, PyEnv):
            return self._collect_foreign(py_env, policy, seed)
        desc = self._check(py_env, policy)
        prm = _lib.AZParams(self.num_episodes, self.episode_offset, self.num_mcts_searches, self.C,
                            self.max_expand_depth,
                            (int(seed) & (2**64 - 1)) if seed is not None else self._next_seed(),
                            _lib.PRECISIONS[self.precision], int(self.merge_order), max(0, int(self.reserve_cus)))
        out = C.c_void_p()
        _lib.check(_lib.lib().tw_az_collect(C.byref(desc), policy._handle(), C.byref(prm), C.byref(out)))
        return CollectedData._from_device(_DeviceResult(out.value))


def _solve_params(deterministic, num_searches, num_mcts_searches, C_, max_expand_depth, seed, precision="fp32"):
    import os
    if seed is None:
        seed = int.from_bytes(os.urandom(8), "little")
    return _lib.SolveParams(int(bool(deterministic)), _u("num_searches", num_searches),
                            _u("num_mcts_searches", num_mcts_searches), float(C_), _u("max_expand_depth", max_expand_depth),
                            int(seed) & (2**64 - 1), _lib.PRECISIONS[precision])


def solve(py_env, policy: Policy, deterministic, num_searches, num_mcts_searches, C, max_expand_depth, *, seed=None):
    """solve(py_env, policy, deterministic, num_searches, num_mcts_searches, C, max_expand_depth)
    -> ((success, reward), actions)   (python_interface/env.rs:180-191 over rl/solve.rs:73-101).
    Best of `num_searches` greedy/sampled roll-outs from the env's CURRENT state; the env is not modified."""
    if not isinstance(policy, Policy):
        raise TypeError("argument 'policy': expected twisterl_amd.nn.Policy")
    prm = _solve_params(deterministic, num_searches, num_mcts_searches, C, max_expand_depth, seed)
    if isinstance(py_env, PyEnv):
        br = _PyEnvBridge(py_env, prototype=py_env._env.copy())             # (the caller's object is not touched)
        cap = br.max_records + 1
        # 32-bit entries: an environment that tracks its own solution returns THAT (solve.rs:57-64), and its entries need not be actions
        cap = max(cap, int(getattr(py_env._env, "max_solution", 0)))
        acts = (_c.c_uint32 * cap)()
        s, r, n = _c.c_float(), _c.c_float(), _c.c_uint32()
        br.finish(_lib.lib().tw_solve_env32(_c.byref(br.vt), policy._handle(), _c.byref(prm), br.max_records, _c.byref(s), _c.byref(r), acts, cap, _c.byref(n)))
        return (float(s.value), float(r.value)), [int(acts[i]) for i in range(n.value)]
    get_env_desc(py_env)
    cap = int(py_env.depth) + 2
    acts = (ctypes_u8 * cap)()
    s, r, n = _c.c_float(), _c.c_float(), _c.c_uint32()
    _lib.check(_lib.lib().tw_solve(py_env._h, policy._handle(), _c.byref(prm), _c.byref(s), _c.byref(r), acts, cap, _c.byref(n)))
    return (float(s.value), float(r.value)), [int(acts[i]) for i in range(n.value)]


def evaluate(py_env, policy: Policy, num_episodes, deterministic, num_searches, num_mcts_searches, seed, C, max_expand_depth,
             num_cores):
    """evaluate(py_env, policy, num_episodes, deterministic, num_searches, num_mcts_searches, seed, C,
    max_expand_depth, num_cores) -> (success_rate, mean_reward)   (python_interface/env.rs:194-207 over
    rl/evaluate.rs:22-89).  `seed` keys the episode scrambles and action draws (the reference ignores it);
    `num_cores` is accepted for compatibility."""
    if not isinstance(policy, Policy):
        raise TypeError("argument 'policy': expected twisterl_amd.nn.Policy")
    _u("num_cores", num_cores)
    prm = _solve_params(deterministic, num_searches, num_mcts_searches, C, max_expand_depth, seed)
    s, r = _c.c_float(), _c.c_float()
    if isinstance(py_env, PyEnv):
        br = _PyEnvBridge(py_env)
        br.finish(_lib.lib().tw_evaluate_env(_c.byref(br.vt), policy._handle(), _c.byref(prm), _u("num_episodes", num_episodes), 0, br.max_records,
                                             _c.byref(s), _c.byref(r)))
        return float(s.value), float(r.value)
    desc = get_env_desc(py_env)
    _lib.check(_lib.lib().tw_evaluate(_c.byref(desc), policy._handle(), _c.byref(prm), _u("num_episodes", num_episodes), 0,
                                      _c.byref(s), _c.byref(r)))
    return float(s.value), float(r.value)
