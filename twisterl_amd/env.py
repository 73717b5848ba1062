"""`twisterl.env` surface: Puzzle with the PyBaseEnv methods.

Mirrors reference rust/src/python_interface/env.rs:39-160 (PyBaseEnv + Puzzle) over
rust/src/envs/puzzle.rs.  The object is a host-side handle owned by libtwisterl_hip.so
(tw_puzzle_*); collectors read only its descriptor -- like the reference, which clones and
resets the env per episode and never mutates the one passed in (collector/ppo.rs:59-60).
"""
from __future__ import annotations

import ctypes as C

from . import _lib


class PyBaseEnv:
    """Base class of envs the HIP collectors accept (python_interface/env.rs:39-114)."""

    _h = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().tw_puzzle_destroy(h)
            except Exception:
                pass

    # -- PyBaseEnv pymethods (env.rs:44-114) ---------------------------------------------------
    def num_actions(self) -> int:
        return int(_lib.lib().tw_puzzle_num_actions(self._h))

    def obs_shape(self) -> list:
        out = (C.c_uint32 * 2)()
        _lib.check(_lib.lib().tw_puzzle_obs_shape(self._h, out))
        return [int(out[0]), int(out[1])]

    @property
    def difficulty(self) -> int:
        return int(_lib.lib().tw_puzzle_get_difficulty(self._h))

    @difficulty.setter
    def difficulty(self, value: int) -> None:
        if int(value) < 0:
            raise OverflowError("can't convert negative int to unsigned")
        _lib.check(_lib.lib().tw_puzzle_set_difficulty(self._h, int(value)))

    def set_state(self, state) -> None:
        arr = (C.c_int64 * len(state))(*[int(s) for s in state])
        _lib.check(_lib.lib().tw_puzzle_set_state(self._h, arr, len(state)))

    def reset(self, seed: int = None, episode: int = 0) -> None:
        """Scramble by `difficulty` uniform moves (puzzle.rs:119-133).  The reference draws from
        the unseedable thread_rng; here the draw is stream 0 of the library's counter-based RNG,
        keyed by (seed, episode).  seed=None picks a fresh OS seed, like the reference."""
        if seed is None:
            import os
            seed = int.from_bytes(os.urandom(8), "little")
        _lib.check(_lib.lib().tw_puzzle_reset(self._h, int(seed) & (2**64 - 1), int(episode)))

    def step(self, action: int) -> None:
        _lib.check(_lib.lib().tw_puzzle_step(self._h, int(action)))

    def masks(self) -> list:
        out = (C.c_uint8 * 4)()
        _lib.check(_lib.lib().tw_puzzle_masks(self._h, out))
        return [bool(x) for x in out]

    def is_final(self) -> bool:
        return bool(_lib.lib().tw_puzzle_is_final(self._h))

    def reward(self) -> float:
        return float(_lib.lib().tw_puzzle_reward(self._h))

    def observe(self) -> list:
        n = self.obs_shape()[0]
        out = (C.c_int64 * n)()
        _lib.check(_lib.lib().tw_puzzle_observe(self._h, out))
        return [int(x) for x in out]

    def twists(self):
        """Env::twists default (rl/env.rs:59): Puzzle does not override it."""
        return ([], [])

    def __extract_env__(self) -> int:
        """Address of the native env object (env.rs:109-113).  Here it is a `tw_puzzle*` of
        libtwisterl_hip.so, not a Rust Box<dyn Env>."""
        return int(self._h)

    # -- used by the collectors ----------------------------------------------------------------
    def _desc(self) -> "_lib.PuzzleDesc":
        d = _lib.PuzzleDesc()
        _lib.check(_lib.lib().tw_puzzle_get_desc(self._h, C.byref(d)))
        return d


class Puzzle(PyBaseEnv):
    """Puzzle(width, height, difficulty, depth_slope, max_depth) (env.rs:117-160)."""

    def __init__(self, width: int, height: int, difficulty: int, depth_slope: int, max_depth: int):
        for v in (width, height, difficulty, depth_slope, max_depth):
            if int(v) < 0:
                raise OverflowError("can't convert negative int to unsigned")
        self._h = _lib.lib().tw_puzzle_create(int(width), int(height), int(difficulty), int(depth_slope),
                                              int(max_depth))
        if not self._h:
            raise ValueError(_lib.last_error())

    def solved(self) -> bool:
        return bool(_lib.lib().tw_puzzle_solved(self._h))

    def get_state(self) -> list:
        n = self.obs_shape()[0]
        out = (C.c_int64 * n)()
        _lib.check(_lib.lib().tw_puzzle_get_state(self._h, out))
        return [int(x) for x in out]

    def display(self) -> None:
        """puzzle.rs:56-69"""
        w = int(self._desc().width)
        line = ""
        for i, v in enumerate(self.get_state()):
            line += "   " if v == 0 else (f"  {v} " if v < 10 else f" {v} ")
            if (i + 1) % w == 0:
                print(line)
                line = ""

    def set_position(self, x: int, y: int, val: int) -> None:
        _lib.check(_lib.lib().tw_puzzle_set_position(self._h, int(x), int(y), int(val)))

    def get_position(self, x: int, y: int) -> int:
        return int(_lib.lib().tw_puzzle_get_position(self._h, int(x), int(y)))

    @property
    def depth(self) -> int:
        return int(_lib.lib().tw_puzzle_depth(self._h))


class PyEnv:
    """PyEnv(pyenv): an environment implemented in Python (reference rust/src/python_interface/pyenv.rs:41-160,
    src/twisterl/envs/__init__.py:20-25).  `pyenv` provides the methods the reference calls on it: copy(), num_actions(),
    obs_shape(), reset(difficulty), next(action), masks(), is_final(), value(), success(), observe(), set_state(state).
    Its code runs on the host -- as in the reference -- while the policy forward of all live episodes of a time step is one
    batched launch on the GPU (tw_ppo_collect_env).  Build extension: if `pyenv` has seed_episode(seed, episode) it is
    called before every reset(), so that a collect is reproducible (the reference's envs draw from OS entropy).
    Restriction: observe() must return the SAME NUMBER of ids for every state (the prototype's) and masks() one flag per
    action -- the C side's per-state buffers have that size (tw_env_vtable.n_obs); the reference's EmbeddingBag would also
    take observations of varying length (layers.rs:56-62).  A different length, or an id outside obs_shape, raises from
    collect() / evaluate() / solve()."""

    def __init__(self, pyenv):
        for m in ("copy", "num_actions", "obs_shape", "reset", "next", "masks", "is_final", "value", "observe"):
            if not callable(getattr(pyenv, m, None)):
                raise TypeError(f"PyEnv: the environment object has no method {m}()")
        self._env = pyenv
        self._difficulty = 1                                   # PyEnvImpl::new (pyenv.rs:41-45)

    def __extract_env__(self) -> int:
        return id(self)

    difficulty = property(lambda self: self._difficulty, lambda self, d: setattr(self, "_difficulty", int(d)))

    def num_actions(self) -> int:
        return int(self._env.num_actions())

    def obs_shape(self) -> list:
        return [int(x) for x in self._env.obs_shape()]

    def set_state(self, state) -> None:
        self._env.set_state([int(x) for x in state])

    def reset(self) -> None:
        self._env.reset(self._difficulty)

    def step(self, action: int) -> None:
        self._env.next(int(action))

    def masks(self) -> list:
        return [bool(m) for m in self._env.masks()]

    def is_final(self) -> bool:
        return bool(self._env.is_final())

    def reward(self) -> float:
        return float(self._env.value())

    def observe(self) -> list:
        return [int(x) for x in self._env.observe()]

    def twists(self):
        """Env::twists (rl/env.rs:58-59).  The reference's PyEnvImpl keeps the trait's default (no twists); a Python environment
        that defines twists() is honoured here (build extension) -- the host builds its Policy with them."""
        if callable(getattr(self._env, "twists", None)):
            op, ap = self._env.twists()
            return [[int(x) for x in p] for p in op], [[int(x) for x in p] for p in ap]
        return ([], [])

    def track_solution(self) -> bool:
        """Env::track_solution (rl/env.rs:62): false unless the Python environment says otherwise (build extension)."""
        return bool(self._env.track_solution()) if callable(getattr(self._env, "track_solution", None)) else False

    def solution(self) -> list:
        """Env::solution (rl/env.rs:65)."""
        return [int(x) for x in self._env.solution()] if callable(getattr(self._env, "solution", None)) else []


def get_env_desc(py_env) -> "_lib.PuzzleDesc":
    """Counterpart of get_env() (env.rs:163-177).  The reference turns the integer returned by
    `__extract_env__` back into a Rust Box<dyn Env>; this library can only run envs whose
    dynamics it implements on the GPU, so the object must be one of ours."""
    if not hasattr(py_env, "__extract_env__"):
        raise TypeError("Object must implement __extract_env__ method")
    if isinstance(py_env, PyEnv):
        raise TypeError("environments implemented in Python are collected by PPOCollector (tw_ppo_collect_env); self-play, "
                        "evaluate and solve run environments whose dynamics the library implements on the GPU (Puzzle)")
    if not isinstance(py_env, PyBaseEnv):
        raise TypeError("Expected environment of type twisterl_amd.env.Puzzle "
                        "(the HIP collectors cannot run a foreign Box<dyn Env>)")
    return py_env._desc()
