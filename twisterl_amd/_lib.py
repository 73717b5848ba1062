"""ctypes binding of the C ABI in include/twisterl_hip.h (libtwisterl_hip.so).

This is the only place the shared library is loaded.  There is no fallback: if the library is
missing it is built with hipcc; if that fails, or no GPU is present when a compute entry point
is called, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

# status codes (include/twisterl_hip.h)
TW_OK, TW_ERR_INVALID, TW_ERR_UNSUPPORTED, TW_ERR_NO_DEVICE, TW_ERR_HIP, TW_ERR_EMPTY = range(6)
TW_PREC_F32_EXACT, TW_PREC_F16, TW_PREC_F16X2 = 0, 1, 2
TW_EVAL_FORWARD, TW_EVAL_PREDICT, TW_EVAL_FULL_PREDICT = 0, 1, 2
TW_OPT_FORCE_GEOM, TW_OPT_NO_PERSIST, TW_OPT_AZ_VARIANT, TW_OPT_AZ_TREE_BUDGET, TW_OPT_AZ_TREE_BUDGET_MIN, TW_OPT_AZ_REUSE = 0, 1, 2, 3, 4, 5
ABI_VERSION = 6
(TW_F_OBS, TW_F_LOGITS, TW_F_PERMS, TW_F_VALUES, TW_F_REWARDS, TW_F_ACTIONS, TW_F_ADVS, TW_F_RETS,
 TW_F_REMAINING, TW_F_EP_LEN, TW_F_EP_START, TW_F_COUNT) = range(12)

PRECISIONS = {"fp32": TW_PREC_F32_EXACT, "f32": TW_PREC_F32_EXACT, "exact": TW_PREC_F32_EXACT,
              "fp16": TW_PREC_F16, "f16": TW_PREC_F16, "fp16x2": TW_PREC_F16X2, "f16x2": TW_PREC_F16X2}


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64), ("compute_units", C.c_int32),
                ("wavefront_size", C.c_int32), ("total_mem_bytes", C.c_uint64),
                ("lds_bytes_per_block", C.c_uint64)]


class PuzzleDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("difficulty", C.c_uint32),
                ("depth_slope", C.c_uint32), ("max_depth", C.c_uint32)]


class LinearDesc(C.Structure):
    _fields_ = [("in_features", C.c_uint32), ("out_features", C.c_uint32),
                ("weights", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float)),
                ("apply_relu", C.c_uint32)]


class PolicyDesc(C.Structure):
    _fields_ = [("obs_size", C.c_uint32), ("emb_size", C.c_uint32),
                ("emb_vectors", C.POINTER(C.c_float)), ("emb_bias", C.POINTER(C.c_float)),
                ("emb_apply_relu", C.c_uint32),
                ("n_common", C.c_uint32), ("common", C.POINTER(LinearDesc)),
                ("n_action", C.c_uint32), ("action", C.POINTER(LinearDesc)),
                ("n_value", C.c_uint32), ("value", C.POINTER(LinearDesc)),
                ("n_perms", C.c_uint32), ("n_actions", C.c_uint32),
                ("obs_perms", C.POINTER(C.c_int32)), ("act_perms", C.POINTER(C.c_int32))]


class PPOParams(C.Structure):
    _fields_ = [("num_episodes", C.c_uint64), ("episode_offset", C.c_uint64),
                ("gamma", C.c_float), ("lambda_", C.c_float), ("seed", C.c_uint64),
                ("precision", C.c_uint32), ("merge_order", C.c_uint32), ("reserve_cus", C.c_uint32)]


class AZParams(C.Structure):
    _fields_ = [("num_episodes", C.c_uint64), ("episode_offset", C.c_uint64),
                ("num_mcts_searches", C.c_uint32), ("C", C.c_float), ("max_expand_depth", C.c_uint32),
                ("seed", C.c_uint64), ("precision", C.c_uint32), ("merge_order", C.c_uint32),
                ("reserve_cus", C.c_uint32)]


class SolveParams(C.Structure):
    _fields_ = [("deterministic", C.c_uint32), ("num_searches", C.c_uint32), ("num_mcts_searches", C.c_uint32),
                ("C", C.c_float), ("max_expand_depth", C.c_uint32), ("seed", C.c_uint64), ("precision", C.c_uint32)]


class CollectStats(C.Structure):
    _fields_ = [("ms_rollout", C.c_float), ("ms_scan", C.c_float), ("ms_finalize", C.c_float),
                ("ms_total", C.c_float), ("records", C.c_uint64), ("episodes", C.c_uint64),
                ("padded_bytes", C.c_uint64), ("rollout_blocks", C.c_uint32),
                ("rollout_threads", C.c_uint32), ("forward_evals", C.c_uint64), ("speculative_evals", C.c_uint64), ("reused_evals", C.c_uint64)]


class EnvVTable(C.Structure):
    _fields_ = [("prototype", C.c_void_p), ("num_actions", C.c_uint32), ("n_obs", C.c_uint32), ("obs_size", C.c_uint32),
                ("clone", C.CFUNCTYPE(C.c_void_p, C.c_void_p)), ("destroy", C.CFUNCTYPE(None, C.c_void_p)),
                ("reset", C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64)), ("step", C.CFUNCTYPE(None, C.c_void_p, C.c_uint32)),
                ("observe", C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int32))), ("masks", C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8))),
                ("reward", C.CFUNCTYPE(C.c_float, C.c_void_p)), ("is_final", C.CFUNCTYPE(C.c_int, C.c_void_p)),
                ("success", C.CFUNCTYPE(C.c_int, C.c_void_p)),
                # the rest of `trait Env` (rust/src/rl/env.rs:30,58-66); NULL = the trait's default body
                ("track_solution", C.CFUNCTYPE(C.c_int, C.c_void_p)),
                ("solution", C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32)),
                ("set_state", C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int64), C.c_uint32)),
                ("twists", C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_uint32))]


class CommId(C.Structure):
    _fields_ = [("bytes", C.c_ubyte * 128)]      # (c_ubyte: a c_char array field would read back truncated at the first NUL)


# every symbol include/twisterl_hip.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "tw_abi_version": (C.c_int, []),
    "tw_last_error": (C.c_char_p, []),
    "tw_device_count": (C.c_int, []),
    "tw_set_device": (C.c_int, [C.c_int]),
    "tw_set_stream": (C.c_int, [_VP]),
    "tw_get_device_info": (C.c_int, [C.POINTER(DeviceInfo)]),
    "tw_release_cached_memory": (C.c_int, []),
    "tw_set_launch_option": (C.c_int, [C.c_int, C.c_int]),
    "tw_debug_counters": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "tw_debug_episode_order": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "tw_puzzle_create": (_VP, [C.c_uint32] * 5),
    "tw_puzzle_clone": (_VP, [_VP]),
    "tw_puzzle_destroy": (None, [_VP]),
    "tw_puzzle_get_desc": (C.c_int, [_VP, C.POINTER(PuzzleDesc)]),
    "tw_puzzle_num_actions": (C.c_uint32, [_VP]),
    "tw_puzzle_obs_shape": (C.c_int, [_VP, C.POINTER(C.c_uint32)]),
    "tw_puzzle_set_difficulty": (C.c_int, [_VP, C.c_uint32]),
    "tw_puzzle_get_difficulty": (C.c_uint32, [_VP]),
    "tw_puzzle_set_state": (C.c_int, [_VP, C.POINTER(C.c_int64), C.c_size_t]),
    "tw_puzzle_reset": (C.c_int, [_VP, C.c_uint64, C.c_uint64]),
    "tw_puzzle_step": (C.c_int, [_VP, C.c_uint32]),
    "tw_puzzle_masks": (C.c_int, [_VP, C.POINTER(C.c_uint8)]),
    "tw_puzzle_is_final": (C.c_int, [_VP]),
    "tw_puzzle_reward": (C.c_float, [_VP]),
    "tw_puzzle_observe": (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    "tw_puzzle_solved": (C.c_int, [_VP]),
    "tw_puzzle_get_state": (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    "tw_puzzle_set_position": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_int64]),
    "tw_puzzle_get_position": (C.c_int64, [_VP, C.c_uint32, C.c_uint32]),
    "tw_puzzle_depth": (C.c_uint32, [_VP]),
    "tw_policy_create": (_VP, [C.POINTER(PolicyDesc)]),
    "tw_policy_destroy": (None, [_VP]),
    "tw_policy_update_device": (C.c_int, [_VP] * 9),
    "tw_policy_update_device_layers": (C.c_int, [_VP, _VP, _VP, C.POINTER(_VP), C.POINTER(_VP), C.c_uint32]),
    "tw_policy_num_actions": (C.c_uint32, [_VP]),
    "tw_policy_num_perms": (C.c_uint32, [_VP]),
    "tw_policy_evaluate": (C.c_int, [_VP, C.c_int, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32, C.c_uint32,
                                     C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                     C.POINTER(C.c_float)]),
    "tw_ppo_collect": (C.c_int, [C.POINTER(PuzzleDesc), _VP, C.POINTER(PPOParams), C.POINTER(_VP)]),
    "tw_az_collect": (C.c_int, [C.POINTER(PuzzleDesc), _VP, C.POINTER(AZParams), C.POINTER(_VP)]),
    "tw_ppo_collect_env": (C.c_int, [C.POINTER(EnvVTable), _VP, C.POINTER(PPOParams), C.c_uint32, C.POINTER(_VP)]),
    "tw_az_collect_env": (C.c_int, [C.POINTER(EnvVTable), _VP, C.POINTER(AZParams), C.c_uint32, C.POINTER(_VP)]),
    "tw_evaluate_env": (C.c_int, [C.POINTER(EnvVTable), _VP, C.POINTER(SolveParams), C.c_uint64, C.c_uint64, C.c_uint32,
                                  C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tw_solve_env": (C.c_int, [C.POINTER(EnvVTable), _VP, C.POINTER(SolveParams), C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                               C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_uint32)]),
    "tw_solve_env32": (C.c_int, [C.POINTER(EnvVTable), _VP, C.POINTER(SolveParams), C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                 C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]),
    "tw_evaluate": (C.c_int, [C.POINTER(PuzzleDesc), _VP, C.POINTER(SolveParams), C.c_uint64, C.c_uint64,
                              C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tw_solve": (C.c_int, [_VP, _VP, C.POINTER(SolveParams), C.POINTER(C.c_float), C.POINTER(C.c_float),
                           C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_uint32)]),
    "tw_collected_num_records": (C.c_uint64, [_VP]),
    "tw_collected_num_episodes": (C.c_uint64, [_VP]),
    "tw_collected_num_cells": (C.c_uint32, [_VP]),
    "tw_collected_num_actions": (C.c_uint32, [_VP]),
    "tw_collected_is_ppo": (C.c_int, [_VP]),
    "tw_collected_obs_width": (C.c_uint32, [_VP]),
    "tw_collected_device_ptr": (_VP, [_VP, C.c_int, C.POINTER(C.c_size_t)]),
    "tw_collected_copy_to_host": (C.c_int, [_VP, C.c_int, _VP, C.c_size_t]),
    "tw_collected_stats": (C.c_int, [_VP, C.POINTER(CollectStats)]),
    "tw_collected_free": (None, [_VP]),
    "tw_collected_pack_trainer": (C.c_int, [_VP, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64, _VP, _VP, _VP, _VP, _VP]),
    "tw_collected_adv_stats": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tw_comm_get_unique_id": (C.c_int, [C.POINTER(CommId)]),
    "tw_comm_init": (C.c_int, [C.c_int, C.c_int, C.POINTER(CommId), C.POINTER(_VP)]),
    "tw_comm_destroy": (None, [_VP]),
    "tw_comm_rank": (C.c_int, [_VP]),
    "tw_comm_world": (C.c_int, [_VP]),
    "tw_comm_broadcast_policy": (C.c_int, [_VP, _VP, C.c_int]),
    "tw_comm_set_timeout_ms": (C.c_int, [_VP, C.c_uint32]),
    "tw_gather_plan": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), _VP]),
    "tw_gather_begin": (C.c_int, [_VP, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(_VP)]),
    "tw_gather_submit": (C.c_int, [_VP, _VP, C.c_uint64]),
    "tw_gather_finish": (C.c_int, [_VP, C.POINTER(_VP)]),
}

_lib = None


def library_path() -> str:
    return _build.LIB_PATH


def lib():
    """Load (building first if needed) libtwisterl_hip.so and declare every prototype."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm bundles its own HIP runtime under the same soname as /opt/rocm's.  Whichever is loaded first
        # serves the whole process, and torch fails to initialise ("No HIP GPUs are available") when it finds the
        # other one already loaded: load torch's first, this library then binds to it as well.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        path = _build.LIB_PATH
        if not os.path.exists(path) or os.environ.get("TWISTERL_AMD_REBUILD"):
            path = _build.build_library()
        try:
            L = C.CDLL(path)
        except OSError as e:  # loud: no CPU fallback exists
            raise RuntimeError(f"cannot load the HIP collector library {path}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.tw_abi_version() != ABI_VERSION:
            raise RuntimeError("libtwisterl_hip.so ABI version mismatch; rebuild with python -m twisterl_amd.build")
        _lib = L
    return _lib


def last_error() -> str:
    msg = lib().tw_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int) -> None:
    """Status-code -> exception, following the reference's convention: anyhow::Error ->
    PyRuntimeError (python_interface/error_mapping.rs:20-33); bad arguments -> ValueError."""
    if status == TW_OK:
        return
    msg = last_error()
    if status == TW_ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


class launch_option:
    """`with launch_option(TW_OPT_NO_PERSIST, 1): ...` -- pins a diagnostic launch override (tw_set_launch_option) for
    the block; the parity tests use it to show that every launch shape gives the same bytes."""

    def __init__(self, option: int, value: int):
        self.option, self.value = option, value

    def __enter__(self):
        check(lib().tw_set_launch_option(self.option, self.value))
        return self

    def __exit__(self, *exc):
        check(lib().tw_set_launch_option(self.option, 0))
        return False


def debug_counters(n: int = 16) -> list:
    """Diagnostic counters of this process's last self-play launch (tw_debug_counters)."""
    buf = (C.c_uint64 * n)()
    check(lib().tw_debug_counters(buf, n))
    return [int(x) for x in buf]


def debug_episode_order(desc, seed: int, episode_offset: int, n: int):
    """Start boards (uint64, nibble i = tile at cell i) of n Puzzle episodes and the order in which the self-play walkers take them
    (tw_debug_episode_order; test hook).  `desc`: a PuzzleDesc."""
    import numpy as np
    boards = np.zeros(n, dtype=np.uint64); order = np.zeros(n, dtype=np.uint32)
    check(lib().tw_debug_episode_order(C.byref(desc), seed, episode_offset, n, boards.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       order.ctypes.data_as(C.POINTER(C.c_uint32))))
    return boards, order


def release_cached_memory() -> None:
    """Gives the library's cached device memory (trajectory workspace, pooled result arenas) back to the driver -- e.g. before
    the trainer needs the GPU's memory for itself, or after a torch out-of-memory error (torch's allocator cannot see it)."""
    check(lib().tw_release_cached_memory())


def device_count() -> int:
    return int(lib().tw_device_count())


def device_info() -> dict:
    info = DeviceInfo()
    check(lib().tw_get_device_info(C.byref(info)))
    return {"name": info.name.decode(), "arch": info.arch.decode(), "compute_units": info.compute_units,
            "wavefront_size": info.wavefront_size, "total_mem_bytes": info.total_mem_bytes,
            "lds_bytes_per_block": info.lds_bytes_per_block}
