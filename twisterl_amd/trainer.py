"""Trainer hand-off: device-resident replacements for PPO.data_to_torch /
AZ.data_to_torch of the reference trainer (src/twisterl/rl/ppo.py:25-61, rl/az.py:28-46).

The reference walks Python lists: a dense one-hot float array built row by row, list -> tensor copies, then
`Categorical(logits).log_prob(actions)` and the optional advantage normalisation.  Here the collected data
never leaves the GPU: the HIP kernels of csrc/tw_trainer.hip write the same tensors, in the same order and
dtypes, straight into torch-allocated device memory (tw_collected_pack_trainer in include/twisterl_hip.h).
torch is used for allocation only.
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from .collector import CollectedData


def _dev(data: CollectedData):
    if not isinstance(data, CollectedData) or data._dev is None:
        raise RuntimeError("data_to_torch needs the device-resident CollectedData returned by collect() "
                           "(this one was modified or built on the host)")
    return data._dev


def _rows(d, rows):
    if rows is None:
        return 0, d.n
    lo, hi = int(rows[0]), int(rows[1])
    if not (0 <= lo <= hi <= d.n):
        raise ValueError(f"rows {rows} outside the {d.n} records")
    return lo, hi - lo


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def ppo_data_to_torch(data: CollectedData, obs_size: int, normalize_advantage: bool = False, rows=None, device="cuda"):
    """-> (pt_obs, pt_log_probs, pt_acts, pt_advs, pt_rets, pt_perm_idx), the tuple PPO.data_to_torch returns
    (ppo.py:25-61): one-hot float obs [n, obs_size], float log-probs of the taken actions, long actions,
    float advantages (normalised over the WHOLE collect when asked), float returns, long twist indices.
    rows=(lo, hi) hands over a mini-batch of records instead of everything."""
    import torch
    d = _dev(data)
    if not d.is_ppo:
        raise RuntimeError("ppo_data_to_torch: this is AlphaZero data (use az_data_to_torch)")
    lo, n = _rows(d, rows)
    dev = torch.device(device)
    pt_obs = torch.empty((n, int(obs_size)), dtype=torch.float32, device=dev)
    pt_logp = torch.empty((n,), dtype=torch.float32, device=dev)
    pt_acts = torch.empty((n,), dtype=torch.int64, device=dev)
    pt_perm = torch.empty((n,), dtype=torch.int64, device=dev)
    pt_advs = torch.empty((n,), dtype=torch.float32, device=dev)
    torch.cuda.current_stream(dev).synchronize()          # the library's stream is its own: order after torch's allocations
    _lib.check(_lib.lib().tw_collected_pack_trainer(d.h, int(obs_size), int(bool(normalize_advantage)), lo, n, _ptr(pt_obs),
                                                    _ptr(pt_logp), _ptr(pt_acts), _ptr(pt_perm), _ptr(pt_advs)))
    rets = data.to_torch()["rets"][lo:lo + n].clone()     # returns are handed over unchanged (ppo.py:48)
    _sync()
    return pt_obs, pt_logp, pt_acts, pt_advs, rets, pt_perm


def az_data_to_torch(data: CollectedData, obs_size: int, rows=None, device="cuda"):
    """-> (pt_obs, pt_probs, pt_vals) as AZ.data_to_torch (az.py:28-46): one-hot obs, MCTS probs, remaining values [n, 1]."""
    import torch
    d = _dev(data)
    if d.is_ppo:
        raise RuntimeError("az_data_to_torch: this is PPO data (use ppo_data_to_torch)")
    lo, n = _rows(d, rows)
    dev = torch.device(device)
    pt_obs = torch.empty((n, int(obs_size)), dtype=torch.float32, device=dev)
    torch.cuda.current_stream(dev).synchronize()
    _lib.check(_lib.lib().tw_collected_pack_trainer(d.h, int(obs_size), 0, lo, n, _ptr(pt_obs), None, None, None, None))
    t = data.to_torch()
    probs = t["logits"][lo:lo + n].clone()
    vals = t["remaining_values"][lo:lo + n].clone().unsqueeze(1)
    _sync()
    return pt_obs, probs, vals


def adv_stats(data: CollectedData):
    """(mean, unbiased std) of the advantages of the whole collect, computed on the device in f64."""
    d = _dev(data)
    m, s = C.c_double(), C.c_double()
    _lib.check(_lib.lib().tw_collected_adv_stats(d.h, C.byref(m), C.byref(s)))
    return m.value, s.value


def _sync():
    import torch
    torch.cuda.synchronize()
