"""Multi-GPU collection: episodes shard trivially over ranks, ONE gather of finished
trajectories to the root.

One process per GPU, `torch.distributed` ("nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  The RNG is keyed by the GLOBAL episode index, so whichever rank collects an episode the
gathered result is bit-identical to a single-GPU collect of all E episodes.  There is no
collective on the data path while collecting.

Episode ranges are dealt out CHUNK-MAJOR: the E episodes are cut into K steps (step_bounds) and
every step's range evenly into G contiguous pieces ("global chunks", K = pipeline steps, G =
ranks); in step s rank r collects global chunk s*G + r.
After each step:
  1. all_gather of (records, records of the chunk's last episode)   -- 16 B per rank
  2. point-to-point sends of every rank's compact SoA buffers to the root, received AT THEIR FINAL
     OFFSETS in the root's output buffers: the offset of global chunk g depends only on the record
     counts of the chunks before it, all of which are known once step s's counts have been
     gathered.  Each non-root rank has its own direct xGMI link to the root, so the G-1 transfers of
     a step run concurrently (a ring would be per-link bound), and the transfer of step s overlaps
     with the collection of step s+1.
The final order is the reference merge order [E-1, 0, 1, ..., E-2]
(rust/src/collector/collector.rs:40-46): the records of episode E-1 -- the tail of the very last
global chunk -- are received in front of everything else.  They are at most one episode long, so
the output buffers keep `max_episode_records` of slack in front and the result is a view; nothing
is staged and nothing is re-ordered on the device.
"""
from __future__ import annotations

import copy
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

FIELD_ORDER = ("obs", "logits", "perms", "values", "rewards", "actions", "advs", "rets", "remaining_values")

# Compute units the persistent rollout grid leaves free on every rank while a gather is in flight (tw_ppo_params.reserve_cus):
# RCCL's send/recv kernels are ordinary workgroups, and a rollout workgroup (512 threads, ~122 KB of LDS, 2 x 213 VGPRs per
# SIMD lane) leaves no room for another workgroup on its CU.  One CU per XCD.
DEFAULT_RESERVE_CUS = 8


def shard_range(num_episodes: int, rank: int, world: int):
    """Episode index range [start, end) owned by `rank` (one step)."""
    return (rank * num_episodes) // world, ((rank + 1) * num_episodes) // world


def step_bounds(num_episodes: int, world: int, chunks: int = 1, step_episodes: Optional[int] = None) -> List[int]:
    """Cumulative episode boundaries of the pipeline steps, from globally known quantities only (every rank issues the
    same collectives): step s covers the episodes [b[s-1], b[s]) (b[-1] = 0), dealt out evenly over the ranks.
    `step_episodes` (episodes per rank and step, e.g. the lanes a GPU keeps resident) takes precedence over `chunks`
    (that many steps of equal size)."""
    E, G = int(num_episodes), int(world)
    per_rank = -(-E // G)
    if step_episodes:
        K = max(1, -(-per_rank // int(step_episodes)))
        return [min(E, (s + 1) * G * int(step_episodes)) for s in range(K - 1)] + [E]
    K = max(1, min(int(chunks), per_rank))
    return [((s + 1) * E) // K for s in range(K)]


def pipeline_steps(num_episodes: int, world: int, chunks: int = 1, step_episodes: Optional[int] = None) -> int:
    return len(step_bounds(num_episodes, world, chunks, step_episodes))


def chunk_range(bounds: List[int], step: int, rank: int, world: int) -> Tuple[int, int]:
    """Episode index range [start, end) (possibly empty) that `rank` collects in `step` (chunk-major order)."""
    a = bounds[step - 1] if step > 0 else 0
    lo, hi = shard_range(bounds[step] - a, rank, world)
    return a + lo, a + hi


def plan_step(counts: List[int], lasts: List[int], last_step: bool, front: int, pos: int):
    """Placement of one step's chunks in the root's buffers (pure arithmetic; the C ABI's twin is tw_gather_plan, and
    tests/test_gather_plan.py holds the two against each other).  counts[r] / lasts[r]: records of rank r's chunk / of its
    last episode.  Returns (tail, tail_rank, pieces, pos_after): pieces[r] = [(src_lo, src_hi, dst_lo), ..] with dst_lo
    relative to the start of the output buffers; episode E-1 -- the last episode of the last NON-EMPTY chunk of the last
    step -- goes in front of everything else (collector.rs:40-46)."""
    world = len(counts)
    tail, tail_rank = 0, -1
    if last_step:
        for r in range(world - 1, -1, -1):
            if counts[r] > 0:
                tail, tail_rank = lasts[r], r
                break
    pieces, p = [], pos
    for r in range(world):
        n = counts[r]
        if n == 0:
            pieces.append([])
        elif r == tail_rank:
            body = n - tail
            pieces.append([(body, n, front - tail)] + ([(0, body, front + p)] if body > 0 else []))
        else:
            pieces.append([(0, n, front + p)])
        p += n - (tail if r == tail_rank else 0)
    return tail, tail_rank, pieces, p


class TrajectoryGather:
    """Gathers compact trajectories to `dst` in the reference merge order, step by step (see the module docstring).

    submit(fields, ep_len): this rank's chunk of the current step, episode-index order (merge_order=0): `fields[name]` has
        the record axis first; may be empty (zero records).  Asynchronous: the tensors are kept alive until finish().
    finish(): waits; on `dst` returns {name: tensor} in merge order (views of the output buffers), None elsewhere.

    steps > 1 needs `max_records` (an upper bound of the records of ALL ranks and steps, e.g. episodes x (horizon+1)) because
    the output buffers are allocated before the totals are known; they are kept and reused by the next round (`reset()`).
    """

    def __init__(self, dst: int = 0, group=None, steps: int = 1, max_records: Optional[int] = None,
                 max_episode_records: Optional[int] = None):
        self.dst, self.group, self.steps = dst, group, int(steps)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.steps > 1 and (max_records is None or max_episode_records is None):
            raise ValueError("TrajectoryGather: steps > 1 needs max_records and max_episode_records")
        self.max_records, self.max_episode_records = max_records, max_episode_records
        self.buf: Optional[Dict[str, torch.Tensor]] = None
        self.reset()

    def reset(self):
        self.step, self.pos, self.front, self.tail = 0, 0, 0, 0
        self.works, self.keep, self.names = [], [], None

    def _alloc(self, fields, names, total_first_step, tail_first_step):
        if self.steps == 1:                      # everything is known: exact size, no slack
            self.front, cap = tail_first_step, total_first_step
        else:
            self.front, cap = int(self.max_episode_records), int(self.max_records) + int(self.max_episode_records)
        ok = self.buf is not None and all(
            n in self.buf and self.buf[n].shape[0] >= cap and self.buf[n].shape[1:] == fields[n].shape[1:]
            and self.buf[n].dtype == fields[n].dtype and self.buf[n].device == fields[n].device for n in names)
        if not ok:
            self.buf = {n: torch.empty((cap,) + tuple(fields[n].shape[1:]), dtype=fields[n].dtype, device=fields[n].device)
                        for n in names}

    def submit(self, fields: Dict[str, torch.Tensor], ep_len: torch.Tensor) -> None:
        if self.step >= self.steps:
            raise RuntimeError("TrajectoryGather: more submits than steps")
        names = [n for n in FIELD_ORDER if n in fields]
        self.names = names
        last_step = self.step == self.steps - 1
        n_local = int(fields[names[0]].shape[0])
        last_len = int(ep_len[-1].item()) if ep_len.numel() else 0
        dev = fields[names[0]].device
        mine = torch.tensor([n_local, last_len], dtype=torch.int64, device=dev)
        allc = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allc, mine, group=self.group)
        counts = [int(c[0].item()) for c in allc]
        lasts = [int(c[1].item()) for c in allc]
        tail, tail_rank, pieces, _ = plan_step(counts, lasts, last_step, 0, self.pos)       # (the tail: front is not known yet)
        if self.rank == self.dst and self.step == 0:
            self._alloc(fields, names, sum(counts), tail)
        if last_step:
            self.tail = tail
        # (src_lo, src_hi, dst_lo) pieces per rank for this step; dst_lo relative to the start of the output buffers
        tail, tail_rank, pieces, p = plan_step(counts, lasts, last_step, self.front, self.pos)
        if self.rank == self.dst and self.front + p > next(iter(self.buf.values())).shape[0]:
            raise RuntimeError(f"TrajectoryGather: {p} records exceed max_records={self.max_records}")

        ops = []
        if self.rank == self.dst:
            for r in range(self.world):
                for (lo, hi, d) in pieces[r]:
                    for n in names:
                        if r == self.rank:
                            self.buf[n][d:d + hi - lo].copy_(fields[n][lo:hi])
                        else:
                            ops.append(dist.P2POp(dist.irecv, self.buf[n][d:d + hi - lo], r, self.group))
        else:
            for (lo, hi, _) in pieces[self.rank]:
                for n in names:
                    t = fields[n][lo:hi].contiguous()
                    self.keep.append(t)
                    ops.append(dist.P2POp(dist.isend, t, self.dst, self.group))
        self.keep.append(fields)
        self.works.append(dist.batch_isend_irecv(ops) if ops else [])
        self.pos = p
        self.step += 1

    def finish(self) -> Optional[Dict[str, torch.Tensor]]:
        if self.step != self.steps:
            raise RuntimeError(f"TrajectoryGather: {self.step} of {self.steps} steps submitted")
        for ws in self.works:
            for w in ws:
                w.wait()
        out = None
        if self.rank == self.dst:
            total, a = self.pos + self.tail, self.front - self.tail
            out = {n: self.buf[n][a:a + total] for n in self.names}
        self.reset()
        return out


class Comm:
    """RCCL communicator of the C ABI (tw_comm_*, include/twisterl_hip.h): what a non-Python host of the collectors uses for
    the multi-GPU exchange.  The unique id travels over whatever the host has; here: the torch.distributed group."""

    def __init__(self, group=None):
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        cid = _lib.CommId()
        if self.rank == 0:
            _lib.check(L.tw_comm_get_unique_id(C.byref(cid)))
        box = [C.string_at(C.byref(cid), 128) if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        C.memmove(C.byref(cid), box[0], 128)
        h = C.c_void_p()
        _lib.check(L.tw_comm_init(self.rank, self.world, C.byref(cid), C.byref(h)))
        self._h = h

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                from . import _lib
                _lib.lib().tw_comm_destroy(h)
            except Exception:          # interpreter shutting down
                pass

    __del__ = close

    def set_timeout_ms(self, ms: int) -> None:
        """Bound of every wait of the exchange (tw_comm_set_timeout_ms): past it the communicator is aborted and the call raises."""
        from . import _lib
        _lib.check(_lib.lib().tw_comm_set_timeout_ms(self._h, int(ms)))

    def broadcast_policy(self, policy, root: int = 0) -> None:
        """Policy sync across GPUs: the root's weight images replace everybody's (one ncclBroadcast)."""
        from . import _lib
        _lib.check(_lib.lib().tw_comm_broadcast_policy(self._h, policy._handle(), int(root)))


class RcclGather:
    """TrajectoryGather over the C ABI (tw_gather_*): same steps, same placement, RCCL send/recv issued by the library on
    the communicator's own stream.  submit() takes the CollectedData of this rank's chunk (or None) and the index of its
    first episode inside the gathered range; finish() returns the merged CollectedData on the root."""

    def __init__(self, comm: Comm, dst: int, steps: int, max_records: int, max_episode_records: int, total_episodes: int,
                 is_ppo: bool, n_cells: int):
        import ctypes as C
        from . import _lib
        self.comm, self.dst = comm, dst
        h = C.c_void_p()
        _lib.check(_lib.lib().tw_gather_begin(comm._h, int(dst), int(steps), int(max_records or 0), int(max_episode_records or 0),
                                              int(total_episodes), int(bool(is_ppo)), int(n_cells), C.byref(h)))
        self._h, self.keep = h, []

    def submit(self, data, episode_offset: int) -> None:
        from . import _lib
        self.keep.append(data)                    # the chunk's buffers are read by the transfer until finish()
        _lib.check(_lib.lib().tw_gather_submit(self._h, data._dev.h if data is not None else None, int(episode_offset)))

    def finish(self):
        import ctypes as C
        from . import _lib
        from .collector import CollectedData, _DeviceResult
        out = C.c_void_p()
        h, self._h = self._h, None
        _lib.check(_lib.lib().tw_gather_finish(h, C.byref(out)))
        self.keep = []
        return CollectedData._from_device(_DeviceResult(out.value)) if out.value else None


def gather_trajectories(fields: Dict[str, torch.Tensor], ep_len: torch.Tensor, dst: int = 0,
                        group=None) -> Optional[Dict[str, torch.Tensor]]:
    """One-step gather: every rank holds the compact trajectories of ITS contiguous episode shard (index order)."""
    tg = TrajectoryGather(dst=dst, group=group, steps=1)
    tg.submit(fields, ep_len)
    return tg.finish()


def broadcast_weights(tensors, src: int = 0, group=None) -> None:
    """Policy sync: one flat broadcast of all weight tensors (Puzzle-15: 264,197 f32 ~ 1.06 MB)."""
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def _split_fields(data):
    t = data.to_torch()
    ep_len = t.pop("ep_len")
    t.pop("ep_start", None)
    return t, ep_len


def _empty_like_fields(template: Dict[str, torch.Tensor]):
    return {k: v[:0] for k, v in template.items()}, torch.zeros((0,), dtype=torch.int64)


def collect_sharded(collector, env, policy, seed: int, dst: int = 0, group=None, gather: bool = True, chunks: int = 1,
                    max_episode_records: Optional[int] = None, gatherer: Optional[TrajectoryGather] = None,
                    reserve_cus: Optional[int] = None, step_episodes: Optional[int] = None, comm: Optional[Comm] = None):
    """Run `collector` (a PPOCollector/AZCollector configured with the GLOBAL num_episodes) over all ranks and gather to
    `dst`.  Returns (merged dict of device tensors on `dst` else None, list of this rank's CollectedData, one per non-empty
    chunk).  chunks > 1 collects in that many pipeline steps so that each step's transfer overlaps with the next step's
    collection; `max_episode_records` (= depth_slope*difficulty + 1 for Puzzle) is then required.  Pass a `gatherer` to
    reuse its output buffers between calls.  `reserve_cus` compute units stay free of rollout workgroups while a transfer
    can be in flight (default DEFAULT_RESERVE_CUS when pipelining on more than one rank, else 0).  `step_episodes` sets the
    episodes per rank and step instead of `chunks` -- with episodes of equal length a step should be a whole number of
    rounds of the GPU's resident lanes ((CUs - reserve_cus) x 256).  With `comm` (a Comm) the exchange runs inside the library
    (tw_gather_*: RCCL issued from C++, the path a non-Python host has) instead of torch.distributed's point-to-point ops."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    E = int(collector.num_episodes)
    bounds = step_bounds(E, world, chunks, step_episodes)
    K = len(bounds)
    if reserve_cus is None:
        reserve_cus = DEFAULT_RESERVE_CUS if (world > 1 and K > 1 and gather) else 0

    def run(a, b, last_step=False):
        local = copy.copy(collector)
        local.num_episodes = b - a
        local.episode_offset = collector.episode_offset + a
        local.merge_order = False
        if hasattr(local, "reserve_cus"):
            # nothing is collected after the last step, so it has no later transfer to make room for: it runs on every CU
            local.reserve_cus = 0 if last_step else int(reserve_cus)
        return local.collect(env, policy, seed=seed)

    if gather and comm is not None:
        from .env import get_env_desc
        desc = get_env_desc(env)
        rg = RcclGather(comm, dst, K, E * int(max_episode_records or 0) if K > 1 else 0, max_episode_records or 0, E,
                        getattr(collector, "_IS_PPO", True), int(desc.width) * int(desc.height))
        datas = []
        for s in range(K):
            a, b = chunk_range(bounds, s, rank, world)
            d = run(a, b, s == K - 1) if b > a else None
            if d is not None:
                datas.append(d)
            rg.submit(d, a)
        merged = rg.finish()
        return (merged.to_torch() if merged is not None else None), datas
    tg = None
    if gather:
        if K > 1 and max_episode_records is None:
            raise ValueError("collect_sharded: chunks > 1 needs max_episode_records")
        tg = gatherer
        if tg is None or tg.steps != K:
            tg = TrajectoryGather(dst=dst, group=group, steps=K,
                                  max_records=E * int(max_episode_records) if K > 1 else None,
                                  max_episode_records=max_episode_records)
    datas: List = []
    template = None
    for s in range(K):
        a, b = chunk_range(bounds, s, rank, world)
        if b > a:
            d = run(a, b, s == K - 1)
            datas.append(d)
            if tg is not None:
                template, ep_len = _split_fields(d)
                tg.submit(template, ep_len)
        elif tg is not None:
            # nothing to collect in this step (E < ranks x steps): the rank still joins the step's collectives, with
            # zero-length fields of the collector's layout
            if template is None:
                template = collector.empty_fields(env)
            tg.submit(*_empty_like_fields(template))
    return (tg.finish() if tg is not None else None), datas
