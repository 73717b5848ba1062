"""Multi-GPU collection: episodes shard trivially over ranks, ONE gather of finished
trajectories to the root.

One process per GPU, `torch.distributed` ("nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  Rank g collects the global episodes [g*E/G, (g+1)*E/G) with the RNG keyed by the GLOBAL
episode index, so the gathered result is bit-identical to a single-GPU collect of all E episodes.
There is no collective on the data path while collecting; afterwards
  1. all_gather of (records, records of the rank's last episode)   -- 16 B per rank
  2. point-to-point send of each rank's compact SoA buffers to the root (each non-root rank has
     its own direct xGMI link to the root, so the 7 transfers run concurrently; a ring would be
     per-link bound), placed so that the final order is the reference merge order
     [E-1, 0, 1, ..., E-2] (rust/src/collector/collector.rs:40-46).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

FIELD_ORDER = ("obs", "logits", "perms", "values", "rewards", "actions", "advs", "rets", "remaining_values")


def shard_range(num_episodes: int, rank: int, world: int):
    """Episode index range [start, end) owned by `rank`."""
    return (rank * num_episodes) // world, ((rank + 1) * num_episodes) // world


def gather_trajectories(fields: Dict[str, torch.Tensor], ep_len: torch.Tensor, dst: int = 0,
                        group=None) -> Optional[Dict[str, torch.Tensor]]:
    """Gather per-rank compact trajectories (episode-index order inside each rank) to `dst` in the
    reference merge order.  `fields[name]` has the record axis first; `ep_len` is this rank's
    per-episode record count.  Returns the merged dict on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    names = [n for n in FIELD_ORDER if n in fields]
    n_local = int(fields[names[0]].shape[0])
    last_len = int(ep_len[-1].item()) if ep_len.numel() else 0
    dev = fields[names[0]].device
    mine = torch.tensor([n_local, last_len], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine, group=group)
    counts = [int(c[0].item()) for c in allc]
    tail = int(allc[world - 1][1].item())           # records of global episode E-1
    total = sum(counts)
    # destination offsets: [tail of last rank][rank 0][rank 1]...[rank G-1 without its tail]
    starts, pos = [], tail
    for r in range(world):
        starts.append(pos)
        pos += counts[r] - (tail if r == world - 1 else 0)
    assert pos == total

    def pieces(r):
        """(src_lo, src_hi, dst_lo) slices rank r contributes"""
        if r == world - 1:
            body = counts[r] - tail
            return [(body, counts[r], 0), (0, body, starts[r])]
        return [(0, counts[r], starts[r])]

    out = None
    ops = []
    if rank == dst:
        out = {n: torch.empty((total,) + tuple(fields[n].shape[1:]), dtype=fields[n].dtype, device=dev) for n in names}
        for r in range(world):
            for (lo, hi, d) in pieces(r):
                if hi <= lo:
                    continue
                for n in names:
                    if r == rank:
                        out[n][d:d + hi - lo].copy_(fields[n][lo:hi])
                    else:
                        ops.append(dist.P2POp(dist.irecv, out[n][d:d + hi - lo], r, group))
    else:
        for (lo, hi, _) in pieces(rank):
            if hi <= lo:
                continue
            for n in names:
                ops.append(dist.P2POp(dist.isend, fields[n][lo:hi].contiguous(), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


class PipelinedGather:
    """Gather that overlaps with collection: a rank's shard is collected in K chunks of episodes; the trajectories
    of chunk c travel to `dst` (point-to-point, as in gather_trajectories) while chunk c+1 is being collected, so
    only the last chunk's transfer and one on-device reorder at `dst` are exposed.  xGMI moves ~3.4 GB per rank and
    Puzzle-15 step into the root at link rate (~50 ms for 7 senders): a third of the collection time if serialised.

    submit(fields, ep_len): chunk in episode-index order (merge_order=0), asynchronous (keeps the tensors alive);
    finish(): waits and returns, on `dst`, every field in the reference merge order [E-1, 0, ..., E-2]
    (rust/src/collector/collector.rs:40-46), bit-identical to gather_trajectories of the un-chunked shard."""

    def __init__(self, dst: int = 0, group=None):
        self.dst, self.group = dst, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.chunks = []          # per chunk: (counts per rank, tail of the last rank, staging dict | None)
        self.works, self.keep = [], []
        self.names = None

    def submit(self, fields: Dict[str, torch.Tensor], ep_len: torch.Tensor) -> None:
        names = [n for n in FIELD_ORDER if n in fields]
        self.names = names
        n_local = int(fields[names[0]].shape[0])
        last_len = int(ep_len[-1].item()) if ep_len.numel() else 0
        dev = fields[names[0]].device
        mine = torch.tensor([n_local, last_len], dtype=torch.int64, device=dev)
        allc = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allc, mine, group=self.group)
        counts = [int(c[0].item()) for c in allc]
        tail = int(allc[self.world - 1][1].item())
        ops, staging = [], None
        if self.rank == self.dst:
            total = sum(counts)
            staging = {n: torch.empty((total,) + tuple(fields[n].shape[1:]), dtype=fields[n].dtype, device=dev) for n in names}
            off = 0
            for r in range(self.world):
                if counts[r]:
                    for n in names:
                        if r == self.rank:
                            staging[n][off:off + counts[r]].copy_(fields[n])
                        else:
                            ops.append(dist.P2POp(dist.irecv, staging[n][off:off + counts[r]], r, self.group))
                off += counts[r]
        elif n_local:
            for n in names:
                t = fields[n].contiguous()
                self.keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, self.dst, self.group))
        self.keep.append(fields)
        self.works.append(dist.batch_isend_irecv(ops) if ops else [])      # per chunk: finish() waits chunk by chunk
        self.chunks.append((counts, tail, staging))

    def finish(self) -> Optional[Dict[str, torch.Tensor]]:
        if self.rank != self.dst:
            for ws in self.works:
                for w in ws:
                    w.wait()
            self.works, self.keep = [], []
            return None
        names, world, K = self.names, self.world, len(self.chunks)
        tail = self.chunks[-1][1] if K else 0                       # records of global episode E-1
        total = sum(sum(c[0]) for c in self.chunks)
        first = self.chunks[0][2]
        out = {n: torch.empty((total,) + tuple(first[n].shape[1:]), dtype=first[n].dtype, device=first[n].device) for n in names}
        # [tail][rank 0: chunk 0..K-1][rank 1: ...]...[rank G-1: ... without its tail]
        plan = [[] for _ in range(K)]          # per chunk: (src_lo, src_hi, dst_lo)
        pos = tail
        for r in range(world):
            for c, (counts, _, _) in enumerate(self.chunks):
                src = sum(counts[:r])
                n_rec = counts[r]
                if r == world - 1 and c == K - 1:
                    body = n_rec - tail
                    plan[c].append((src + body, src + n_rec, 0))
                    n_rec = body
                if n_rec > 0:
                    plan[c].append((src, src + n_rec, pos))
                pos += n_rec
        assert pos == total
        # chunk by chunk: the reorder of the early chunks runs while the last chunk is still arriving
        for c, (_, _, staging) in enumerate(self.chunks):
            for w in self.works[c]:
                w.wait()
            for (lo, hi, d) in plan[c]:
                if hi > lo:
                    for n in names:
                        out[n][d:d + hi - lo].copy_(staging[n][lo:hi])
        self.works, self.keep, self.chunks = [], [], []
        return out


def broadcast_weights(tensors, src: int = 0, group=None) -> None:
    """Policy sync: one flat broadcast of all weight tensors (Puzzle-15: 264,197 f32 ~ 1.06 MB)."""
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def collect_sharded(collector, env, policy, seed: int, dst: int = 0, group=None, gather: bool = True, chunks: int = 1):
    """Run `collector` (a PPOCollector/AZCollector configured with the GLOBAL num_episodes) on this
    rank's shard and gather to `dst`.  Returns (merged dict of device tensors or None, local
    CollectedData -- a list of them, one per chunk, when chunks > 1).  chunks > 1 collects the shard in that
    many pieces and overlaps each piece's transfer with the collection of the next (PipelinedGather)."""
    import copy
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(collector.num_episodes, rank, world)
    chunks = max(1, min(int(chunks), hi - lo)) if hi > lo else 1

    def run(a, b):
        local = copy.copy(collector)
        local.num_episodes = b - a
        local.episode_offset = collector.episode_offset + a
        local.merge_order = False
        return local.collect(env, policy, seed=seed)

    if chunks == 1:
        data = run(lo, hi)
        if not gather:
            return None, data
        t = data.to_torch()
        ep_len = t.pop("ep_len")
        t.pop("ep_start", None)
        return gather_trajectories(t, ep_len, dst=dst, group=group), data
    pg = PipelinedGather(dst=dst, group=group) if gather else None
    datas = []
    for c in range(chunks):
        a, b = lo + ((hi - lo) * c) // chunks, lo + ((hi - lo) * (c + 1)) // chunks
        d = run(a, b)
        datas.append(d)
        if pg is not None:
            t = d.to_torch()
            ep_len = t.pop("ep_len")
            t.pop("ep_start", None)
            pg.submit(t, ep_len)
    return (pg.finish() if pg is not None else None), datas
