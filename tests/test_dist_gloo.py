"""CPU, world_size 2, 3, 4 and 8 over gloo: the N>1 path of the collector.

Each rank plays one GPU.  The PRODUCT's sharding and gather logic (twisterl_amd.dist: chunk-major episode ranges,
count all_gather, point-to-point receives at their final offsets, the E-1 rotation) runs unchanged; only the per-rank
collect is stood in for by the CPU oracle behind the collector attributes dist.collect_sharded uses (num_episodes,
episode_offset, merge_order, collect(), to_torch()) -- exactly what tw_ppo_collect returns with merge_order=0.  Rank 0
must end up with the reference merge order [E-1, 0, ..., E-2] (collector.rs:40-46), bit-identical to an un-sharded
collect, for even and uneven shards, more ranks than episodes, and any number of pipeline steps.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _OracleData:
    def __init__(self, d):
        self.d = d

    def __len__(self):
        return int(self.d.obs.shape[0])

    def to_torch(self):
        d = self.d
        return {
            "obs": torch.from_numpy(d.obs.astype(np.uint8)), "logits": torch.from_numpy(d.logits),
            "perms": torch.from_numpy(d.perms.astype(np.int8)), "values": torch.from_numpy(d.values),
            "rewards": torch.from_numpy(d.rewards), "actions": torch.from_numpy(d.actions.astype(np.uint8)),
            "advs": torch.from_numpy(d.additional_data["advs"]), "rets": torch.from_numpy(d.additional_data["rets"]),
            "ep_len": torch.from_numpy(d.ep_len.astype(np.int64)),
        }


class _OracleCollector:
    """Stand-in for twisterl_amd.collector.PPOCollector on a box without a GPU (test double, not a product path)."""

    def __init__(self, O, num_episodes):
        self.O, self.num_episodes, self.episode_offset, self.merge_order, self.reserve_cus = O, num_episodes, 0, True, 0
        self.calls = []

    def collect(self, env, policy, seed=None):
        self.calls.append((self.episode_offset, self.num_episodes, self.reserve_cus))
        return _OracleData(self.O.ppo_collect(env, policy, self.num_episodes, 0.995, 0.995, seed=seed, episode_offset=self.episode_offset,
                                              arith=self.O.ARITH_CHAIN, det_log=True, merge_order=self.merge_order))

    def empty_fields(self, env):
        z = lambda shape, dt: torch.empty(shape, dtype=dt)
        return {"obs": z((0, env.n_cells), torch.uint8), "logits": z((0, 4), torch.float32), "perms": z((0,), torch.int8),
                "values": z((0,), torch.float32), "rewards": z((0,), torch.float32), "actions": z((0,), torch.uint8),
                "advs": z((0,), torch.float32), "rets": z((0,), torch.float32)}


def _same(merged, full):
    return (np.array_equal(merged["obs"].numpy().astype(np.int64), full.obs)
            and np.array_equal(merged["logits"].numpy().view(np.uint32), full.logits.view(np.uint32))
            and np.array_equal(merged["values"].numpy().view(np.uint32), full.values.view(np.uint32))
            and np.array_equal(merged["rewards"].numpy().view(np.uint32), full.rewards.view(np.uint32))
            and np.array_equal(merged["actions"].numpy().astype(np.int64), full.actions)
            and np.array_equal(merged["perms"].numpy().astype(np.int32), full.perms)
            and np.array_equal(merged["advs"].numpy().view(np.uint32), full.additional_data["advs"].view(np.uint32))
            and np.array_equal(merged["rets"].numpy().view(np.uint32), full.additional_data["rets"].view(np.uint32)))


def _worker(rank, world, port, E, chunks, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import oracle as O
        from tests.util import make_policy_arrays
        from twisterl_amd.dist import (TrajectoryGather, broadcast_weights, chunk_range, collect_sharded, gather_trajectories,
                                       pipeline_steps, shard_range, step_bounds)

        arrs = make_policy_arrays(9, seed=5, emb=32, hidden=32)
        # policy sync: rank 0 owns the weights, the others receive them in one flat broadcast
        flat = [torch.from_numpy(np.array(a, copy=True)) for a in (arrs[0], arrs[1], arrs[2][0][0], arrs[2][0][1])]
        if rank != 0:
            for t in flat:
                t.zero_()
        broadcast_weights(flat, src=0)
        assert np.array_equal(flat[0].numpy(), arrs[0]) and np.array_equal(flat[3].numpy(), arrs[2][0][1])

        pol = O.Policy(*arrs)
        env = O.Puzzle(3, 3, 6, 2, 256)
        t_max = 2 * 6 + 1
        full = O.ppo_collect(env, pol, E, 0.995, 0.995, seed=77, arith=O.ARITH_CHAIN, det_log=True, merge_order=True) if rank == 0 else None

        # (1) one-step gather of contiguous shards (gather_trajectories), when every rank has episodes
        ok1 = True
        if E >= world:
            lo, hi = shard_range(E, rank, world)
            d = _OracleData(O.ppo_collect(env, pol, hi - lo, 0.995, 0.995, seed=77, episode_offset=lo, arith=O.ARITH_CHAIN,
                                          det_log=True, merge_order=False)).to_torch()
            ep_len = d.pop("ep_len")
            merged = gather_trajectories(d, ep_len, dst=0)
            ok1 = _same(merged, full) if rank == 0 else merged is None

        # (2) the whole sharded path through collect_sharded: `chunks` pipeline steps, twice with one reused gatherer
        coll = _OracleCollector(O, E)
        step_eps = 3 if chunks == 0 else None            # chunks == 0: steps of 3 episodes per rank instead (the last one shorter)
        bounds = step_bounds(E, world, chunks, step_eps)
        K = len(bounds)
        assert K == pipeline_steps(E, world, chunks, step_eps)
        tg = TrajectoryGather(dst=0, steps=K, max_records=E * t_max if K > 1 else None, max_episode_records=t_max)
        ok2 = True
        for rep in range(2):
            merged, datas = collect_sharded(coll, env, pol, seed=77, dst=0, chunks=chunks, max_episode_records=t_max, gatherer=tg,
                                            step_episodes=step_eps)
            ok2 = ok2 and (_same(merged, full) if rank == 0 else merged is None)
        # every rank issued K steps; its collects cover exactly its chunk-major ranges; CUs are reserved only while pipelining
        mine = [chunk_range(bounds, s, rank, world) for s in range(K)]
        want = [(a, b - a, 8 if (world > 1 and K > 1 and s < K - 1) else 0) for s, (a, b) in enumerate(mine) if b > a]
        ok3 = coll.calls == want + want
        covered = sorted(chunk_range(bounds, s, r, world) for s in range(K) for r in range(world))
        ok3 = ok3 and covered[0][0] == 0 and covered[-1][1] == E and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
        if rank == 0:
            q.put(("ok" if (ok1 and ok2 and ok3) else f"mismatch {ok1} {ok2} {ok3}", int(full.obs.shape[0])))
        else:
            assert ok1 and ok2 and ok3
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover - surfaced through the queue
        import traceback
        q.put(("error", traceback.format_exc()))
        raise


@pytest.mark.parametrize("world,E,chunks", [(2, 37, 3), (3, 10, 4), (2, 2, 1), (4, 23, 2), (3, 2, 4), (4, 1, 3), (2, 9, 1), (2, 20, 0), (3, 11, 0),
                                            (8, 50, 3), (8, 100, 0)])     # the driver's widest run: eight ranks
def test_sharded_gather_matches_unsharded_merge_order(oracle, world, E, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    status, info = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert status == "ok", info


def test_shard_range_partitions_exactly():
    from twisterl_amd.dist import chunk_range, shard_range, step_bounds
    for E in (1, 7, 8, 262144, 2097152 + 3):
        for G in (1, 2, 3, 8):
            parts = [shard_range(E, r, G) for r in range(G)]
            assert parts[0][0] == 0 and parts[-1][1] == E
            assert all(parts[i][1] == parts[i + 1][0] for i in range(G - 1))
            for chunks, step_eps in ((1, None), (4, None), (1, 63488)):
                b = step_bounds(E, G, chunks, step_eps)
                ch = [chunk_range(b, s, r, G) for s in range(len(b)) for r in range(G)]
                assert ch[0][0] == 0 and ch[-1][1] == E and all(ch[i][1] == ch[i + 1][0] for i in range(len(ch) - 1))
                assert ch[-1][1] > ch[-1][0]            # the last global chunk holds episode E-1
