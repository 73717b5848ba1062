"""CPU, world_size 2 and 3 over gloo: the N>1 path of the collector.

Each rank plays one GPU: it holds the compact trajectories of its episode shard (produced here by
the CPU oracle with episode_offset = the shard's first global episode, index order -- exactly what
tw_ppo_collect returns with merge_order=0) and calls the PRODUCT's gather
(twisterl_amd.dist.gather_trajectories).  Rank 0 must end up with the reference merge order
[E-1, 0, ..., E-2] (collector.rs:40-46), bit-identical to an un-sharded collect.
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


def _worker(rank, world, port, E, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import oracle as O
        from tests.util import make_policy_arrays
        from twisterl_amd.dist import broadcast_weights, gather_trajectories, shard_range

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
        lo, hi = shard_range(E, rank, world)
        d = O.ppo_collect(env, pol, hi - lo, 0.995, 0.995, seed=77, episode_offset=lo, arith=O.ARITH_CHAIN,
                          det_log=True, merge_order=False)
        fields = {
            "obs": torch.from_numpy(d.obs.astype(np.uint8)), "logits": torch.from_numpy(d.logits),
            "perms": torch.from_numpy(d.perms.astype(np.int8)), "values": torch.from_numpy(d.values),
            "rewards": torch.from_numpy(d.rewards), "actions": torch.from_numpy(d.actions.astype(np.uint8)),
            "advs": torch.from_numpy(d.additional_data["advs"]), "rets": torch.from_numpy(d.additional_data["rets"]),
        }
        merged = gather_trajectories(fields, torch.from_numpy(d.ep_len.astype(np.int64)), dst=0)
        # the pipelined form: the same shard handed over in K chunks of episodes must give the same result
        from twisterl_amd.dist import PipelinedGather
        K = 3
        pg = PipelinedGather(dst=0)
        for c in range(K):
            a, b = lo + ((hi - lo) * c) // K, lo + ((hi - lo) * (c + 1)) // K
            if b <= a:
                dc_fields = {k: v[:0] for k, v in fields.items()}
                pg.submit(dc_fields, torch.zeros((0,), dtype=torch.int64))
                continue
            dc = O.ppo_collect(env, pol, b - a, 0.995, 0.995, seed=77, episode_offset=a, arith=O.ARITH_CHAIN,
                               det_log=True, merge_order=False)
            pg.submit({
                "obs": torch.from_numpy(dc.obs.astype(np.uint8)), "logits": torch.from_numpy(dc.logits),
                "perms": torch.from_numpy(dc.perms.astype(np.int8)), "values": torch.from_numpy(dc.values),
                "rewards": torch.from_numpy(dc.rewards), "actions": torch.from_numpy(dc.actions.astype(np.uint8)),
                "advs": torch.from_numpy(dc.additional_data["advs"]), "rets": torch.from_numpy(dc.additional_data["rets"]),
            }, torch.from_numpy(dc.ep_len.astype(np.int64)))
        piped = pg.finish()
        if rank == 0:
            for k in merged:
                assert torch.equal(merged[k], piped[k]), k
        else:
            assert piped is None
        if rank == 0:
            full = O.ppo_collect(env, pol, E, 0.995, 0.995, seed=77, arith=O.ARITH_CHAIN, det_log=True, merge_order=True)
            ok = (np.array_equal(merged["obs"].numpy().astype(np.int64), full.obs)
                  and np.array_equal(merged["logits"].numpy().view(np.uint32), full.logits.view(np.uint32))
                  and np.array_equal(merged["values"].numpy().view(np.uint32), full.values.view(np.uint32))
                  and np.array_equal(merged["actions"].numpy().astype(np.int64), full.actions)
                  and np.array_equal(merged["perms"].numpy().astype(np.int32), full.perms)
                  and np.array_equal(merged["advs"].numpy().view(np.uint32), full.additional_data["advs"].view(np.uint32))
                  and np.array_equal(merged["rets"].numpy().view(np.uint32), full.additional_data["rets"].view(np.uint32)))
            q.put(("ok" if ok else "mismatch", int(merged["obs"].shape[0])))
        else:
            assert merged is None
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover - surfaced through the queue
        import traceback
        q.put(("error", traceback.format_exc()))
        raise


@pytest.mark.parametrize("world,E", [(2, 37), (3, 10), (2, 2), (4, 23)])
def test_sharded_gather_matches_unsharded_merge_order(oracle, world, E):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, q)) for r in range(world)]
    for p in procs:
        p.start()
    status, info = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert status == "ok", info


def test_shard_range_partitions_exactly():
    from twisterl_amd.dist import shard_range
    for E in (1, 7, 8, 262144, 2097152 + 3):
        for G in (1, 2, 3, 8):
            parts = [shard_range(E, r, G) for r in range(G)]
            assert parts[0][0] == 0 and parts[-1][1] == E
            assert all(parts[i][1] == parts[i + 1][0] for i in range(G - 1))
