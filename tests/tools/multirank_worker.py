"""One rank of tests/test_gpu_multirank.py: a fresh process that runs the library's multi-rank exchange (tw_comm_* /
tw_gather_*) against the other ranks ON THE SAME GPU through the host-staged stand-in for RCCL (tests/stub_rccl.hip, selected
with TW_RCCL_LIBRARY).  Not collected by pytest.

    python -m tests.tools.multirank_worker --rank R --world W --port P --stub LIB --scenario NAME --out FILE

Every rank writes {"rank", "ok", "checks": [names that passed], "error"} to FILE.  The unique id travels over a gloo group
(the host's own channel, as twisterl_amd.dist.Comm does it); all device data moves through the C ABI.
"""
import argparse
import hashlib
import json
import os
import signal
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _digest(arrs):
    h = hashlib.sha256()
    for k in sorted(arrs):
        h.update(k.encode()); h.update(arrs[k].tobytes())
    return h.hexdigest()


def _same(merged, want, keys):
    import numpy as np
    for k in keys:
        got = merged[k].cpu().numpy()
        if got.shape != want[k].shape or not np.array_equal(got.view(np.uint8) if got.dtype.kind == "f" else got,
                                                            want[k].view(np.uint8) if want[k].dtype.kind == "f" else want[k]):
            return k
    return None


PPO_KEYS = ("obs", "logits", "perms", "values", "rewards", "actions", "advs", "rets", "ep_len", "ep_start")
AZ_KEYS = ("obs", "logits", "perms", "remaining_values", "ep_len", "ep_start")


def scenario_exchange(rank, world, checks):
    """Healthy runs: policy broadcast, then sharded collects of every shape; the root's merged result byte-equal to the
    un-sharded collect (reference merge order [E-1, 0, .., E-2], rust/src/collector/collector.rs:40-46)."""
    import numpy as np
    import torch.distributed as dist
    import twisterl_amd
    from tests.util import amd_policy, make_deep_policy_arrays, make_policy_arrays, puzzle_transpose_twist
    from twisterl_amd.dist import Comm, collect_sharded
    tw = twisterl_amd.twisterl

    comm = Comm()
    assert (comm.rank, comm.world) == (rank, world)
    comm.set_timeout_ms(60_000)                      # a defect must fail the test, not hang the box
    op, ap = puzzle_transpose_twist(3)
    env = tw.env.Puzzle(3, 3, 5, 2, 256)

    def all_digests(d):
        box = [None] * world
        dist.all_gather_object(box, d)
        return box

    # --- broadcast_policy: every rank starts from ITS OWN weights; afterwards every rank computes with the root's
    for name, mk in (("mfma_policy", lambda s: amd_policy(make_policy_arrays(9, seed=s, emb=32, hidden=32), op, ap)),
                     ("deep_policy", lambda s: amd_policy(make_deep_policy_arrays(9, seed=s, emb=32, common=(48, 32), scale=2.0)))):
        gp = mk(20 + rank)
        probe = tw.collector.PPOCollector(64, 0.99, 0.95, 1)
        before = all_digests(_digest(probe.collect(env, gp, seed=3).to_numpy()))
        assert len(set(before)) == world, f"{name}: the ranks' own weights must give different collects"
        comm.broadcast_policy(gp, root=0)
        after = all_digests(_digest(probe.collect(env, gp, seed=3).to_numpy()))
        assert len(set(after)) == 1 and after[0] == before[0], f"{name}: after the broadcast every rank must compute rank 0's collect"
        checks.append(f"broadcast_{name}")
        if name == "mfma_policy":
            pol = gp

    # --- PPO: uneven shards, one step / three chunks / steps sized in episodes with CUs reserved
    def check(coll, env_, pol_, keys, tag, dst=0, **kw):
        want = coll.collect(env_, pol_, seed=3).to_numpy() if rank == dst else None
        merged, parts = collect_sharded(coll, env_, pol_, seed=3, dst=dst, comm=comm, **kw)
        if rank == dst:
            bad = _same(merged, want, keys)
            assert bad is None, f"{tag}: field {bad} of the merged result differs from the un-sharded collect"
            assert merged["obs"].shape[0] == want["obs"].shape[0]
        else:
            assert merged is None
        ns = all_digests(sum(len(p) for p in parts))
        if rank == dst:
            assert sum(ns) == want["obs"].shape[0], f"{tag}: the ranks collected {ns}"
        checks.append(tag)

    ppo = tw.collector.PPOCollector(301, 0.99, 0.95, 1)
    check(ppo, env, pol, PPO_KEYS, "ppo_one_step_uneven")
    check(ppo, env, pol, PPO_KEYS, "ppo_chunks3", chunks=3, max_episode_records=11)
    check(ppo, env, pol, PPO_KEYS, "ppo_step_episodes64_reserve8", step_episodes=64, max_episode_records=11, reserve_cus=8)
    check(ppo, env, pol, PPO_KEYS, "ppo_root_is_last_rank", dst=world - 1, chunks=2, max_episode_records=11)
    # more ranks than episodes: some ranks submit nothing at all, in one step and in several
    for E in sorted({1, world - 1}):
        few = tw.collector.PPOCollector(E, 0.99, 0.95, 1)
        check(few, env, pol, PPO_KEYS, f"ppo_{E}_episodes_on_{world}_ranks")
        check(few, env, pol, PPO_KEYS, f"ppo_{E}_episodes_on_{world}_ranks_chunks3", chunks=3, max_episode_records=11)
    # --- self-play
    az = tw.collector.AZCollector(91, 12, 1.41, 1, 1)
    check(az, env, pol, AZ_KEYS, "az_chunks4", chunks=4, max_episode_records=11)
    check(az, env, pol, AZ_KEYS, "az_one_step")
    # --- a 5 x 5 board: obs ids beyond 255 travel as two bytes (obs_width 2)
    op5, ap5 = puzzle_transpose_twist(5)
    pol5 = amd_policy(make_deep_policy_arrays(25, seed=5, emb=64, common=(64,), scale=2.0), op5, ap5)
    env5 = tw.env.Puzzle(5, 5, 6, 2, 256)
    check(tw.collector.PPOCollector(37, 0.995, 0.995, 1), env5, pol5, PPO_KEYS, "ppo_5x5_obs_width2_chunks2", chunks=2, max_episode_records=13)
    # --- Puzzle-15 with the twists of config 3, steps of 64 episodes
    op4, ap4 = puzzle_transpose_twist(4)
    pol4 = amd_policy(make_policy_arrays(16, seed=1, emb=64, hidden=32), op4, ap4)
    check(tw.collector.PPOCollector(203, 0.995, 0.995, 1), tw.env.Puzzle(4, 4, 7, 2, 256), pol4, PPO_KEYS, "ppo_puzzle15_twists_steps",
          step_episodes=32, max_episode_records=15, reserve_cus=8)
    comm.close()


def _raises(fn):
    try:
        fn()
    except (RuntimeError, ValueError) as e:          # TW_ERR_HIP -> RuntimeError, TW_ERR_INVALID -> ValueError (_lib.check)
        return str(e)
    return None


def scenario_faults(rank, world, checks):
    """A chunk that does not fit fails on EVERY rank, in the same tw_gather_submit, with the same message, and leaves the
    communicator usable."""
    import torch.distributed as dist
    import twisterl_amd
    from tests.util import amd_policy, make_policy_arrays
    from twisterl_amd.dist import Comm, RcclGather, collect_sharded
    tw = twisterl_amd.twisterl
    comm = Comm()
    comm.set_timeout_ms(60_000)
    pol = amd_policy(make_policy_arrays(9, seed=2, emb=32, hidden=32))
    env = tw.env.Puzzle(3, 3, 5, 2, 256)

    def gather_all(x):
        box = [None] * world
        dist.all_gather_object(box, x)
        return box

    def local(E, off, az=False):
        c = tw.collector.AZCollector(E, 8, 1.41, 1, 1) if az else tw.collector.PPOCollector(E, 0.99, 0.95, 1)
        c.merge_order = False
        c.episode_offset = off
        return c.collect(env, pol, seed=4)

    # (1) rank 1 holds far more records than max_records admits: tw_gather_plan refuses on every rank
    rg = RcclGather(comm, 0, 2, 10 * 11, 11, 10 * world, True, 9)            # 2 steps, room for 10 episodes of <= 11 records
    d = local(200 if rank == 1 else 2, 2 * rank)
    msg = _raises(lambda: rg.submit(d, 2 * rank))
    msgs = gather_all(msg)
    assert all(m is not None and "exceed max_records" in m for m in msgs), msgs
    assert len(set(msgs)) == 1, msgs
    assert _raises(rg.finish) is not None                                    # "0 of 2 steps submitted"; frees the gather
    checks.append("oversized_chunk_same_error_everywhere")
    # (2) rank 1 submits self-play data to a PPO gather: reported through the count exchange's status word
    rg = RcclGather(comm, 0, 1, 0, 0, 4 * world, True, 9)
    d = local(4, 4 * rank, az=(rank == 1))
    msgs = gather_all(_raises(lambda: rg.submit(d, 4 * rank)))
    assert all(m is not None and "rank 1 reports a chunk that does not fit" in m for m in msgs), msgs
    assert _raises(rg.finish) is not None
    checks.append("layout_mismatch_same_error_everywhere")
    # (3) the communicator survived both: a healthy gather still gives the un-sharded bytes
    coll = tw.collector.PPOCollector(50, 0.99, 0.95, 1)
    want = coll.collect(env, pol, seed=3).to_numpy() if rank == 0 else None
    merged, _ = collect_sharded(coll, env, pol, seed=3, comm=comm, chunks=2, max_episode_records=11)
    if rank == 0:
        assert _same(merged, want, PPO_KEYS) is None
    checks.append("communicator_usable_after_refusals")
    comm.close()


def scenario_killed_before_exchange(rank, world, checks):
    """The last rank is killed (SIGKILL) before it joins a step's count all-gather.  Every surviving rank must get TW_ERR_HIP
    out of the bounded wait (tw_comm_set_timeout_ms) instead of hanging, and its communicator must then refuse further work."""
    import twisterl_amd
    from tests.util import amd_policy, make_policy_arrays
    from twisterl_amd.dist import Comm, RcclGather
    tw = twisterl_amd.twisterl
    victim = world - 1
    comm = Comm()
    TIMEOUT = 1500
    comm.set_timeout_ms(TIMEOUT)
    pol = amd_policy(make_policy_arrays(9, seed=2, emb=32, hidden=32))
    env = tw.env.Puzzle(3, 3, 5, 2, 256)
    c = tw.collector.PPOCollector(6, 0.99, 0.95, 1)
    c.merge_order = False
    c.episode_offset = 6 * rank
    d = c.collect(env, pol, seed=4)
    rg = RcclGather(comm, 0, 2, 6 * world * 2 * 11, 11, 6 * world * 2, True, 9)
    if rank == victim:
        os.kill(os.getpid(), signal.SIGKILL)
    t0 = time.perf_counter()
    msg = _raises(lambda: rg.submit(d, 6 * rank))
    dt = time.perf_counter() - t0
    assert msg is not None and "did not complete within" in msg, msg
    assert dt < TIMEOUT / 1e3 + 3.0, dt
    checks.append(f"count_exchange_times_out_in_{dt:.2f}s")
    msg2 = _raises(rg.finish)
    assert msg2 is not None and "aborted" in msg2, msg2
    checks.append("finish_raises_after_failure")
    msg3 = _raises(lambda: RcclGather(comm, 0, 1, 0, 0, 4, True, 9))
    assert msg3 is not None and "aborted" in msg3, msg3
    checks.append("aborted_communicator_refuses_work")


def scenario_killed_transfers(rank, world, checks):
    """One step; the counts go round; the victim (last rank) accepts its sends without ever publishing them and is killed.  The
    root's tw_gather_submit succeeds (every receive is posted), tw_gather_finish gives TW_ERR_HIP within the timeout."""
    import twisterl_amd
    from tests.util import amd_policy, make_policy_arrays
    from twisterl_amd.dist import Comm, RcclGather
    tw = twisterl_amd.twisterl
    victim = world - 1
    comm = Comm()
    TIMEOUT = 1500
    comm.set_timeout_ms(TIMEOUT)
    pol = amd_policy(make_policy_arrays(9, seed=2, emb=32, hidden=32))
    env = tw.env.Puzzle(3, 3, 5, 2, 256)
    c = tw.collector.PPOCollector(6, 0.99, 0.95, 1)
    c.merge_order = False
    c.episode_offset = 6 * rank
    d = c.collect(env, pol, seed=4)
    rg = RcclGather(comm, 0, 1, 0, 0, 6 * world, True, 9)
    rg.submit(d, 6 * rank)
    checks.append("submit_ok")
    if rank == victim:
        os.kill(os.getpid(), signal.SIGKILL)
    t0 = time.perf_counter()
    msg = _raises(rg.finish)
    dt = time.perf_counter() - t0
    if rank == 0:
        assert msg is not None and "did not complete within" in msg, msg
        assert dt < TIMEOUT / 1e3 + 3.0, dt
        checks.append(f"finish_times_out_in_{dt:.2f}s")
        msg3 = _raises(lambda: RcclGather(comm, 0, 1, 0, 0, 4, True, 9))
        assert msg3 is not None and "aborted" in msg3, msg3
        checks.append("aborted_communicator_refuses_work")
    else:
        assert msg is None, msg                     # a healthy non-root rank's sends completed
        checks.append("healthy_sender_finishes")


def scenario_killed_mid_pipeline(rank, world, checks):
    """Two steps.  Step 1's counts go round, the victim (last rank) never publishes its records and is killed.  Every survivor's
    SECOND tw_gather_submit must fail within the timeout: the root's already behind its stuck receive (the bounded read of the
    chunk's last episode length on the exchange stream), the others' in the count exchange the victim never joins."""
    import twisterl_amd
    from tests.util import amd_policy, make_policy_arrays
    from twisterl_amd.dist import Comm, RcclGather
    tw = twisterl_amd.twisterl
    victim = world - 1
    comm = Comm()
    TIMEOUT = 1500
    comm.set_timeout_ms(TIMEOUT)
    pol = amd_policy(make_policy_arrays(9, seed=2, emb=32, hidden=32))
    env = tw.env.Puzzle(3, 3, 5, 2, 256)

    def chunk(off):
        c = tw.collector.PPOCollector(6, 0.99, 0.95, 1)
        c.merge_order = False
        c.episode_offset = off
        return c.collect(env, pol, seed=4)

    rg = RcclGather(comm, 0, 2, 12 * world * 11, 11, 12 * world, True, 9)
    rg.submit(chunk(6 * rank), 6 * rank)
    checks.append("first_submit_ok")
    if rank == victim:
        os.kill(os.getpid(), signal.SIGKILL)
    t0 = time.perf_counter()
    msg = _raises(lambda: rg.submit(chunk(6 * world + 6 * rank), 6 * world + 6 * rank))
    dt = time.perf_counter() - t0
    assert msg is not None and "did not complete within" in msg, msg
    assert dt < TIMEOUT / 1e3 + 3.0, dt
    checks.append(f"second_submit_times_out_in_{dt:.2f}s")
    msg2 = _raises(rg.finish)
    assert msg2 is not None and "aborted" in msg2, msg2
    checks.append("finish_raises_after_failure")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--stub", required=True)
    ap.add_argument("--scenario", required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.environ["TW_RCCL_LIBRARY"] = a.stub
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(a.port)
    if a.scenario in ("killed_transfers", "killed_mid_pipeline") and a.rank == a.world - 1:
        os.environ["TWSTUB_DROP_SENDS"] = "1"
    res = {"rank": a.rank, "ok": False, "checks": [], "error": None}

    def write():
        with open(a.out + ".tmp", "w") as f:
            json.dump(res, f)
        os.replace(a.out + ".tmp", a.out)

    write()                                         # (a killed rank leaves this behind)
    try:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
        if a.scenario == "exchange":
            scenario_exchange(a.rank, a.world, res["checks"])
        elif a.scenario == "faults":
            scenario_faults(a.rank, a.world, res["checks"])
        elif a.scenario == "killed_before_exchange":
            scenario_killed_before_exchange(a.rank, a.world, res["checks"])
        elif a.scenario == "killed_mid_pipeline":
            scenario_killed_mid_pipeline(a.rank, a.world, res["checks"])
        elif a.scenario == "killed_transfers":
            scenario_killed_transfers(a.rank, a.world, res["checks"])
        else:
            raise ValueError(a.scenario)
        res["ok"] = True
    except BaseException:
        res["error"] = traceback.format_exc()
    write()
    sys.stdout.flush(); sys.stderr.flush()
    # no group teardown: after a killed peer gloo's destroy would wait for it; the process simply ends
    os._exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
