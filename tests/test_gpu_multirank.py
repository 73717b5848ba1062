"""GPU: the library's own multi-rank exchange (tw_comm_* / tw_gather_*, twisterl_amd/csrc/tw_comm.hip) with 2 and 3 RANKS,
each a fresh process with its own HIP context, device buffers and communicator, on the one GPU a test box has.

RCCL refuses two ranks on one device, so the eleven nccl* symbols the library resolves with dlopen come from
tests/stub_rccl.hip (TW_RCCL_LIBRARY): stream-ordered, asynchronous, host-staged through /dev/shm (a polling kernel holds the stream until a message is in).  Everything ABOVE those
symbols is the product: the count all-gather with its status word, tw_gather_plan's placement, the grouped send/recv posting
at final offsets in the root's result, ep_len / ep_start of the merged result, the policy image broadcast with its pointer
table fix-up, and the abort / bounded-wait path.  The bar: rank 0's merged result is BYTE-EQUAL to the un-sharded collect
(reference merge order [E-1, 0, .., E-2], rust/src/collector/collector.rs:40-46) -- and the un-sharded collect
is what tests/test_gpu_parity.py holds against the oracle.

Children are started as new processes (never an exec of this one); at most 3 ranks + this process use the GPU at once.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.util import build_stub_rccl

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_ranks(tmp_path, world, scenario, timeout=420, expect_killed=()):
    import twisterl_amd
    assert twisterl_amd.device_count() >= 1, "no GPU visible: the -m gpu tests need the MI355X box"
    stub = build_stub_rccl()
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("TWSTUB_DROP_SENDS", None)
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"{scenario}_w{world}_r{r}.json")
        log = open(str(tmp_path / f"{scenario}_w{world}_r{r}.log"), "w")
        cmd = [sys.executable, "-m", "tests.tools.multirank_worker", "--rank", str(r), "--world", str(world), "--port", str(port),
               "--stub", stub, "--scenario", scenario, "--out", out]
        procs.append((subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT), log))
        outs.append(out)
    codes = []
    try:
        for p, _ in procs:
            codes.append(p.wait(timeout=timeout))
    finally:
        for p, log in procs:
            if p.poll() is None:
                p.kill()                       # exactly the children started above
                p.wait()
            log.close()
    results = []
    for r in range(world):
        res = json.load(open(outs[r]))
        logtxt = open(str(tmp_path / f"{scenario}_w{world}_r{r}.log")).read()[-3000:]
        if r in expect_killed:
            assert codes[r] == -9, (r, codes[r], logtxt)
        else:
            assert codes[r] == 0 and res["ok"], f"rank {r} of {world} ({scenario}) exit {codes[r]}:\n{res['error']}\n--- log ---\n{logtxt}"
        results.append(res)
    # nothing may be left behind in /dev/shm by a healthy run
    return results


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_collects_through_the_library_exchange(tmp_path, world):
    """Policy broadcast; PPO and self-play; uneven shards; more ranks than episodes; chunks=3; step_episodes=64 with
    reserve_cus=8; a non-zero root; a 5 x 5 board (two-byte obs ids); Puzzle-15 with twists: the root's merged result byte-equal
    to the un-sharded collect, every field incl. ep_len / ep_start."""
    res = _run_ranks(tmp_path, world, "exchange")
    want = {"broadcast_mfma_policy", "broadcast_deep_policy", "ppo_one_step_uneven", "ppo_chunks3", "ppo_step_episodes64_reserve8",
            "ppo_root_is_last_rank", f"ppo_1_episodes_on_{world}_ranks", f"ppo_1_episodes_on_{world}_ranks_chunks3", "az_chunks4", "az_one_step",
            "ppo_5x5_obs_width2_chunks2", "ppo_puzzle15_twists_steps"}
    for r in res:
        assert want <= set(r["checks"]), (r["rank"], sorted(want - set(r["checks"])))


@pytest.mark.parametrize("world", [2, 3])
def test_a_chunk_that_does_not_fit_fails_on_every_rank_alike(tmp_path, world):
    res = _run_ranks(tmp_path, world, "faults")
    for r in res:
        assert r["checks"] == ["oversized_chunk_same_error_everywhere", "layout_mismatch_same_error_everywhere", "communicator_usable_after_refusals"]


@pytest.mark.parametrize("world", [2, 3])
def test_killed_peer_before_the_count_exchange(tmp_path, world):
    """Every survivor's tw_gather_submit returns TW_ERR_HIP within tw_comm_set_timeout_ms; the communicator is aborted."""
    res = _run_ranks(tmp_path, world, "killed_before_exchange", timeout=120, expect_killed=(world - 1,))
    for r in res[:-1]:
        assert "aborted_communicator_refuses_work" in r["checks"], r


@pytest.mark.parametrize("world", [2, 3])
def test_killed_peer_between_count_exchange_and_transfers(tmp_path, world):
    """The root's tw_gather_finish returns TW_ERR_HIP within tw_comm_set_timeout_ms; a healthy sender's finish succeeds."""
    res = _run_ranks(tmp_path, world, "killed_transfers", timeout=120, expect_killed=(world - 1,))
    assert any(c.startswith("finish_times_out_in_") for c in res[0]["checks"]) and "aborted_communicator_refuses_work" in res[0]["checks"], res[0]
    if world == 3:
        assert "healthy_sender_finishes" in res[1]["checks"], res[1]


@pytest.mark.parametrize("world", [2, 3])
def test_killed_peer_in_the_middle_of_a_pipelined_gather(tmp_path, world):
    """Step 1's records of the victim never arrive: every survivor's step-2 tw_gather_submit returns TW_ERR_HIP within the timeout
    (the root's while it reads its chunk's last episode length behind the stuck receive) instead of waiting for ever."""
    res = _run_ranks(tmp_path, world, "killed_mid_pipeline", timeout=120, expect_killed=(world - 1,))
    for r in res[:-1]:
        assert r["checks"][0] == "first_submit_ok" and r["checks"][-1] == "finish_raises_after_failure" and len(r["checks"]) == 3, r


def test_bench_line_of_two_ranks_rehearsed_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` started the way the driver starts it (torch.distributed.run, one process per rank), on the ONE GPU of a test
    box: TW_BENCH_REHEARSAL puts every rank on cuda:0 with gloo for the barriers and the library's exchange (tw_gather_*) over the
    stand-in transport.  Every line of the N > 1 path runs on device memory -- shards of the seeded workload, pipeline steps with
    reserved CUs, the gather at final offsets inside the timed region, MAX-over-ranks time, one JSON line from rank 0 -- which no
    8-GPU node has run yet.  The rate it prints is not a measurement (two ranks share a GPU) and the line says so."""
    import twisterl_amd
    assert twisterl_amd.device_count() >= 1, "no GPU visible: the -m gpu tests need the MI355X box"
    stub = build_stub_rccl()
    env = dict(os.environ, TW_BENCH_REHEARSAL="1", TW_RCCL_LIBRARY=stub, MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TW_GATHER", "TW_STEP_EPISODES", "TW_RESERVE_CUS", "TWSTUB_DROP_SENDS"):
        env.pop(k, None)
    envs = 65536
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--envs", str(envs), "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]                        # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and "rehearsal" in d
    assert d["config"]["total_envs"] == 2 * envs and d["config"]["envs_per_gpu"] == envs
    cus = twisterl_amd.device_info()["compute_units"]
    g = d["config"]["gather"]
    assert g["reserved_cus"] == 8 and g["episodes_per_rank_and_step"] == (cus - 8) * 256 and g["pipeline_steps"] == 2
    assert "the library" in g["transport"]
    # the untrained policy solves next to nothing at difficulty 128: close to 257 records per episode, from BOTH ranks
    assert 0.98 * 257 * 2 * envs <= d["config"]["records_per_step"] <= 257 * 2 * envs
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - d["config"]["records_per_step"]) <= 1e-6 * d["config"]["records_per_step"]
