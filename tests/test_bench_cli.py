"""CPU: bench.py's launch logic.  `python bench.py --gpus N` started plainly (the way the driver starts N=1) must plan N ranks
of itself under torch.distributed.run as a CHILD process before anything touches the GPU; under a launcher whose
WORLD_SIZE disagrees with --gpus it must stop with a non-zero exit code, never report `n_gpus: 1`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_dry_launch_plans_two_ranks():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-launch"])
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["n_ranks"] == 2 and [x["rank"] for x in plan["ranks"]] == [0, 1]
    assert [x["device"] for x in plan["ranks"]] == ["cuda:0", "cuda:1"]
    cmd = plan["command"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index(BENCH) + 1:] == ["--gpus", "2", "--steps", "5", "--warmup", "1"]     # the child gets the same arguments


def test_dry_launch_single_gpu_needs_no_launcher():
    r = _run(["--dry-launch"])
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["n_ranks"] == 1 and plan["command"] is None


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "8"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and '"n_gpus"' not in r.stdout
    r = _run(["--gpus", "1"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout


def test_self_launch_relays_the_childs_failure():
    """No GPU here: the two ranks fail ("needs a GPU"), and the parent must pass that on as a non-zero exit code without
    printing a result line."""
    import twisterl_amd
    if twisterl_amd.device_count() > 0:
        import pytest
        pytest.skip("GPU present")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and '"metric"' not in r.stdout
