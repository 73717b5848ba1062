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


def test_side_config_schema():
    """bench.py's side entries (BASELINE.json configs 1, 2 and 5 on the driver's clock): the keys and fields the line carries,
    checked against a stand-in collector (no GPU here; on the GPU box the real line is produced by bench.py itself)."""
    import types
    sys.path.insert(0, ROOT)
    import bench
    import torch

    class _Data:
        def __init__(self, n, stats):
            self._n, self.stats = n, stats

        def __len__(self):
            return self._n

    class _Coll:
        def __init__(self, *a, **k):
            self.E = a[0] if a else k["num_episodes"]

        def collect(self, env, pol, seed=None):
            return _Data(10 * self.E, {"ms_rollout": 1.0, "forward_evals": 40 * self.E, "reused_evals": 10 * self.E, "speculative_evals": 0,
                                       "rollout_blocks": 256, "rollout_threads": 512})

    fake = types.SimpleNamespace(collector=types.SimpleNamespace(PPOCollector=_Coll, AZCollector=_Coll, evaluate=lambda env, pol, **kw: (0.5, -0.25)),
                                 env=types.SimpleNamespace(Puzzle=lambda *a: None))
    real_build, real_sync = bench.build_policy, torch.cuda.synchronize
    bench.build_policy, torch.cuda.synchronize = (lambda *a: None), (lambda: None)
    try:
        out = bench.side_configs(fake, torch)
    finally:
        bench.build_policy, torch.cuda.synchronize = real_build, real_sync
    assert set(out) == {"config1_puzzle8_1k_f32", "config2_puzzle8_65k_fp16", "config2_puzzle8_65k_f32", "config5_az_4096x100", "config5_az_4096x1000",
                        "config5_az_512x1000_reference_default", "evaluations_100_episodes"}
    ev = out.pop("evaluations_100_episodes")
    assert set(ev) == {"ppo_deterministic", "ppo_1", "ppo_10", "mcts_100"} and all({"ms", "success_rate", "mean_reward"} <= set(v) and v["ms"] >= 0 for v in ev.values())
    for k, v in out.items():
        assert {"value", "unit", "ms_per_step", "kernel_ms", "records", "roofline"} <= set(v), k
        assert {"bound", "achieved", "peak", "unit", "frac"} <= set(v["roofline"]) and 0 < v["roofline"]["frac"]
        if k.startswith("config5"):
            assert {"forward_evals", "reused_evals", "episodes", "searches", "launch", "collects", "per_collect"} <= set(v)
            # the three collects together: evaluations of all three over the time of all three
            assert v["collects"] == 3 and v["forward_evals"] == 3 * 40 * v["episodes"] and v["records"] == 3 * 10 * v["episodes"]
            assert abs(v["kernel_ms"] - 1.0) < 1e-9 and len(v["per_collect"]) == 3
            # roofline.frac is the EXECUTED fraction (outputs that came out of a forward); the reference-equivalent rate is a separate key
            assert v["roofline"]["frac"] < v["roofline"]["reference_equivalent_frac"]
            assert abs(v["roofline"]["frac"] / v["roofline"]["reference_equivalent_frac"] - 0.75) < 1e-9       # (40 consumed, 10 of them reused)
    json.dumps(out)


def test_headline_line_for_every_documented_choice():
    """bench.headline (rank 0's contract line) for both puzzles, every precision, one rank and several: JSON-serialisable, the
    contract's keys present, `traffic` null with its source null where no PMC profile exists (Puzzle-8: ADVICE r03, the line used
    to die on an un-packable None after all the timed work)."""
    import types
    sys.path.insert(0, ROOT)
    import bench
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}
    for puzzle in (8, 15):
        for prec in ("fp32", "fp16", "fp16x2"):
            for world in (1, 8):
                args = bench.parse_args(["--puzzle", str(puzzle), "--precision", prec, "--gpus", str(world), "--steps", "4"])
                gi = None if world == 1 else {"pipeline_steps": 5, "episodes_per_rank_and_step": 63488, "reserved_cus": 8, "transport": "x"}
                out = bench.headline(args, world, world > 1, 4.0e6 * world, 0.5, 4.0e6, [100.0, 102.0, 98.0, 100.0], 262_144 * world, gi)
                json.dumps(out)
                assert need <= set(out) and out["n_gpus"] == world and out["vs_baseline"] is None and out["higher_is_better"] is True
                assert out["value"] == 4.0e6 * world / 0.5 and abs(out["ms_per_step"] - 125.0) < 1e-9
                rf = out["roofline"]
                assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf) and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
                assert abs(rf["achieved"] - 1.0e6 * bench.FLOP_PER_RECORD[16 if puzzle == 15 else 9] / 0.1 / 1e12) < 1e-9
                if puzzle == 8:
                    assert rf["traffic"] is None and rf["traffic_source"] is None and rf["traffic_from_profile"] is False
                else:
                    assert (rf["traffic"] is None) == (rf["traffic_source"] is None)
                assert "workload" in out["config"] and "model" not in out["config"] and out["config"]["gather"] == gi


def test_traffic_profile_is_not_older_than_the_newest_bench_round():
    """`roofline.traffic` is bytes per record of the newest committed PMC passes (profiles/rNN_hbm_traffic.json) x the run's records.
    The kernels change between rounds: a traffic file from before the newest BENCH_rNN.json round describes another kernel (VERDICT r03
    weak 5).  Refresh: scripts/profile_r04.sh (or scripts/pmc_traffic.sh) + scripts/summarize_traffic.py."""
    import glob
    import re
    rounds = lambda pat, rx: sorted(int(re.search(rx, os.path.basename(p)).group(1)) for p in glob.glob(os.path.join(ROOT, pat)))
    bench_rounds = rounds("BENCH_r[0-9][0-9].json", r"BENCH_r(\d+)")
    traffic_rounds = rounds(os.path.join("profiles", "r[0-9][0-9]_hbm_traffic.json"), r"r(\d+)_hbm")
    assert traffic_rounds, "no profiles/rNN_hbm_traffic.json"
    if bench_rounds:
        assert traffic_rounds[-1] >= bench_rounds[-1], f"profiles/r{traffic_rounds[-1]:02d}_hbm_traffic.json is older than BENCH_r{bench_rounds[-1]:02d}.json"
    sys.path.insert(0, ROOT)
    import bench
    newest = f"profiles/r{traffic_rounds[-1]:02d}_hbm_traffic.json"
    for kernel in ("rollout_f32", "rollout_f16", "rollout_f16x2"):
        tpr, src = bench.measured_traffic_per_record(kernel)
        assert src == newest and 58.0 <= tpr <= 200.0, (kernel, tpr, src)      # (58 B per record is the algorithmic minimum)
    d = json.load(open(os.path.join(ROOT, newest)))
    fin = d["tw::finalize_ppo_kernel"]["bytes_per_record"]
    assert 98.0 <= fin <= 120.0, fin                                            # each padded record is read once (round 4; 188 before)
