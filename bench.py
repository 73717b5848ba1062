#!/usr/bin/env python3
"""bench.py -- env-steps/s of the Puzzle-15 PPO rollout on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full `PPOCollector.collect()` over the workload: reset+scramble, the fused
step/observe/reward/mask + policy forward + Gumbel sampling rollout, the GAE pass, compaction into
the reference merge order, and (N > 1) the RCCL gather of the finished trajectories to rank 0.
1 env-step = 1 trajectory record.  Weak scaling: every GPU gets --envs episodes.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel (rollout) against the f32 MFMA peak, measured with HIP events
                  around the kernel on its own stream (tw_collect_stats.ms_rollout)
  "cpu_baseline": the CPU oracle (reference algorithm restated in C, rayon pool -> pthreads)
                  timed on this box's host cores over a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

FLOP_PER_RECORD = {16: 272_896, 9: 269_312}
BYTES_PER_RECORD = {16: 58, 9: 51}
PEAK_TFLOPS = {"fp32": 157.3, "fp16": 2500.0, "fp16x2": 2500.0}     # MI355X_MICROARCH.md: f32 MFMA (=vector) / dense f16 MFMA


def measured_traffic_per_record(kernel="rollout_f32"):
    """HBM bytes per record of the rollout kernel from the committed rocprofv3 PMC passes (the latest round's
    profiles/rNN_hbm_traffic.json: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes)."""
    import glob
    # the NEWEST round's file (tests/test_bench_cli.py fails when it is older than the newest committed BENCH round)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")), reverse=True):
        name = os.path.basename(path)
        try:
            d = json.load(open(path))
            for k, v in d.items():
                if isinstance(v, dict) and (k.endswith(kernel) or (kernel == "rollout_f32" and kernel in k)):
                    return float(v["bytes_per_record"]), "profiles/" + name
        except Exception:
            pass
    return None, None


def synthetic_weights(n2: int, seed: int = 0):
    """torch default-init weights of BasicPolicy(obs n2*n2 -> 512 -> 256 -> 4|1) under
    torch.manual_seed(seed), exported in the reference layout (src/twisterl/nn/utils.py:17-79)."""
    import torch
    torch.manual_seed(seed)
    emb = torch.nn.Linear(n2 * n2, 512)
    common = torch.nn.Linear(512, 256)
    action = torch.nn.Linear(256, 4)
    value = torch.nn.Linear(256, 1)
    f = lambda t: np.ascontiguousarray(t.detach().numpy(), dtype=np.float32)
    return (f(emb.weight.T), f(emb.bias),
            [(f(common.weight.T).reshape(-1), f(common.bias), True)],
            [(f(action.weight.T).reshape(-1), f(action.bias), False)],
            [(f(value.weight.T).reshape(-1), f(value.bias), False)])


def transpose_twist(n: int):
    n2 = n * n
    T = [(i % n) * n + (i // n) for i in range(n2)]
    return ([list(range(n2 * n2)), [T[o // n2] * n2 + T[o % n2] for o in range(n2 * n2)]],
            [[0, 1, 2, 3], [1, 0, 3, 2]])


def build_policy(arrs, obs_perms, act_perms):
    from twisterl_amd import twisterl
    emb, eb, common, action, value = arrs
    seq = lambda ls: twisterl.nn.Sequential([twisterl.nn.Linear(w, b, r) for (w, b, r) in ls])
    return twisterl.nn.Policy(twisterl.nn.EmbeddingBag(emb, eb, True, [emb.shape[0]], 0), seq(common), seq(action),
                              seq(value), obs_perms, act_perms)


def cpu_baseline(arrs, obs_perms, act_perms, side, difficulty, target_seconds, threads=16):
    """Times the CPU oracle (kind 'port': the reference binary cannot be built here) on all host
    cores over a bounded sample of the same workload."""
    from oracle import oracle as O
    # the GPU box gives one GPU's job a 16-core share of the host; never oversubscribe it
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, threads))
    pol = O.Policy(*arrs, obs_perms, act_perms)
    env = O.Puzzle(side, side, difficulty, 2, 256)
    run = lambda E, seed: O.ppo_collect(env, pol, E, 0.995, 0.995, seed=seed, arith=O.ARITH_REF, det_log=False,
                                        num_threads=cores, merge_order=True)
    pilot_E = 4 * cores
    t0 = time.perf_counter(); d = run(pilot_E, 1); dt = time.perf_counter() - t0
    rate = len(d.values) / dt
    E = int(max(pilot_E, min(1 << 16, rate * target_seconds / max(1.0, len(d.values) / pilot_E))))
    t0 = time.perf_counter(); d = run(E, 2); dt = time.perf_counter() - t0
    return {"value": len(d.values) / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{E} episodes ({len(d.values)} records) of the same Puzzle workload, reference-order f32 "
                      f"arithmetic, {cores} threads, {dt:.1f} s"}


def side_configs(twisterl, torch):
    """BASELINE.json configs 1, 2 and 5 at full size, N = 1 only: Puzzle-8 PPO rollouts (1,024 envs exact f32 = config 1's batch on
    the GPU; 65,536 envs with the f16-input forward = config 2), AlphaZero self-play on Puzzle-15 at the reference's per-GPU batch
    (4,096 episodes x 100 and x 1,000 searches, src/twisterl/defaults.py:84-91) and at the reference's default (512 x 1,000).
    PPO entries: the median of three collects after one warm-up.  Self-play entries: the three collects TOGETHER (evaluations of
    all three / time of all three, as the headline sums its steps): a self-play collect is as long as its longest episode's chain
    of searches whatever the batch holds, while the evaluations it counts follow the batch's total number of moves -- one seed's
    rate moves by +- 8 %.  Every entry carries its own roofline fraction."""
    out = {}

    def timed(c, env, pol, together=False):
        c.collect(env, pol, seed=1)
        rs = []
        for i in range(3):
            torch.cuda.synchronize(); t1 = time.perf_counter()
            d = c.collect(env, pol, seed=2 + i)
            torch.cuda.synchronize()
            rs.append((time.perf_counter() - t1, len(d), dict(d.stats)))
            del d
        rs.sort(key=lambda r: r[0])
        return rs[1] if not together else rs

    try:
        arrs8 = synthetic_weights(9, seed=0)
        pol8 = build_policy(arrs8, [], [])
        env8 = twisterl.env.Puzzle(3, 3, 32, 2, 256)                      # examples/ppo_puzzle8_v1.json at diff_max 32: <= 65 records
        for key, E, prec in (("config1_puzzle8_1k_f32", 1024, "fp32"), ("config2_puzzle8_65k_fp16", 65_536, "fp16"), ("config2_puzzle8_65k_f32", 65_536, "fp32")):
            c = twisterl.collector.PPOCollector(**{"num_episodes": E, "gamma": 0.995, "lambda": 0.995, "num_cores": 32}, precision=prec)
            dt1, n, st = timed(c, env8, pol8)
            k = st["ms_rollout"] * 1e-3
            tf = n * FLOP_PER_RECORD[9] / k / 1e12
            out[key] = {"value": n / dt1, "unit": "env-steps/s", "ms_per_step": dt1 * 1e3, "kernel_ms": k * 1e3, "records": n, "envs": E,
                        "dtype": "f16" if prec == "fp16" else "f32",
                        "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS[prec], "unit": "TFLOP/s", "frac": tf / PEAK_TFLOPS[prec]}}
    except Exception as e:           # the headline line must not depend on a side measurement
        out["config2_error"] = str(e)
    try:
        arrs = synthetic_weights(16, seed=0)
        pol = build_policy(arrs, [], [])                                  # the reference's AZ clears the twists (src/twisterl/rl/az.py:23-25)
        env = twisterl.env.Puzzle(4, 4, 8, 2, 256)
        for key, E, S in (("config5_az_4096x100", 4096, 100), ("config5_az_4096x1000", 4096, 1000), ("config5_az_512x1000_reference_default", 512, 1000)):
            c = twisterl.collector.AZCollector(E, S, 1.41, 1, 32)
            rs = timed(c, env, pol, together=True)
            dt1 = sum(r[0] for r in rs); n = sum(r[1] for r in rs)
            k = sum(r[2]["ms_rollout"] for r in rs) * 1e-3
            consumed = sum(r[2]["forward_evals"] for r in rs); reused = sum(r[2]["reused_evals"] for r in rs)
            # EXECUTED useful work: outputs the searches consumed that came out of a forward of this collect.  The outputs served from the
            # board-keyed table (`reused_evals`) were never computed again: they count in the reference-equivalent RATE (what the
            # reference would have evaluated, `value`), not in a roofline fraction.
            fwd_only = consumed - reused
            tf_exec = fwd_only * FLOP_PER_RECORD[16] / k / 1e12
            tf_ref = consumed * FLOP_PER_RECORD[16] / k / 1e12
            st = rs[1][2]
            out[key] = {"value": consumed / dt1, "unit": "leaf evaluations/s (Policy::full_predict calls the searches consume = what the reference evaluates)",
                        "ms_per_step": dt1 * 1e3 / len(rs), "kernel_ms": k * 1e3 / len(rs), "collects": len(rs), "records": n, "episodes": E, "searches": S,
                        "forward_evals": consumed, "reused_evals": reused, "speculative_evals": sum(r[2]["speculative_evals"] for r in rs),
                        "per_collect": [{"ms": round(r[0] * 1e3, 3), "forward_evals": r[2]["forward_evals"]} for r in rs],
                        "launch": [st["rollout_blocks"], st["rollout_threads"]],
                        "roofline": {"bound": "mfma", "achieved": tf_exec, "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s", "frac": tf_exec / PEAK_TFLOPS["fp32"],
                                     "counts": "outputs consumed by the searches that a forward of this collect computed (forward_evals - reused_evals)",
                                     "reference_equivalent_TFLOPs": tf_ref, "reference_equivalent_frac": tf_ref / PEAK_TFLOPS["fp32"]}}
    except Exception as e:
        out["config5_error"] = str(e)
    try:
        # the reference's default evaluations (src/twisterl/defaults.py:27-57: 100 episodes each; `learn_step` runs them beside the collect,
        # algorithm.py:117-121) on the same Puzzle-15 policy: best of three calls after one warm-up, ms per call
        arrs = synthetic_weights(16, seed=0)
        pol = build_policy(arrs, [], [])
        env = twisterl.env.Puzzle(4, 4, 8, 2, 256)
        ev = {}
        for name, kw in (("ppo_deterministic", dict(deterministic=True, num_searches=1, num_mcts_searches=0)), ("ppo_1", dict(deterministic=False, num_searches=1, num_mcts_searches=0)),
                         ("ppo_10", dict(deterministic=False, num_searches=10, num_mcts_searches=0)), ("mcts_100", dict(deterministic=True, num_searches=1, num_mcts_searches=100))):
            args = dict(num_episodes=100, seed=0, C=1.41, max_expand_depth=1, num_cores=32, **kw)
            twisterl.collector.evaluate(env, pol, **args)
            ts = []
            for i in range(3):
                torch.cuda.synchronize(); t1 = time.perf_counter()
                r = twisterl.collector.evaluate(env, pol, **args)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
            ev[name] = {"ms": min(ts) * 1e3, "success_rate": float(r[0]), "mean_reward": float(r[1])}
        out["evaluations_100_episodes"] = ev
    except Exception as e:
        out["evaluations_error"] = str(e)
    return out


def headline(args, world, use_dist, total_records, wall, records, ms_rollout, E_total, gather_info):
    """The contract line of rank 0 (without the side entries): pure arithmetic on what the timed region measured, so that the CPU
    tests can run it for every documented choice of arguments (tests/test_bench_cli.py)."""
    side = 4 if args.puzzle == 15 else 3
    n2 = side * side
    envs_per_gpu = E_total / world
    kern_s = float(np.mean(ms_rollout)) * 1e-3
    rec_per_launch = records / args.steps
    tpr, tpr_file = measured_traffic_per_record({"fp32": "rollout_f32", "fp16": "rollout_f16", "fp16x2": "rollout_f16x2"}[args.precision]) if args.puzzle == 15 else (None, None)
    achieved = rec_per_launch * FLOP_PER_RECORD[n2] / kern_s / 1e12
    peak = PEAK_TFLOPS[args.precision]
    return {
        "metric": "env-steps/s (whole node) Puzzle-15 PPO rollout" if args.puzzle == 15 else "env-steps/s Puzzle-8 PPO rollout",
        "value": total_records / wall,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": {"fp32": "f32", "fp16": "f16", "fp16x2": "f16x2 (f32-equivalent: two f16 terms per operand)"}[args.precision],
        "data": "synthetic",
        "config": {
            "workload": f"Puzzle-{args.puzzle} PPO rollout + GAE + merge: {envs_per_gpu:g} envs/GPU, difficulty {args.difficulty} "
                        f"(<= {2 * args.difficulty + 1} records/episode), twists "
                        f"{'none' if args.no_twists else '{identity, transpose}'}, BasicPolicy {n2 * n2}->512->256->4|1, "
                        f"gamma=lambda=0.995, torch-default-init weights seed 0",
            "envs_per_gpu": envs_per_gpu, "total_envs": E_total, "records_per_step": total_records / args.steps,
            "mean_records_per_episode": total_records / args.steps / E_total, "parallelism": f"episodes sharded x{world}",
            "gather": gather_info,
        },
        "roofline": {
            "bound": "mfma", "kernel": "tw::rollout_f32_kernel" if args.precision == "fp32" else "tw::rollout_f16_kernel", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak,
            # HBM bytes per launch -- NOT a counter read in this run (PMC passes need rocprofv3 around the process): the
            # PMC-measured bytes per record of the committed profile x the records of this launch
            "traffic": (tpr * rec_per_launch) if tpr is not None else None,
            "traffic_from_profile": tpr is not None,
            "traffic_source": None if tpr is None else
                              f"{tpr_file}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel (FETCH x2 per the gfx950 note), "
                              "bytes per record x the records of this run's launch; not measured in this run",
            "kernel_ms": kern_s * 1e3, "flop_per_record": FLOP_PER_RECORD[n2],
            "hbm_bytes_per_record": BYTES_PER_RECORD[n2],
            "hbm_frac": rec_per_launch * BYTES_PER_RECORD[n2] / kern_s / 8e12,
        },
    }


STRONG_TOTAL_ENVS = 2_097_152      # BASELINE.json config 4: "2M envs sharded over 8 x MI355X"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=262_144, help="episodes per GPU per step (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help=f"weak: --envs episodes on every GPU; strong: {STRONG_TOTAL_ENVS} episodes in total, split over the GPUs")
    ap.add_argument("--total-envs", type=int, default=STRONG_TOTAL_ENVS, help="episodes in total with --scaling strong")
    ap.add_argument("--puzzle", type=int, default=15, choices=[8, 15])
    ap.add_argument("--difficulty", type=int, default=128)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "fp16x2"])
    ap.add_argument("--no-twists", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-launch", action="store_true", help="print the launch plan (ranks, command) as JSON and exit; needs no GPU")
    return ap.parse_args(argv)


def launch_plan(args, argv):
    """How `bench.py --gpus N` gets its N ranks when it was NOT started under torchrun: one child process running
    torch.distributed.run with N ranks of this very script (one process per GPU, RCCL)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"]
    return {"n_ranks": args.gpus, "ranks": [{"rank": r, "local_rank": r, "device": f"cuda:{r}"} for r in range(args.gpus)], "command": cmd}


def self_launch(args, argv) -> int:
    """Runs the N-rank job as a fresh CHILD (never exec: nothing here has touched the GPU yet, and nothing will in this
    process) and relays rank 0's JSON line."""
    import subprocess
    plan = launch_plan(args, argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    child = subprocess.run(plan["command"], env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in child.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if child.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank child exited with code {child.returncode}" + ("" if line else " without a result line"), file=sys.stderr)
        return child.returncode or 1
    print(line, flush=True)
    return 0


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    under_launcher = "RANK" in os.environ
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.dry_launch:
        plan = launch_plan(args, argv) if (args.gpus > 1 and not under_launcher) else \
            {"n_ranks": world if under_launcher else 1, "ranks": [{"rank": rank, "local_rank": local_rank, "device": f"cuda:{local_rank}"}], "command": None}
        print(json.dumps(plan), flush=True)
        return
    if under_launcher and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.gpus > 1 and not under_launcher:
        # started plainly (`python bench.py --gpus N`): become the launcher of N ranks BEFORE anything touches torch or HIP
        raise SystemExit(self_launch(args, argv))

    # stdout carries ONE line (rank 0's JSON); libraries that print there (RCCL's version banner) go to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import twisterl_amd
    from twisterl_amd import _lib, twisterl
    if twisterl_amd.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP collector has no CPU fallback")
    # TW_BENCH_REHEARSAL=1 (tests/test_gpu_multirank.py): the N-rank job on ONE GPU -- every rank on cuda:0, gloo for the control
    # collectives, the gather through the library's own exchange over the stand-in transport named by TW_RCCL_LIBRARY (RCCL refuses
    # two ranks on a device).  It runs every line of the N > 1 path on device memory; its rate is not a measurement and the line says so.
    rehearsal = os.environ.get("TW_BENCH_REHEARSAL") == "1"
    if rehearsal and not os.environ.get("TW_RCCL_LIBRARY"):
        raise SystemExit("bench.py: TW_BENCH_REHEARSAL needs TW_RCCL_LIBRARY (the stand-in transport)")
    device_index = 0 if rehearsal else local_rank
    if device_index >= twisterl_amd.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{device_index}, {twisterl_amd.device_count()} device(s) visible")
    torch.cuda.set_device(device_index)
    _lib.check(_lib.lib().tw_set_device(device_index))
    dist = None
    # under torchrun the distributed path runs even at N=1; strong scaling at N=1 needs the chunked path as well
    use_dist = world > 1 or under_launcher or args.scaling == "strong"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not under_launcher:
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        from twisterl_amd.dist import DEFAULT_RESERVE_CUS, Comm, TrajectoryGather, collect_sharded, pipeline_steps

    side = 4 if args.puzzle == 15 else 3
    n2 = side * side
    arrs = synthetic_weights(n2, seed=0)
    obs_perms, act_perms = ([], []) if args.no_twists else transpose_twist(side)
    policy = build_policy(arrs, obs_perms, act_perms)
    env = twisterl.env.Puzzle(side, side, args.difficulty, 2, 256)
    E_total = args.envs * world if args.scaling == "weak" else args.total_envs
    envs_per_gpu = E_total / world
    coll = twisterl.collector.PPOCollector(**{"num_episodes": E_total, "gamma": 0.995, "lambda": 0.995, "num_cores": 32},
                                           precision=args.precision)
    t_max = 2 * args.difficulty + 1                       # records per episode at most (depth_slope 2)

    gatherer, reserve, step_eps, comm = None, 0, None, None
    if use_dist:
        # N > 1: every rank's share is collected in pipeline steps so that a step's xGMI transfer to rank 0 overlaps with the
        # collection of the next step (twisterl_amd.dist).  While a transfer can be in flight the persistent rollout grid
        # leaves `reserve` CUs to RCCL's send/recv kernels; the untrained policy runs every episode to full length, so a step
        # is one whole round of the resident lanes ((CUs - reserve) x 256 episodes per rank) -- anything else would leave most
        # lanes idle for an episode's duration at the end of the step.
        reserve = int(os.environ.get("TW_RESERVE_CUS", str(DEFAULT_RESERVE_CUS if world > 1 else 0)))
        cus = twisterl_amd.device_info()["compute_units"]
        step_eps = int(os.environ.get("TW_STEP_EPISODES", str((cus - reserve) * 256)))
        K = pipeline_steps(E_total, world, 1, step_eps)
        gatherer = TrajectoryGather(dst=0, steps=K, max_records=E_total * t_max if K > 1 else None, max_episode_records=t_max)
        # transport of the gather: torch.distributed's point-to-point ops (default), or TW_GATHER=cabi: RCCL issued by the
        # library itself (tw_gather_*, what a non-Python host uses) -- same steps, same placement, same result
        if os.environ.get("TW_GATHER", "torch") == "cabi" or rehearsal:
            comm = Comm()

    def step(i):
        seed = 1000 + i
        if use_dist:
            merged, datas = collect_sharded(coll, env, policy, seed=seed, dst=0, max_episode_records=t_max, gatherer=gatherer,
                                            reserve_cus=reserve, step_episodes=step_eps, comm=comm)       # the gather is inside the timed region
            n = sum(len(d) for d in datas)
            del merged
            return n, {"ms_rollout": sum(d.stats["ms_rollout"] for d in datas)}
        data = coll.collect(env, policy, seed=seed)
        return len(data), data.stats

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    records, ms_rollout = 0, []
    trace = [] if os.environ.get("TW_BENCH_TRACE") else None     # per-step wall times on stderr (diagnostic)
    for i in range(args.steps):
        ts = time.perf_counter()
        n, st = step(args.warmup + i)
        records += n
        ms_rollout.append(st["ms_rollout"])
        if trace is not None:
            trace.append(round((time.perf_counter() - ts) * 1e3, 1))
    fence()
    dt = time.perf_counter() - t0
    if trace is not None:
        print(f"[bench trace] rank {rank}: ms per step {trace}", file=sys.stderr)

    rec_t = torch.tensor([float(records)], device="cpu" if rehearsal else "cuda")
    dt_t = torch.tensor([dt], device="cpu" if rehearsal else "cuda")
    if dist is not None:
        dist.all_reduce(rec_t, op=dist.ReduceOp.SUM)
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    total_records, wall = float(rec_t.item()), float(dt_t.item())

    if rank == 0:
        gather_info = None if not use_dist else {"pipeline_steps": gatherer.steps, "episodes_per_rank_and_step": step_eps, "reserved_cus": reserve,
                                                 "transport": "RCCL send/recv at final offsets, issued by " + ("the library (tw_gather_*)" if comm is not None else "torch.distributed")}
        out = headline(args, world, use_dist, total_records, wall, records, ms_rollout, E_total, gather_info)
        if rehearsal:
            out["rehearsal"] = f"{world} ranks on ONE GPU over a stand-in transport (TW_BENCH_REHEARSAL): a run of the N > 1 code path, not a measurement"
        if world == 1 and args.precision == "fp32" and not use_dist:
            # side measurements, not the headline: the same workload in the two f16-matrix-core modes.
            #   fp16x2: every f32 operand as two f16 terms -- logits within 5e-8 of the reference f32 arithmetic on sampled
            #           records (tests/tools/acc_modes.py; the exact f32 mode is within 2e-8), tests allow BASELINE.json's 1e-5
            #   fp16:   f16-input forward (BASELINE.json config 2's "MLP policy fp16"), logits within 4e-5
            # env transitions / masks / rewards / sampling are bit-exact in every mode.
            def side_mode(prec):
                try:
                    c = twisterl.collector.PPOCollector(**{"num_episodes": E_total, "gamma": 0.995, "lambda": 0.995, "num_cores": 32},
                                                        precision=prec)
                    c.collect(env, policy, seed=1)
                    ts, ks, n = [], [], 0
                    for i in range(3):          # one collect at a time (the result is released before the next): median of three
                        torch.cuda.synchronize(); t1 = time.perf_counter()
                        d = c.collect(env, policy, seed=2 + i)
                        torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
                        n = len(d); ks.append(d.stats["ms_rollout"] * 1e-3)
                        del d
                    dt1, k = float(np.median(ts)), float(np.median(ks))
                    return {"value": n / dt1, "unit": "env-steps/s", "ms_per_step": dt1 * 1e3, "kernel_ms": k * 1e3,
                            "algorithmic_TFLOPs": n * FLOP_PER_RECORD[n2] / k / 1e12}
                except Exception as e:   # the headline line must not depend on a side measurement
                    return {"error": str(e)}
            out["f16x2_mode_f32_equivalent"] = side_mode("fp16x2")
            out["f16_input_mode"] = side_mode("fp16")
            # BASELINE.json's other single-GPU configurations on the same clock (side entries, outside the headline's timed region)
            out.update(side_configs(twisterl, torch))
        if not args.no_cpu_baseline and world == 1 and not use_dist:     # timed on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(arrs, obs_perms, act_perms, side, args.difficulty, args.cpu_seconds, args.cpu_threads)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
