#!/usr/bin/env python3
"""Summary of the walker self-play kernel's SQ counters (scripts/profile_rNN.sh: gpurun_out/rNN/az_pmc_{100,1000}) beside the
bench line's side entries of the same build:  python scripts/summarize_az_pmc.py gpurun_out/r04 > profiles/r04_az_pmc_walker_kernel.txt"""
import collections, csv, json, sys
root = sys.argv[1]
bench = json.loads(open(root + "/bench.json").read().strip().splitlines()[-1])
print("# rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -- python scripts/bench_az.py --envs 4096 --searches S --steps 1")
print("# (scripts/profile_r04.sh; tw::mcts_deep_kernel on the 16-column engine, the two launches of the run (warm-up + step) summed;")
print("#  round 4: the decoupled shape inside one workgroup -- four engine-only waves + the walkers; a forward runs only when a walker waits for one -- pinned with")
print("#  --variant 1024: counter collection serialises kernels, and the SPLIT shape the product takes at this size (walker kernel + engine kernel) needs both at once.")
print("#  round 3, for comparison (profiles/r03_az_pmc_walker_kernel.txt): x 100: matrix pipe busy 0.107, useful fraction 0.20, SQ_WAIT_ANY / SQ_WAVE_CYCLES 0.67;")
print("#  x 1,000: busy 0.061, useful fraction 0.12, wait 0.45)")
for S, shape in ((100, "four engine-only waves + eight walkers x 2 columns"), (1000, "four engine-only waves + four walkers x 4 columns")):
    tot = collections.defaultdict(float); launches = collections.Counter()
    for r in csv.DictReader(open(f"{root}/az_pmc_{S}/p_counter_collection.csv")):
        if "mcts_deep" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); launches[r["Counter_Name"]] += 1
    n = max(launches.values())
    side = bench[f"config5_az_4096x{S}"]       # (the bench entry is the split shape's: its consumed / reused counts are the same searches)
    print(f"== 4,096 x {S} ({shape}): " + ", ".join(f"{k} {int(v)}" for k, v in sorted(tot.items())))
    busy = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * tot["GRBM_GUI_ACTIVE"] / 8)
    print(f"   matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1,024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = {busy:.3f}")
    fw = tot["SQ_INSTS_MFMA"] / n / 1024
    print(f"   forwards per launch = SQ_INSTS_MFMA / {n} launches / 1,024 MFMAs per 16-column forward = {fw:.0f} -> {fw * 16:.3g} column evaluations issued")
    c = side["collects"]
    cons, reu = side["forward_evals"] / c, side["reused_evals"] / c
    print(f"   outputs the searches consume per launch {cons:.0f} (bench.py side entry, {side['kernel_ms']:.1f} ms): {reu:.0f} from the table, {cons - reu:.0f} columns of a forward")
    print(f"   useful fraction of the issued columns = {(cons - reu) / (fw * 16):.2f}")
    print(f"   SQ_WAIT_ANY / SQ_WAVE_CYCLES = {tot['SQ_WAIT_ANY'] / tot['SQ_WAVE_CYCLES']:.2f}")
