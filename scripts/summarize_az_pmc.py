#!/usr/bin/env python3
"""Summary of the walker self-play kernel's SQ counters (scripts/profile_r03.sh: gpurun_out/r03/az_pmc_{100,1000}) beside the
bench line's side entries of the same build:  python scripts/summarize_az_pmc.py gpurun_out/r03 > profiles/r03_az_pmc_walker_kernel.txt"""
import collections, csv, json, sys
root = sys.argv[1]
bench = json.loads(open(root + "/bench.json").read().strip().splitlines()[-1])
print("# rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -- python scripts/bench_az.py --envs 4096 --searches S --steps 1")
print("# (scripts/profile_r03.sh on the round's final build; tw::mcts_deep_kernel on the 16-column engine, the two launches of the run (warm-up + step) summed;")
print("#  round 3: board-keyed output table, best-child links, streaks of searches not cut short, longest-looking episodes first)")
for S, shape in ((100, "eight walkers x 2 columns (four of the eight waves only walk)"), (1000, "four walkers x 4 columns")):
    tot = collections.defaultdict(float); launches = collections.Counter()
    for r in csv.DictReader(open(f"{root}/az_pmc_{S}/p_counter_collection.csv")):
        if "mcts_deep" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); launches[r["Counter_Name"]] += 1
    n = max(launches.values())
    side = bench[f"config5_az_4096x{S}"]
    print(f"== 4,096 x {S} ({shape}): " + ", ".join(f"{k} {int(v)}" for k, v in sorted(tot.items())))
    busy = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * tot["GRBM_GUI_ACTIVE"] / 8)
    print(f"   matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1,024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = {busy:.3f}   (round 2: 0.196 at x 100)")
    fw = tot["SQ_INSTS_MFMA"] / n / 1024
    print(f"   forwards per launch = SQ_INSTS_MFMA / {n} launches / 1,024 MFMAs per 16-column forward = {fw:.0f} -> {fw * 16:.3g} column evaluations issued")
    c = side["collects"]
    cons, reu = side["forward_evals"] / c, side["reused_evals"] / c
    print(f"   outputs the searches consume per launch {cons:.0f} (bench.py side entry, {side['kernel_ms']:.1f} ms): {reu:.0f} from the table, {cons - reu:.0f} columns of a forward")
    print(f"   useful fraction of the issued columns = {(cons - reu) / (fw * 16):.2f}")
    print(f"   SQ_WAIT_ANY / SQ_WAVE_CYCLES = {tot['SQ_WAIT_ANY'] / tot['SQ_WAVE_CYCLES']:.2f}")
