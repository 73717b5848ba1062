#!/usr/bin/env python3
"""HBM traffic per record of the rollout / finalize kernels from the rocprofv3 PMC passes of scripts/pmc_traffic.sh:

    python scripts/summarize_traffic.py gpurun_out/traffic profiles/r04_hbm_traffic.json

FETCH_SIZE and WRITE_SIZE come from SEPARATE passes (they do not fit one), in KB as rocprofv3 reports them, averaged per dispatch of
the kernel; bytes = FETCH x 2 (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads) +
WRITE.  Records per launch: the bench line each pass printed (the same seed in every pass)."""
import collections, csv, glob, json, os, sys

root, out_path = sys.argv[1], sys.argv[2]
KERNELS = {   # key -> (precision directory, substring of the kernel name)
    "tw::rollout_f32_kernel (persistent lanes) rollout_f32": ("fp32", "rollout_f32_kernel<8, 16, 0, 8, true>"),
    "tw::finalize_ppo_kernel": ("fp32", "finalize_ppo_"),          # (the grouped form or the wave-per-episode one)
    "tw::rollout_f16_kernel<EngineS> rollout_f16x2": ("fp16x2", "EngineS<8, 16>, 16, true>"),
    "tw::rollout_f16_kernel<Engine16> rollout_f16": ("fp16", "Engine16<8, 16>, 16, true>"),
}


def counter(prec, name, sub):
    vals, kname = [], None
    for f in glob.glob(os.path.join(root, f"{prec}_{name}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and sub in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"])); kname = r["Kernel_Name"]
    # the bench line's side entries launch the same kernels on small batches: only the dispatches of the headline's size count
    big = max(vals) if vals else 0.0
    return [v for v in vals if v >= 0.5 * big], kname


def records(prec):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            for ln in open(os.path.join(root, f"{prec}_{c}.log")):
                if ln.startswith("{") and '"metric"' in ln:
                    return float(json.loads(ln)["config"]["records_per_step"])
        except OSError:
            pass
    return None


res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; scripts/pmc_traffic.sh), python3 bench.py --precision P --steps 1 --warmup 0 "
               "--no-cpu-baseline; units: KB as reported by rocprofv3, averaged per dispatch of the kernel; bytes = FETCH x2 (gfx950 note) + WRITE; "
               "written by scripts/summarize_traffic.py"}
for key, (prec, sub) in KERNELS.items():
    f, kname = counter(prec, "FETCH_SIZE", sub)
    w, _ = counter(prec, "WRITE_SIZE", sub)
    n = records(prec)
    if not f or not w or not n:
        continue
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    b = 2.0 * fk * 1024.0 + wk * 1024.0
    # (the finalize kernel runs once per collect like the rollout: `dispatches_averaged` counts the passes' collects)
    res[key] = {"kernel": kname, "dispatches_averaged": len(f), "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "records_per_launch": n,
                "hbm_bytes_per_launch_fetch_x2_plus_write": b, "bytes_per_record": b / n}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: round(v["bytes_per_record"], 1) for k, v in res.items() if isinstance(v, dict)}))
