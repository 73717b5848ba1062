"""Builds the HIP collector library for gfx950 with hipcc (in-tree, no JIT cache).

    python -m twisterl_amd.build            # incremental
    python -m twisterl_amd.build --force

Output: twisterl_amd/lib/libtwisterl_hip.so (git-ignored; it travels to the GPU box with the tree).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
# The diagnostic variant (TW_ABLATE=1 in the environment: -DTW_ABLATE, cycle stamps and knock-out switches) is a DIFFERENT
# library in its own directory: the product library is never overwritten by an instrumented build, and a process loads the
# instrumented one only while TW_ABLATE is set.
ABLATE = bool(os.environ.get("TW_ABLATE"))
# A measurement variant (TW_VARIANT=<name> with TW_EXTRA_FLAGS=-D..., e.g. the builtin forms of the inline-asm MFMAs) likewise
# builds into, and loads from, a directory of its own.
VARIANT = os.environ.get("TW_VARIANT", "")
LIB_DIR = os.path.join(PKG, "lib", "ablate") if ABLATE else os.path.join(PKG, "lib", "variants", VARIANT) if VARIANT else os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libtwisterl_hip_ablate.so" if ABLATE else f"libtwisterl_hip_{VARIANT}.so" if VARIANT else "libtwisterl_hip.so")
SOURCES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "*.hip")))
# every header is a dependency of every object (an edited header must never leave a stale object behind)
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))

# -ffp-contract=off: the numeric spec allows only the explicit fma() calls
# -pragma-unroll-threshold: the pinned MFMA schedules are fully unrolled position loops whose bodies call constexpr
# schedule functions; the default limit (16K IR instructions before folding) would leave them as real loops
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-fPIC",
         "-mllvm", "-pragma-unroll-threshold=400000",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


ASM_DIR = os.path.join(LIB_DIR, "asm")          # the device assembly of every object (what scripts/scan_mfma_hazards.py reads)
_TEMP_SUFFIXES = (".bc", ".hipi", ".out", ".out.resolution.txt", ".hipfb", "-host-x86_64-unknown-linux-gnu.s", "-hip-amdgcn-amd-amdhsa-gfx950.o")


def _keep_device_asm(stem: str) -> None:
    """-save-temps=obj leaves every intermediate file beside the object: the gfx950 assembly (the text the object was assembled
    from) moves to lib/asm/, the rest goes."""
    os.makedirs(ASM_DIR, exist_ok=True)
    for f in os.listdir(LIB_DIR):
        if not f.startswith(stem + "-") and not f.startswith(stem + ".hip-"):
            continue
        path = os.path.join(LIB_DIR, f)
        if f == stem + "-hip-amdgcn-amd-amdhsa-gfx950.s":
            os.replace(path, os.path.join(ASM_DIR, stem + ".s"))
        elif f.endswith(_TEMP_SUFFIXES):
            os.remove(path)


def scan_hazards(verbose: bool = False) -> int:
    """MFMA wait-state check of the assembly the library was built from (scripts/scan_mfma_hazards.py): inline-asm MFMAs get no
    padding from hipcc, so a hazard there is a BUILD ERROR, not something a parity test may or may not catch."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import scan_mfma_hazards as scan
    finally:
        sys.path.pop(0)
    files = sorted(glob.glob(os.path.join(ASM_DIR, "*.s")))
    missing = [s for s in SOURCES if not os.path.exists(os.path.join(ASM_DIR, s.replace(".hip", ".s")))]
    if missing:
        raise RuntimeError(f"no device assembly for {missing}: rebuild with --force")
    report, mfmas = [], 0
    for f in files:
        hits, counts = scan.scan_file(f)
        mfmas += sum(counts.values())
        report += [f"{os.path.basename(f)}: {h}" for h in hits]
    if report:
        raise RuntimeError("MFMA hazards in the compiled kernels (scripts/scan_mfma_hazards.py):\n" + "\n".join(report[:40]))
    if verbose:
        print(f"[build] MFMA hazard scan: 0 in {mfmas} MFMAs of {len(files)} files")
    return mfmas


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    flags = list(FLAGS)
    if ABLATE:                           # timing-only ablation variants of the kernels (profiling aid)
        flags.append("-DTW_ABLATE")
    if os.environ.get("TW_EXTRA_FLAGS"):
        flags += os.environ["TW_EXTRA_FLAGS"].split()
    objs = []
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    procs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIB_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc(), *flags, "-save-temps=obj", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name}:\n{out}")
        if verbose and out.strip():
            print(out)
        _keep_device_asm(name.replace(".hip", ""))
    # The scan belongs to the compile: whenever an object was rebuilt, all of the library's assembly is checked.  (A tree that
    # holds objects without their assembly -- the snapshot on the GPU box leaves lib/asm/ behind, .gpurunignore -- is not
    # recompiled for it; tests/test_mfma_hazards.py insists on the assembly where the library is built.)
    if procs:
        scan_hazards(verbose)
    if force or procs or _stale(LIB_PATH, objs):
        cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB_PATH, *objs]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB_PATH


def build_c_example() -> str:
    """examples/collect_from_c.c: a compiled host over the C ABI alone (gcc, links the library built above)."""
    src = os.path.join(ROOT, "examples", "collect_from_c.c")
    exe = os.path.join(ROOT, "examples", "collect_from_c")
    if _stale(exe, [src, LIB_PATH, os.path.join(ROOT, "include", "twisterl_hip.h")]):
        cmd = ["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-L", LIB_DIR, "-ltwisterl_hip",
               "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,$ORIGIN/../twisterl_amd/lib", "-lm", "-o", exe]
        if ABLATE or VARIANT:
            # the example links the PRODUCT library only: an instrumented or variant build neither builds nor uses it
            if not os.path.exists(exe):
                raise RuntimeError("examples/collect_from_c is built against the product library: run `python -m twisterl_amd.build` without TW_ABLATE / TW_VARIANT first")
            return exe
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"gcc failed on collect_from_c.c:\n{r.stdout}")
    return exe


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_c_example())
