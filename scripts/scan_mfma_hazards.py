#!/usr/bin/env python3
"""Static check of the compiled kernels for the hazard an inline-asm MFMA is exposed to (tw_engine_generic.hpp, ASM_MFMA): a vector-ALU
instruction that writes a register the MFMA reads (A, B or the accumulator), within the two instructions in front of it and without an
s_nop between them.  (hipcc pads the wait states for its own MFMAs, not for an asm string.)  Usage:
    hipcc ... -S --cuda-device-only -o x.s twisterl_amd/csrc/X.hip ; python scripts/scan_mfma_hazards.py x.s ..."""
import re, sys
reg = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")
def regs(tok):
    out = set()
    for m in reg.finditer(tok):
        if m.group(3) is not None: out.add(int(m.group(3)))
        else: out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
total = 0
for path in sys.argv[1:]:
    kernel = None; window = []; hits = {}
    n_mfma = {}
    for ln in open(path):
        s = ln.strip()
        if s.startswith("_Z") and ":" in s.split()[0]: kernel = s.split(":")[0]; window = []; continue
        if not s or s.startswith(";") or s.startswith("."): continue
        op = s.split()[0]
        if op.startswith("v_mfma"):
            ops = [t.strip() for t in s[len(op):].split(",")]
            src = regs(ops[1]) | regs(ops[2]) | (regs(ops[3]) if len(ops) > 3 else set())
            n_mfma[kernel] = n_mfma.get(kernel, 0) + 1
            nops = 0
            for back, (pop, pdst) in enumerate(reversed(window[-3:])):
                if pop == "s_nop": nops += 1; continue
                if pop.startswith("v_") and not pop.startswith("v_mfma") and pdst & src and nops == 0 and back < 2:
                    hits.setdefault(kernel, []).append((pop, sorted(pdst & src), s[:70]))
        dst = set()
        if op.startswith("v_") and not op.startswith("v_cmp"):
            first = s[len(op):].split(",")[0]
            dst = regs(first)
        window.append((op, dst))
        if len(window) > 8: window.pop(0)
    for k, v in hits.items():
        total += len(v)
        print(path.split("/")[-1], str(k)[:90], ":", len(v), "of", n_mfma.get(k, 0), "MFMAs; e.g.", v[0])
print("MFMAs with a freshly written operand:", total)
