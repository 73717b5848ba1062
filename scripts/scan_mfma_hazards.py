#!/usr/bin/env python3
"""Static check of the compiled gfx950 kernels for MFMA data hazards the hardware does not interlock.

hipcc pads the wait states around the MFMAs IT emits; an MFMA inside an `asm volatile` string is opaque to its hazard
recognizer (an inline-asm site once produced a wrong visit count: commit c7dd731).  This scanner re-derives the padding rule from
the assembly alone and is run over EVERY MFMA of every kernel -- the compiler's own (which must pass: that is the negative
control, and it calibrates the table) and the inline-asm ones (which nothing else checks).

Wait states: every instruction between producer and consumer counts 1, `s_nop N` counts N+1, an MFMA counts its passes (it
cannot start before the matrix pipe has finished the producer; measured on the MI355X with scripts/micro/mfma_hazard_probe.hip,
profiles/r04_mfma_hazard_probe.txt -- the same run shows that the f32-input MFMAs ARE interlocked against a VALU read of their
result while v_mfma_f32_32x32x16_f16 is not: exactly 12 states, the compiler's number).  The table (gfx950) was read off
hipcc's own padding in tests/golden/mfma_hazard_probes.hip (tests/test_mfma_hazards.py re-derives it on every run):

  producer                         consumer                                                   states needed
  VALU write of a VGPR/AGPR        MFMA reading it as A, B or C                               2
  MFMA, f32 inputs, P passes       VALU / LDS / VMEM / export reading or writing D            P + 2      (16x16x4: 10, 32x32x2: 18)
  (v_mfma_f32_*_f32: "SGEMM")      MFMA reading D as A or B                                   P + 2
                                   MFMA whose C overlaps D without being the same registers   P          (32x32x2 -> 16)
  MFMA, other inputs, P passes     VALU / LDS / VMEM / export reading or writing D            P + 4      (32x32x16_f16: 12, 16x16x32_f16: 8)
  ("XDL")                          MFMA reading D as A or B                                   P + 4
                                   MFMA whose C overlaps D without being the same registers   P + 2      (32x32x16_f16 -> 10)
  any MFMA                         the next MFMA of the same kind taking D whole as its C      0          (accumulate chain)

Pairs are looked for along every control-flow path (basic blocks and their predecessors), 24 wait states deep.  Usage:
    python scripts/scan_mfma_hazards.py file.s ...        # exit code 1 when anything is found
"""
from __future__ import annotations

import re
import sys
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Set, Tuple

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")
MFMA_SHAPE = re.compile(r"v_(?:mfma|smfmac)_\w+?_(\d+)x(\d+)x(\d+)")
VALU_WRITES_2 = 2
WINDOW = 24          # > the largest requirement (18 + 4)


def regs_of(tok: str) -> Set[Tuple[str, int]]:
    out: Set[Tuple[str, int]] = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def mfma_class(op: str) -> Tuple[int, bool]:
    """(passes, is_xdl) of an MFMA opcode on gfx950.  Unknown shapes are taken as the slowest XDL (conservative)."""
    m = MFMA_SHAPE.match(op)
    if not m:
        return 16, True
    mm, kk = int(m.group(1)), int(m.group(3))
    in_type = op.rsplit("_", 1)[1]
    if "f64" in op:
        return 16, True
    if in_type in ("f32", "xf32"):        # v_mfma_f32_MxNxK_f32: the "SGEMM" class
        if mm == 32:
            return 16, False
        if mm == 16:
            return 8, False
        return 2, False
    # 16-bit / 8-bit inputs: the gfx950 double-rate shapes finish in half the passes of the legacy ones
    if mm == 32:
        return (8 if kk >= 16 else 16), True
    if mm == 16:
        return (4 if kk >= 32 else 8), True
    return 2, True


@dataclass
class Ins:
    op: str
    text: str
    line: int
    states: int = 1                       # wait states this instruction stands for
    dst: Set[Tuple[str, int]] = field(default_factory=set)
    src: Set[Tuple[str, int]] = field(default_factory=set)
    is_mfma: bool = False
    in_asm: bool = False
    # MFMA only
    d_tok: str = ""
    a: Set[Tuple[str, int]] = field(default_factory=set)
    b: Set[Tuple[str, int]] = field(default_factory=set)
    c: Set[Tuple[str, int]] = field(default_factory=set)
    c_tok: str = ""


NO_DST_PREFIX = ("v_cmp", "v_cmpx", "v_nop", "v_readlane", "v_readfirstlane")
STORE_LIKE = ("global_store", "buffer_store", "flat_store", "scratch_store", "ds_write", "ds_store", "global_atomic", "buffer_atomic", "exp")


def parse_instruction(s: str, line: int, in_asm: bool) -> Optional[Ins]:
    s = s.split(";")[0].strip() if not s.lstrip().startswith(";") else ""
    if not s or s.startswith(".") or s.endswith(":"):
        return None
    parts = s.split(None, 1)
    op = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    ins = Ins(op=op, text=s, line=line, in_asm=in_asm)
    if op == "s_nop":
        try:
            ins.states = int(rest.strip(), 0) + 1
        except ValueError:
            ins.states = 1
        return ins
    toks = [t.strip() for t in rest.split(",")] if rest else []
    if op.startswith(("v_mfma", "v_smfmac")):
        ins.is_mfma = True
        ins.d_tok = toks[0]
        ins.dst = regs_of(toks[0])
        ins.a, ins.b = regs_of(toks[1]), regs_of(toks[2])
        if op.startswith("v_smfmac"):                  # D is also the accumulator input
            ins.c, ins.c_tok = set(ins.dst), toks[0]
        elif len(toks) > 3:
            ins.c, ins.c_tok = regs_of(toks[3]), toks[3].split()[0]
        ins.src = ins.a | ins.b | ins.c
        # An MFMA between a producer and its consumer stands for its own passes: it cannot start before the matrix pipe has finished
        # the producer.  Measured (scripts/micro/mfma_hazard_probe.hip, profiles/r04_mfma_hazard_probe.txt): behind
        # v_mfma_f32_32x32x16_f16 the last result register is stale for 11 wait states, for 3 with one independent MFMA of the same
        # kind in between (8 + 3 + the reader's predecessor = 12), never with two.
        ins.states = mfma_class(op)[0]
        return ins
    if op.startswith(STORE_LIKE) or op.startswith(NO_DST_PREFIX) or op.startswith(("s_", "buffer_wbl2", "buffer_inv")):
        ins.src = regs_of(rest)
        return ins
    if toks:
        ins.dst = regs_of(toks[0])
        ins.src = regs_of(",".join(toks[1:]))
        if op.startswith(("v_swap", "v_permlane")):    # both operands are read and written
            both = regs_of(rest)
            ins.dst, ins.src = both, both
    return ins


@dataclass
class Hit:
    kernel: str
    rule: str
    producer: Ins
    consumer: Ins
    have: int
    need: int

    def __str__(self) -> str:
        where = "inline asm" if (self.producer.in_asm or self.consumer.in_asm) else "compiler code"
        return (f"{self.kernel[:80]}: {self.rule}: {self.have} wait state(s), {self.need} needed ({where})\n"
                f"    line {self.producer.line}: {self.producer.text[:110]}\n    line {self.consumer.line}: {self.consumer.text[:110]}")


def pair_need(prod: Ins, cons: Ins) -> Tuple[int, str]:
    """Wait states `cons` needs behind `prod` (0: none)."""
    if prod.is_mfma:
        passes, xdl = mfma_class(prod.op)
        raw = passes + (4 if xdl else 2)
        if cons.is_mfma:
            if (cons.a | cons.b) & prod.dst:
                return raw, "MFMA result read as A/B of an MFMA"
            if (cons.c | cons.dst) & prod.dst:
                # the accumulate chain: C is exactly the producer's D and the instruction is of the same class
                if not (cons.c == prod.dst and mfma_class(cons.op) == (passes, xdl)):
                    return passes + (2 if xdl else 0), "MFMA result overlapping the C/D of an MFMA of another shape or register range"
            return 0, ""
        touched = (cons.src | cons.dst) if cons.op.startswith("v_") else cons.src      # VALU: reads and writes; memory / export: reads
        if touched & prod.dst:
            return raw, "MFMA result touched by a non-MFMA instruction"
        return 0, ""
    if cons.is_mfma and prod.op.startswith("v_") and prod.dst & cons.src:
        return VALU_WRITES_2, "VALU write of an MFMA operand"
    return 0, ""


UNCONDITIONAL = ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64")


def check_kernel(kernel: str, body: List[Ins], labels: Dict[str, int], hits: List[Hit]) -> None:
    """Basic blocks + predecessor edges; from every instruction walk backwards over every path, WINDOW wait states deep."""
    n = len(body)
    starts = {0} | set(labels.values())
    for i, x in enumerate(body):
        if x.op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")) and i + 1 < n:
            starts.add(i + 1)
    order = sorted(x for x in starts if x < n)
    block_of, blocks = {}, []
    for k, st in enumerate(order):
        en = order[k + 1] if k + 1 < len(order) else n
        blocks.append((st, en))
        block_of[st] = k
    preds: List[List[int]] = [[] for _ in blocks]
    for k, (st, en) in enumerate(blocks):
        last = body[en - 1]
        if last.op.startswith(("s_cbranch", "s_branch")):
            tgt = last.text.split()[-1]
            if tgt in labels and labels[tgt] in block_of:
                preds[block_of[labels[tgt]]].append(k)
        if not last.op.startswith(UNCONDITIONAL) and k + 1 < len(blocks):
            preds[k + 1].append(k)
    seen: Set[Tuple[int, int, str]] = set()

    def walk(cons: Ins, k: int, upto: int, have: int, best: Dict[int, int]) -> None:
        st, _ = blocks[k]
        for i in range(upto - 1, st - 1, -1):
            if have >= WINDOW:
                return
            prod = body[i]
            need, rule = pair_need(prod, cons)
            if need > have:
                key = (prod.line, cons.line, rule)
                if key not in seen:
                    seen.add(key)
                    hits.append(Hit(kernel, rule, prod, cons, have, need))
            have += prod.states
        if have >= WINDOW:
            return
        for p in preds[k]:
            if best.get(p, WINDOW) <= have:
                continue
            best[p] = have
            walk(cons, p, blocks[p][1], have, best)

    for k, (st, en) in enumerate(blocks):
        for j in range(st, en):
            cons = body[j]
            if cons.op == "s_nop" or cons.op.startswith("s_"):
                continue
            walk(cons, k, j, 0, {})


def scan_text(text: str) -> Tuple[List[Hit], Dict[str, int]]:
    """Returns (hits, {kernel: number of MFMAs}) for one assembly file."""
    hits: List[Hit] = []
    counts: Dict[str, int] = {}
    kernel, body, labels = None, [], {}
    in_asm = False

    def flush():
        if kernel is None or not body:
            return
        counts[kernel] = sum(1 for x in body if x.is_mfma)
        if counts[kernel]:
            check_kernel(kernel, body, labels, hits)

    for ln_no, ln in enumerate(text.splitlines(), 1):
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = LABEL.match(ln)
        if m and not ln[0].isspace():
            name = m.group(1)
            if name.startswith(".L") or name.startswith("BB"):
                labels[name] = len(body)
            else:
                flush()
                kernel, body, labels = name, [], {}
            continue
        ins = parse_instruction(ln, ln_no, in_asm)
        if ins is not None:
            body.append(ins)
    flush()
    return hits, counts


def scan_file(path: str) -> Tuple[List[Hit], Dict[str, int]]:
    with open(path) as f:
        return scan_text(f.read())


def main(argv: List[str]) -> int:
    total, mfmas = 0, 0
    for p in argv:
        hits, counts = scan_file(p)
        mfmas += sum(counts.values())
        for h in hits:
            print(f"{p.split('/')[-1]}: {h}")
        total += len(hits)
    print(f"MFMA hazards: {total} in {mfmas} MFMAs of {len(argv)} file(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
