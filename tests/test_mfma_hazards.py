"""CPU: the MFMA wait-state check is part of the build, and the checker is itself checked.

hipcc pads the hazards of the MFMAs it emits; the engines also issue MFMAs from `asm volatile` strings (pinned schedules), which
get no padding -- one such site produced a wrong visit count in round 3.  twisterl_amd.build now keeps the device assembly of
every object (lib/asm/*.s, the text the objects were assembled from) and FAILS the build when scripts/scan_mfma_hazards.py finds
a producer/consumer pair with too few wait states between them.  Here:

  * the scanner's table is re-derived from hipcc itself: tests/golden/mfma_hazard_probes.hip is compiled, the compiler's own
    padding must pass (negative control), and with its s_nop lines removed every probe must be flagged with the expected rule
    and number of states (positive control);
  * the committed hand-written snippets tests/golden/mfma_hazard_{positive,negative}.s (inline-asm markers, a loop back-edge,
    an unreachable fall-through, an intervening MFMA standing for its passes -- the rule measured on the MI355X,
    profiles/r04_mfma_hazard_probe.txt);
  * the product: every csrc/*.hip has its assembly, newer than its source and every header, and the scan of all of it is clean.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import scan_mfma_hazards as scan          # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _by_kernel(hits):
    out = {}
    for h in hits:
        out.setdefault(h.kernel, []).append(h)
    return out


def test_committed_positive_control_is_flagged_case_by_case():
    hits, counts = scan.scan_file(os.path.join(GOLD, "mfma_hazard_positive.s"))
    got = {k: [(h.rule, h.have, h.need, h.producer.in_asm or h.consumer.in_asm) for h in v] for k, v in _by_kernel(hits).items()}
    assert got == {
        "valu_write_then_asm_mfma": [("VALU write of an MFMA operand", 0, 2, True)],
        "accvgpr_write_then_mfma": [("VALU write of an MFMA operand", 1, 2, False)],
        "xdl_result_read_by_valu": [("MFMA result touched by a non-MFMA instruction", 10, 12, True)],
        "mfma_result_as_c_of_another_shape": [("MFMA result overlapping the C/D of an MFMA of another shape or register range", 8, 16, False)],
        "mfma_result_as_a_of_the_next": [("MFMA result read as A/B of an MFMA", 9, 10, False)],
        "hazard_across_a_loop_back_edge": [("MFMA result touched by a non-MFMA instruction", 3, 12, False)],
    }
    assert sum(counts.values()) == 9


def test_committed_negative_control_passes():
    hits, counts = scan.scan_file(os.path.join(GOLD, "mfma_hazard_negative.s"))
    assert hits == [] and sum(counts.values()) == 8


def test_table_agrees_with_hipcc_own_padding(tmp_path):
    """hipcc's padding of its OWN MFMAs passes the scanner, and is exactly what the scanner asks for: with the s_nop lines taken
    out every probe kernel is flagged, with the compiler's number of states."""
    from twisterl_amd.build import hipcc
    out = str(tmp_path / "probes.s")
    r = subprocess.run([hipcc(), "-O3", "--offload-arch=gfx950", "-S", "--cuda-device-only", os.path.join(GOLD, "mfma_hazard_probes.hip"), "-o", out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    text = open(out).read()
    hits, counts = scan.scan_text(text)
    assert hits == [], "\n".join(str(h) for h in hits)
    assert sum(counts.values()) == 18 and sum(1 for v in counts.values() if v) == 13
    # what hipcc inserted behind each kind of producer (states = N + 1 of `s_nop N`, summed over consecutive ones)
    stripped = "\n".join(ln for ln in text.splitlines() if not ln.strip().startswith("s_nop"))
    hits, _ = scan.scan_text(stripped)
    need = {}
    for h in hits:
        need.setdefault(h.kernel, set()).add((h.rule.split(":")[0], h.need))
    assert len(need) == 13                                  # every probe kernel is flagged once its nops are gone
    res = "MFMA result touched by a non-MFMA instruction"
    assert (res, 18) in need["k_f32_32x32x2_valu"] and (res, 10) in need["k_f32_16x16x4_valu"] and (res, 12) in need["k_f16_32x32x16_valu"]
    assert (res, 10) in need["k_f32_16x16x4_store"]
    assert ("VALU write of an MFMA operand", 2) in need["k_valu_f32_16x16x4"] and ("VALU write of an MFMA operand", 2) in need["k_valuC_f32_16x16x4"]
    assert ("MFMA result read as A/B of an MFMA", 10) in need["k_f32_16x16x4_to_AB"] and ("MFMA result read as A/B of an MFMA", 18) in need["k_f32_32x32x2_to_AB"]
    ovl = "MFMA result overlapping the C/D of an MFMA of another shape or register range"
    assert (ovl, 16) in need["k_f32_32_to_C16"] and (ovl, 10) in need["k_f16_32_to_C16"]
    # and the requirement is TIGHT: hipcc's padding minus one state is already flagged
    import re
    def one_less(m):
        n = int(m.group(2))
        return m.group(1) + (f"s_nop {n - 1}" if n > 0 else "v_nop")
    tight = re.sub(r"(\t)s_nop (\d+)", one_less, text)
    hits, _ = scan.scan_text(tight)
    flagged = {h.kernel for h in hits}
    assert {"k_f32_32x32x2_valu", "k_f32_16x16x4_valu", "k_f16_32x32x16_valu", "k_f32_16x16x4_to_AB", "k_f32_32_to_C16", "k_f16_32_to_C16"} <= flagged


def test_the_built_library_has_no_mfma_hazard():
    from twisterl_amd import build as tb
    have_asm = all(os.path.exists(os.path.join(tb.ASM_DIR, s.replace(".hip", ".s"))) for s in tb.SOURCES)
    tb.build_library(force=not have_asm)                     # (incremental; itself raises on a hazard)
    newest_header = max(os.path.getmtime(h) for h in tb.HEADERS)
    inline, total = 0, 0
    for s in tb.SOURCES:
        asm = os.path.join(tb.ASM_DIR, s.replace(".hip", ".s"))
        assert os.path.exists(asm), asm
        assert os.path.getmtime(asm) >= max(os.path.getmtime(os.path.join(tb.CSRC, s)), newest_header) - 1.0, f"{asm} is older than its sources"
        hits, counts = scan.scan_file(asm)
        assert hits == [], "\n".join(str(h) for h in hits[:10])
        total += sum(counts.values())
        text = open(asm).read()
        inline += sum(1 for blk in text.split(";;#ASMSTART")[1:] if "v_mfma" in blk.split(";;#ASMEND")[0])
    assert total > 30_000 and inline > 1_000, (total, inline)          # the scan saw the engines, asm sites included
