"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/twisterl_hip.h declares.  No compute entry point is called here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "twisterl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tw_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared_symbols()
    for must in ("tw_ppo_collect", "tw_az_collect", "tw_policy_create", "tw_policy_evaluate", "tw_puzzle_create",
                 "tw_collected_device_ptr", "tw_collected_copy_to_host", "tw_last_error"):
        assert must in names
    assert len(names) >= 40


def test_library_exports_every_declared_symbol():
    from twisterl_amd import _lib
    lib = _lib.lib()                      # builds with hipcc if missing, then dlopen()s
    raw = ctypes.CDLL(_lib.library_path())
    declared = _declared_symbols()
    missing = [n for n in declared if not hasattr(raw, n)]
    assert not missing, missing
    # the ctypes prototype table covers the header exactly
    assert sorted(_lib.SYMBOLS) == declared
    assert lib.tw_abi_version() == _lib.ABI_VERSION == 6


def test_launch_options_are_validated():
    """tw_set_launch_option (diagnostic launch overrides, include/twisterl_hip.h): values outside an option's set are refused with a
    message, accepted ones can be set back to automatic.  Needs no device."""
    from twisterl_amd import _lib
    lib = _lib.lib()
    ok = [(_lib.TW_OPT_FORCE_GEOM, 8), (_lib.TW_OPT_FORCE_GEOM, 32), (_lib.TW_OPT_NO_PERSIST, 1), (_lib.TW_OPT_AZ_VARIANT, 2),
          (_lib.TW_OPT_AZ_VARIANT, 16 + 5), (_lib.TW_OPT_AZ_VARIANT, 32 + 6), (_lib.TW_OPT_AZ_VARIANT, 64), (_lib.TW_OPT_AZ_VARIANT, 64 + 2), (_lib.TW_OPT_AZ_VARIANT, 128 + 16 + 5), (_lib.TW_OPT_AZ_VARIANT, 256), (_lib.TW_OPT_AZ_VARIANT, 512), (_lib.TW_OPT_AZ_VARIANT, 1024), (_lib.TW_OPT_AZ_TREE_BUDGET, 72000), (_lib.TW_OPT_AZ_TREE_BUDGET_MIN, 8000),
          (_lib.TW_OPT_AZ_REUSE, 1), (_lib.TW_OPT_AZ_REUSE, 4)]
    bad = [(_lib.TW_OPT_FORCE_GEOM, 5), (_lib.TW_OPT_AZ_VARIANT, 7), (_lib.TW_OPT_AZ_VARIANT, 48 + 3), (_lib.TW_OPT_AZ_VARIANT, 4096), (_lib.TW_OPT_AZ_VARIANT, 1536), (_lib.TW_OPT_AZ_VARIANT, 384), (_lib.TW_OPT_AZ_VARIANT, -1),
           (_lib.TW_OPT_AZ_TREE_BUDGET, 10), (_lib.TW_OPT_AZ_TREE_BUDGET_MIN, -5), (_lib.TW_OPT_AZ_REUSE, 5), (_lib.TW_OPT_AZ_REUSE, -1), (99, 0)]
    if not os.environ.get("TW_ABLATE"):
        bad.append((_lib.TW_OPT_AZ_REUSE, 2))      # the form that returns different bytes exists in the diagnostic build only
    try:
        for opt, v in ok:
            assert lib.tw_set_launch_option(opt, v) == 0, (opt, v)
        for opt, v in bad:
            assert lib.tw_set_launch_option(opt, v) != 0, (opt, v)
            assert lib.tw_last_error()
    finally:
        for opt in (_lib.TW_OPT_FORCE_GEOM, _lib.TW_OPT_NO_PERSIST, _lib.TW_OPT_AZ_VARIANT, _lib.TW_OPT_AZ_TREE_BUDGET, _lib.TW_OPT_AZ_TREE_BUDGET_MIN,
                    _lib.TW_OPT_AZ_REUSE):
            assert lib.tw_set_launch_option(opt, 0) == 0


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a device every compute entry point must fail with a clear error (this container
    has no GPU; on the GPU box the test is skipped)."""
    import numpy as np
    import twisterl_amd
    from twisterl_amd import twisterl
    if twisterl_amd.device_count() > 0:
        pytest.skip("GPU present")
    from tests.util import amd_policy, make_policy_arrays
    pol = amd_policy(make_policy_arrays(9, emb=32, hidden=32))
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        twisterl.collector.PPOCollector(4, 0.9, 0.9, 1).collect(twisterl.env.Puzzle(3, 3, 2, 2, 256), pol)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        pol.predict(list(range(0, 81, 9)), [True] * 4)
    assert np.isfinite(1.0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under twisterl_amd/ may reference it."""
    pkg = os.path.join(ROOT, "twisterl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "tw_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_compiled_c_host_builds_and_fails_loudly_without_a_gpu():
    """examples/collect_from_c.c links against the C ABI with plain gcc; with no GPU it must stop with an error
    (exit code 2), not fall back to anything."""
    import subprocess
    import twisterl_amd
    from twisterl_amd import build as tb
    exe = tb.build_c_example()
    if twisterl_amd.device_count() > 0:
        return                                  # on the GPU box the parity test runs it for real
    r = subprocess.run([exe, "4", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 2 and "no GPU" in r.stderr and "records" not in r.stdout


def test_stub_transport_exports_what_the_library_resolves():
    """tests/stub_rccl.hip (the stand-in for RCCL behind TW_RCCL_LIBRARY in the multi-rank GPU tests) compiles with hipcc and exports
    every nccl* symbol tw_comm.hip resolves with dlsym -- no more, no fewer."""
    import re
    import subprocess
    from tests.util import build_stub_rccl
    src = open(os.path.join(ROOT, "twisterl_amd", "csrc", "tw_comm.hip")).read()
    wanted = set(re.findall(r'sym\("(nccl\w+)"\)', src))
    assert len(wanted) == 11 and "ncclCommAbort" in wanted
    out = subprocess.run(["nm", "-D", "--defined-only", build_stub_rccl()], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("nccl")}
    assert exported == wanted
    assert "TW_RCCL_LIBRARY" in src
