"""CPU: the product library loads without a GPU and exports every symbol include/ksw2_amd.h declares."""
import ctypes
import os
import re
import subprocess

import ksw2_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ksw2_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(ksw2amd_\w+|ksw_\w+)\s*\(", src))
    return sorted(n for n in names if n not in ("ksw2amd_plan_s", "ksw2amd_error_fn"))


def test_header_symbols_exported():
    if not os.path.exists(ksw2_amd.DEFAULT_SO):
        subprocess.run(["make", "-C", os.path.join(ROOT, "ksw2_amd", "csrc")], check=True, capture_output=True)
    lib = ctypes.CDLL(ksw2_amd.DEFAULT_SO)
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(ksw2_amd.EXPORTS)


def test_no_oracle_in_product():
    """The product never links, imports or executes the oracle (or the compiled reference)."""
    out = subprocess.run(["ldd", ksw2_amd.DEFAULT_SO], capture_output=True, text=True).stdout
    assert "oracle" not in out and "ksw2ref" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ksw2_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "kso_" not in txt and "libksw2_oracle" not in txt, f


def test_struct_layout_matches_reference():
    assert ctypes.sizeof(ksw2_amd.KswExtz) == 56 and ksw2_amd.KswExtz.cigar.offset == 48
    assert ctypes.sizeof(ksw2_amd.Pair) == 40


def test_fails_loudly_without_gpu():
    import pytest
    lib = ksw2_amd.library()
    if lib.device_count() > 0:
        pytest.skip("GPU present")
    import numpy as np
    one = np.array([1], dtype=np.uint8)
    with pytest.raises(ksw2_amd.Ksw2Error):
        lib.extz_batch([one], [one], np.ones(25, np.int8), 4, 2)
