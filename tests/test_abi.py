"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/cniic_hip.h declares, mirrors the reference's Codec trait surface (names, lossless flags,
--codec= parsing), and FAILS LOUDLY when no GPU is usable (there is no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cniic_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cniic_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cniic_amd import _lib
    L = _lib.lib()
    decl = declared_symbols()
    assert len(decl) >= 35
    missing = [s for s in decl if not hasattr(L, s)]
    assert not missing, missing
    # the Python loader's own list is the same set
    assert sorted(_lib.SYMBOLS) == decl


def test_release_library_has_no_test_hooks():
    """CNIIC_TEST_* / CNIIC_DBG_* and the other route-forcing knobs are compiled into the testing build only.
    The release library exports the same symbols, says it is the release build, and none of those names is in its binary."""
    import ctypes as C
    rel = os.path.join(ROOT, "cniic_amd", "libcniic_hip.so")
    tst = os.path.join(ROOT, "cniic_amd", "libcniic_hip_testing.so")
    assert os.path.exists(rel) and os.path.exists(tst), "build both with make -C cniic_amd/csrc"
    R, T = C.CDLL(rel), C.CDLL(tst)
    assert R.cniic_is_testing_build() == 0 and T.cniic_is_testing_build() == 1
    missing = [s for s in declared_symbols() if not hasattr(R, s)]
    assert not missing, missing
    blob_r, blob_t = open(rel, "rb").read(), open(tst, "rb").read()
    hooks = [b"CNIIC_TEST_", b"CNIIC_KM_PS_REQUIRE", b"CNIIC_KM_UNFUSED", b"CNIIC_KM_BATCH", b"CNIIC_HD_PHASES", b"CNIIC_TRACE_HOST", b"CNIIC_XY_UNFUSED"]
    for h in hooks:
        assert h not in blob_r, "release library still carries %s" % h.decode()
        assert h in blob_t, "testing library lost %s" % h.decode()
    # what the release build does read: the documented fallbacks of the context options
    for name in (b"CNIIC_SP_MIN_PIXELS", b"CNIIC_KERNEL_TIMERS", b"CNIIC_COLLECTIVE_TIMEOUT_MS"):
        assert name in blob_r
    from cniic_amd import _lib
    assert _lib.lib().cniic_is_testing_build() == 1     # the suite itself runs on the testing build (conftest.py)


def test_no_oracle_in_product():
    """the product path must not route through oracle/ (tests-only infrastructure)"""
    pkg = os.path.join(ROOT, "cniic_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in src.lower(), os.path.join(dp, f)


@pytest.mark.parametrize("expr,name,lossless", [
    ("hufman", "Hufman", True), ("HUFMAN", "Hufman", True),                 # hufc.rs:42-59
    ("cluster-colors(256)", "cluster-colors_256", False), ("ccol(16)", "cluster-colors_16", False),
    ("c-colors(3)", "cluster-colors_3", False),                             # clusterc.rs:59-65,116-141
    ("voronoi(2048)", "voronoi_2048", False),                               # clusterc.rs:191-197
    ("delta", "delta", True),                                               # hilbertc.rs:433-439
    ("hilbert(rle)", "hilbert-rle", True), ("Hilbert(rle(0))", "hilbert-rle", True),   # hilbertc.rs:81-97,341-383
    ("hilbert(rle(0.0))", "hilbert-rle", True),
])
def test_codec_names_and_flags(expr, name, lossless):
    from cniic_amd import _lib
    assert _lib.codec_name(expr) == name
    assert _lib.codec_is_lossless(expr) == lossless


@pytest.mark.parametrize("expr", ["", "huffman", "delta2", "Delta", "voronoi()", "cluster-colors(x)", "hilbert-rle", "zip-dict",
                                  "hilbert(zip)", "hilbert(rle(0.5))", "hilbert()", "HILBERT(rle)", "hilbert(rle)x"])
def test_codec_parse_rejects(expr):
    from cniic_amd import _lib
    assert _lib.codec_parse(expr) is None


def test_huf_size_is_host_side_and_matches_oracle():
    import numpy as np

    import oracle_lib as O
    from cniic_amd import _lib
    counts = np.array([5, 1, 1, 2, 9, 30, 2], np.uint64)
    nb = _lib.C.c_uint64(0)
    rc = _lib.lib().cniic_huf_size(_lib.SYM_RGB, counts.ctypes.data_as(_lib.C.c_void_p), _lib.C.c_uint64(counts.size), _lib.C.byref(nb))
    assert rc == 0 and nb.value == O.huf_size(O.SYM_RGB, counts)


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import cniic_amd
    with pytest.raises(cniic_amd.CniicError):
        cniic_amd.Context(0)
