"""The hand-over for the un-vendored scan (reference src/hilbert.rs:40-43 -> zhang_hilbert 0.1.1, Cargo.toml:15).
tests/golden/zhang/<w>x<h>.xy, when somebody with `cargo` has produced them, hold the crate's order:
w*h little-endian (x, y) u32 pairs.  CPU: each file is a scan (every pixel once) and the oracle's delta / hilbert(rle) round-trip
under it.  GPU: injected through cniic_ctx_set_scan, the HIP streams equal the oracle's under the same order and decode undoes them.
No files (the state of this repository: the crate cannot be obtained here): every test skips, and says why."""
import glob
import hashlib
import os
import re

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "zhang", "*.xy")))
NEED = "no tests/golden/zhang/*.xy: the crate's order has not been dumped (INTEGRATION.md section 7 holds the program)"


def load(path):
    m = re.match(r"(\d+)x(\d+)\.xy$", os.path.basename(path))
    assert m, "name the file <w>x<h>.xy"
    w, h = int(m.group(1)), int(m.group(2))
    xy = np.fromfile(path, dtype="<u4")
    assert xy.size == 2 * w * h, "%s: %d words for %d x %d pixels" % (path, xy.size, w, h)
    return w, h, xy.reshape(-1, 2).astype(np.uint32)


def rearranged(img, s, w, h):
    """the image whose pixels the ORACLE's own scan meets in s's order of img's: same linear sequence, same dimensions"""
    own = O.hilbert_iter(w, h)
    perm = np.empty_like(img)
    perm[own[:, 1], own[:, 0]] = img[s[:, 1], s[:, 0]]
    return perm


@pytest.mark.skipif(not FILES, reason=NEED)
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_dumped_order_is_a_scan_and_the_oracle_round_trips_under_it(path):
    from cniic_amd import synth
    w, h, s = load(path)
    assert s[:, 0].max() < w and s[:, 1].max() < h
    assert np.unique(s[:, 1].astype(np.uint64) * w + s[:, 0]).size == w * h, "the order must visit every pixel exactly once"
    step = np.abs(np.diff(s.astype(np.int64), axis=0)).sum(1)
    print("%s: %.2f%% of the steps go to a 4-neighbour (a Hilbert-like scan: nearly all)" % (os.path.basename(path), 100.0 * (step == 1).mean()))
    img = synth.photo(w, h, synth.SEED0 + 700)
    perm = rearranged(img, s, w, h)
    for expr in ("delta", "hilbert(rle)"):
        rc, data, _ = O.encode(expr, perm)
        rcd, back = O.decode(expr, data)
        assert rc == rcd == 0 and np.array_equal(back, perm)
        print("%s %s under the crate's order: %d bytes, sha256 %s" % (os.path.basename(path), expr, len(data), hashlib.sha256(data).hexdigest()))


@pytest.mark.gpu
@pytest.mark.skipif(not FILES, reason=NEED)
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_hip_streams_under_the_dumped_order_equal_the_oracles(path):
    import cniic_amd
    from cniic_amd import synth
    w, h, s = load(path)
    img = synth.photo(w, h, synth.SEED0 + 700)
    perm = rearranged(img, s, w, h)
    with cniic_amd.Context(0) as ctx:
        ctx.set_scan(w, h, s)
        assert np.array_equal(ctx.hilbert_xy(w, h), s)
        for expr in ("delta", "hilbert(rle)"):
            rco, exp, _ = O.encode(expr, perm)
            rc, data, _ = ctx.encode(expr, img)
            assert rc == rco == 0 and data == exp, expr
            rc, back = ctx.decode(expr, data)
            assert rc == 0 and np.array_equal(back, img), expr


def test_the_hand_over_is_in_place():
    """what a maintainer with `cargo` needs is committed: the dump program and the place for its output"""
    text = open(os.path.join(os.path.dirname(HERE), "INTEGRATION.md")).read()
    assert "ArbHilbertScan32::new" in text and "tests/golden/zhang" in text and "cniic_ctx_set_scan" in text
    assert os.path.isdir(os.path.join(HERE, "golden", "zhang"))
