"""the reference's REAL assign step is not plain Lloyd but a triangle-inequality search over
dynamically truncated neighbour lists (src/kmeans.rs:330-416, 189-248), exact while the lists are full and
heuristic afterwards -- oracle mode R restates it line by line.  The HIP path is exact Lloyd (= oracle mode L,
bit for bit, tests/test_gpu_parity.py).  This file pins the distance between the two on IDENTICAL images:
bytes/px within 1 % and MSE within 2 % of mode R, on the photo-like (P) and the uniform-noise (U) generator
of SURVEY 8(d) at 512 x 512 (configs[0]'s size).  Measured when the test was written: bytes -0.04 % .. 0.00 %,
MSE -0.8 % .. +0.06 %."""
import numpy as np
import pytest

import oracle_lib as O

BYTES_BAND, MSE_BAND = 0.01, 0.02


def images():
    from cniic_amd import synth
    return {"P": synth.photo(512, 512, synth.SEED0 + 2), "U": synth.uniform(512, 512, synth.SEED0 + 2)}


def mode_r(expr, img):
    rc, data, st = O.encode(expr, img, mode=O.MODE_R)
    assert rc == 0
    rc, back = O.decode(expr, data)
    assert rc == 0
    return len(data), O.mse(img, back), st


def check_band(n, mse, n_r, mse_r, what):
    assert abs(n / n_r - 1.0) <= BYTES_BAND, "%s: %d bytes against mode R's %d" % (what, n, n_r)
    assert abs(mse / mse_r - 1.0) <= MSE_BAND, "%s: MSE %.3f against mode R's %.3f" % (what, mse, mse_r)


@pytest.mark.parametrize("kind", ["P", "U"])
def test_oracle_mode_l_within_band_of_mode_r(kind):
    """CPU leg (K = 16 keeps exact Lloyd on the CPU to a second): the two oracle modes against each other"""
    img = images()[kind]
    expr = "cluster-colors(16)"
    n_r, mse_r, _ = mode_r(expr, img)
    rc, data, _ = O.encode(expr, img, mode=O.MODE_L)
    rc2, back = O.decode(expr, data)
    assert rc == rc2 == 0
    check_band(len(data), O.mse(img, back), n_r, mse_r, "%s K=16 mode L" % kind)


@pytest.mark.gpu
@pytest.mark.parametrize("K", [16, 256])
@pytest.mark.parametrize("kind", ["P", "U"])
def test_hip_cluster_colors_within_band_of_reference_assign(kind, K):
    """the HIP encode against the reference's own (heuristic) assign on the same image"""
    from cniic_amd import Context
    img = images()[kind]
    expr = "cluster-colors(%d)" % K
    n_r, mse_r, st_r = mode_r(expr, img)
    with Context(0) as ctx:
        rc, data, st = ctx.encode(expr, img)
        assert rc == 0
        rc, back = ctx.decode(expr, data)
        assert rc == 0 and back.shape == img.shape
        mse = ctx.mse(img, back)
    assert abs(mse - O.mse(img, back)) <= 1e-9 * max(1.0, mse)
    check_band(len(data), mse, n_r, mse_r, "%s K=%d HIP" % (kind, K))
    # the two searches converge after a similar number of iterations (not part of the band, a sanity bound)
    assert 0.5 <= st["iterations"] / max(1, st_r["iterations"]) <= 2.0
    assert np.unique(back.reshape(-1, 3), axis=0).shape[0] <= K


@pytest.mark.gpu
def test_hip_within_band_of_reference_assign_at_1536():
    """the same band on a 1536 x 1536 photo-like image, K = 256 (2.4 Mpixels, ~1.4 M distinct colours: the partition route of the
    encoder, tens of iterations) -- VERDICT r02: the band was only ever checked at 512 x 512.  Mode R takes ~4 s here."""
    from cniic_amd import Context, synth
    img = synth.photo(1536, 1536, synth.SEED0 + 2)
    expr = "cluster-colors(256)"
    n_r, mse_r, st_r = mode_r(expr, img)
    with Context(0) as ctx:
        rc, data, st = ctx.encode(expr, img)
        assert rc == 0
        rc, back = ctx.decode(expr, data)
        assert rc == 0
        mse = ctx.mse(img, back)
    check_band(len(data), mse, n_r, mse_r, "P 1536^2 K=256 HIP")
    assert 0.5 <= st["iterations"] / max(1, st_r["iterations"]) <= 2.0
