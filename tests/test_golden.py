"""Golden vectors (tests/golden/hotpath_golden.npz, made by tests/golden/make_golden.py):
CPU: the oracle still reproduces them.  GPU: the HIP library reproduces them byte for byte,
independently of the oracle."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.npz"))
IMGS = ["photo_96x64", "uniform_40x33", "photo_64x64"]
EXPRS = ["hufman", "delta", "hilbert(rle)", "cluster-colors(16)", "voronoi(8)"]


def keys_of(img):
    p = img.reshape(-1, 3).astype(np.uint32)
    return (p[:, 0] << 16) | (p[:, 1] << 8) | p[:, 2]


def test_golden_inputs_are_the_documented_generator():
    from cniic_amd import synth
    assert np.array_equal(G["img_photo_96x64"], synth.photo(96, 64, synth.SEED0 + 1))
    assert np.array_equal(G["img_uniform_40x33"], synth.uniform(40, 33, synth.SEED0 + 1))


@pytest.mark.parametrize("name", IMGS)
@pytest.mark.parametrize("expr", EXPRS)
def test_oracle_reproduces_golden(name, expr):
    import oracle_lib as O
    img = G["img_" + name]
    rc, data, st = O.encode(expr, img, mode=O.MODE_L)
    assert rc == 0 and data == G["enc_%s_%s" % (name, expr)].tobytes()
    rc, back = O.decode(expr, data)
    assert rc == 0 and np.array_equal(back, G["dec_%s_%s" % (name, expr)])


@pytest.fixture(scope="module")
def ctx():
    from cniic_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", IMGS)
@pytest.mark.parametrize("expr", EXPRS)
def test_hip_reproduces_golden_streams(ctx, name, expr):
    img = G["img_" + name]
    rc, data, st = ctx.encode(expr, img)
    assert rc == 0 and data == G["enc_%s_%s" % (name, expr)].tobytes()
    if "col" in expr or "voronoi" in expr:
        assert st["iterations"] == int(G["iters_%s_%s" % (name, expr)][0])
    rc, back = ctx.decode(expr, data)
    assert rc == 0 and np.array_equal(back, G["dec_%s_%s" % (name, expr)])


@pytest.mark.gpu
@pytest.mark.parametrize("name", IMGS)
def test_hip_reproduces_golden_histogram_and_delta(ctx, name):
    img = G["img_" + name]
    k, c = ctx.hist_rgb24(img)
    assert np.array_equal(k, G["hist_keys_" + name]) and np.array_equal(c, G["hist_counts_" + name])
    assert np.array_equal(ctx.hilbert_delta(img), G["delta_syms_" + name])


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(16, 16), (13, 8), (5, 31)])
def test_hip_reproduces_golden_hilbert(ctx, w, h):
    assert np.array_equal(ctx.hilbert_xy(w, h), G["hilbert_%dx%d" % (w, h)])


@pytest.mark.gpu
def test_hip_reproduces_golden_kmeans(ctx):
    img = G["img_photo_64x64"]
    k, c = ctx.hist_rgb24(img)
    rc, r = ctx.kmeans_rgbw(k, c.astype(np.uint32), 32)
    assert rc == 0
    assert np.array_equal(r["centroids"], G["km_rgbw_centroids"])
    assert np.array_equal(r["labels"], G["km_rgbw_labels"])
    assert np.array_equal(r["members"], G["km_rgbw_members"])
    assert r["stats"]["iterations"] == int(G["km_rgbw_iters"][0])
