"""The reference takes ANY cluster count and image size (src/codec/clusterc.rs:116-141, 274-297: `\\d+` into a usize; src/kmeans.rs:67-68
only asks for len >= K).  The tuned kernels stop at K = 2048 (cluster-colors) and at K = 2048 / sides of 16384 (voronoi); beyond that
cniic_amd/csrc/k_kmeans_wide.hip runs the same K-means exactly and slowly.  Here: those routes against the oracle -- K = 4096 and 5000,
a 1 x 20000 strip, a 17000-wide sliver -- through the K-means ABI and through the codecs."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from cniic_amd import Context
    c = Context(0)
    yield c
    c.close()


def keys_of(img):
    p = img.reshape(-1, 3).astype(np.uint32)
    return (p[:, 0] << 16) | (p[:, 1] << 8) | p[:, 2]


def pts_of_keys(keys):
    return np.stack([(keys >> 16) & 255, (keys >> 8) & 255, keys & 255], axis=1).astype(np.int32)


def xy_pts(img):
    h, w = img.shape[:2]
    y, x = np.mgrid[0:h, 0:w]
    return np.concatenate([x.reshape(-1, 1), y.reshape(-1, 1), img.reshape(-1, 3)], axis=1).astype(np.int32)


@pytest.mark.parametrize("K", [2049, 4096, 5000])
def test_kmeans_rgbw_beyond_2048_clusters(ctx, K):
    from cniic_amd import synth
    img = synth.uniform(112, 112, synth.SEED0 + K)                       # ~12500 distinct colours
    keys, counts = O.count_freqs(keys_of(img))
    w = counts.astype(np.uint32)
    assert keys.size >= K
    rc, got = ctx.kmeans_rgbw(keys, w, K)
    rco, exp = O.kmeans(O.PT_RGBW, O.MODE_L, pts_of_keys(keys), w, K)
    assert rc == rco == 0 and got["stats"]["iterations"] == exp["stats"]["iterations"]
    assert got["stats"]["empty_reseeds"] == exp["stats"]["empty_reseeds"]
    assert np.array_equal(got["centroids"].astype(np.int32), exp["centroids"])
    assert np.array_equal(got["labels"], exp["labels"]) and np.array_equal(got["members"], exp["members"])


@pytest.mark.parametrize("sp_min", ["0", str(1 << 40)])
def test_cluster_colors_4096_stream_equals_the_oracles(ctx, monkeypatch, sp_min):
    from cniic_amd import synth
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", sp_min)                    # both ways to the point list: the pixel partition, the dense table
    img = synth.uniform(96, 128, synth.SEED0 + 77)
    rc, data, st = ctx.encode("cluster-colors(4096)", img)
    rco, edata, est = O.encode("cluster-colors(4096)", img, mode=O.MODE_L)
    assert rc == rco == 0 and data == edata and st["iterations"] == est["iterations"]
    rc, back = ctx.decode("cluster-colors(4096)", data)
    rcd, eback = O.decode("cluster-colors(4096)", edata)
    assert rc == rcd == 0 and np.array_equal(back, eback)


@pytest.mark.parametrize("w,h,K", [(20000, 1, 16), (1, 20000, 7), (17000, 3, 33), (96, 96, 4100), (64, 80, 2049)])
def test_kmeans_xyrgb_beyond_the_tiled_kernels_limits(ctx, w, h, K):
    from cniic_amd import synth
    img = synth.photo(w, h, synth.SEED0 + w + K)
    rc, r = ctx.kmeans_xyrgb(img, K)
    rco, exp = O.kmeans(O.PT_XYRGB, O.MODE_L, xy_pts(img), None, K)
    assert rc == rco == 0 and r["stats"]["iterations"] == exp["stats"]["iterations"]
    c5 = np.concatenate([r["centroids"]["x"][:, None], r["centroids"]["y"][:, None], r["centroids"]["rgb"]], axis=1)
    assert np.array_equal(c5.astype(np.int32), exp["centroids"])
    assert np.array_equal(r["labels"], exp["labels"]) and np.array_equal(r["members"], exp["members"])


@pytest.mark.parametrize("w,h,K", [(20000, 1, 12), (3, 16500, 20), (80, 64, 2500)])
def test_voronoi_codec_beyond_the_limits(ctx, w, h, K):
    from cniic_amd import synth
    img = synth.photo(w, h, synth.SEED0 + h + K)
    expr = "voronoi(%d)" % K
    rc, data, st = ctx.encode(expr, img)
    rco, edata, est = O.encode(expr, img, mode=O.MODE_L)
    assert rc == rco == 0 and data == edata and len(data) == 16 + 19 * K
    rc, back = ctx.decode(expr, data)
    rcd, eback = O.decode(expr, edata)
    assert rc == rcd == 0 and np.array_equal(back, eback)


def test_what_is_still_refused_says_so(ctx):
    """u16 labels end at 65535 clusters; the refusal is an error code, never a wrong result"""
    from cniic_amd import _lib, synth
    img = synth.uniform(300, 300, synth.SEED0 + 1)
    keys, counts = O.count_freqs(keys_of(img))
    rc, _ = ctx.kmeans_rgbw(keys, counts.astype(np.uint32), 70000, allow=(_lib.UNSUPPORTED, _lib.TOO_FEW_POINTS))
    assert rc in (_lib.UNSUPPORTED, _lib.TOO_FEW_POINTS)
