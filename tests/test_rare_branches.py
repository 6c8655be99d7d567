"""The rare branches of kmeans::cluster, forced by committed inputs (tests/golden/reseed_golden.npz, made by
tests/golden/make_reseed_golden.py): update_centroids' empty-cluster re-seed (src/kmeans.rs:117-134) and
check_enough_active_clusters failing (src/kmeans.rs:41-57).  CPU: the oracle reproduces the fixture.
GPU: every HIP route (update folded into the assign launch, separate update kernel, brute force, the
pixel-partition route of the codec, the 5-D kernels) gives the fixture's centroids, labels, iteration
count and re-seed count bit for bit."""
import os

import numpy as np
import pytest

import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reseed_golden.npz"))
RGBW = ["rgbw38", "rgbw621", "rgbw1268", "rgbw2355"]
XY = ["xy10", "xy31", "xy43", "xy4"]


def pts_of_keys(keys):
    return np.stack([(keys >> 16) & 255, (keys >> 8) & 255, keys & 255], axis=1).astype(np.int32)


def xy_pts(img):
    h, w = img.shape[:2]
    y, x = np.mgrid[0:h, 0:w]
    return np.concatenate([x.reshape(-1, 1), y.reshape(-1, 1), img.reshape(-1, 3)], axis=1).astype(np.int32)


def image_of(keys, weight):
    """an image whose distinct colours are `keys` with pixel counts `weight` (one row)"""
    k = np.repeat(keys, weight.astype(np.int64))
    np.random.default_rng(1).shuffle(k)
    return np.stack([(k >> 16) & 255, (k >> 8) & 255, k & 255], 1).astype(np.uint8).reshape(1, -1, 3)


# ------------------------------------------------------------------ CPU: the oracle against the fixture
@pytest.mark.parametrize("name", RGBW)
def test_oracle_reproduces_rgbw_reseed_fixture(name):
    K = int(G[name + "_K"][0])
    rc, r = O.kmeans(O.PT_RGBW, O.MODE_L, pts_of_keys(G[name + "_keys"]), G[name + "_weight"], K)
    it, res, _ = (int(v) for v in G[name + "_stats"])
    assert rc == 0 and res >= 2 and r["stats"]["empty_reseeds"] == res and r["stats"]["iterations"] == it
    assert np.array_equal(r["centroids"], G[name + "_centroids"]) and np.array_equal(r["labels"], G[name + "_labels"])


@pytest.mark.parametrize("name", XY)
def test_oracle_reproduces_xy_reseed_fixture(name):
    K = int(G[name + "_K"][0])
    rc, r = O.kmeans(O.PT_XYRGB, O.MODE_L, xy_pts(G[name + "_img"]), None, K)
    it, res, _ = (int(v) for v in G[name + "_stats"])
    assert rc == 0 and res >= 1 and r["stats"]["empty_reseeds"] == res and r["stats"]["iterations"] == it
    assert np.array_equal(r["centroids"], G[name + "_centroids"]) and np.array_equal(r["labels"], G[name + "_labels"])


def test_oracle_few_active_fixture():
    rc, r = O.kmeans(O.PT_RGBW, O.MODE_L, pts_of_keys(G["rgbw_few_keys"]), G["rgbw_few_weight"], int(G["rgbw_few_K"][0]),
                     max_iters=int(G["rgbw_few_max_iters"][0]))
    assert rc == O.FEW_ACTIVE and np.array_equal(r["members"], G["rgbw_few_members"])
    K = int(G["rgbw_few_K"][0])
    assert int((r["members"] > 0).sum()) < min(G["rgbw_few_keys"].size, int(0.99 * K))      # kmeans.rs:41-57
    rc, r = O.kmeans(O.PT_XYRGB, O.MODE_L, xy_pts(G["xy_few_img"]), None, int(G["xy_few_K"][0]), max_iters=int(G["xy_few_max_iters"][0]))
    assert rc == O.FEW_ACTIVE and np.array_equal(r["members"], G["xy_few_members"])


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ctx():
    from cniic_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["persistent", "persistent_no_skip", "fused", "unfused", "brute", "no_skip"])
@pytest.mark.parametrize("name", RGBW)
def test_hip_rgbw_empty_cluster_reseed(ctx, monkeypatch, name, route):
    """kmeans.rs:117-134 through every ColorCount route: the persistent launch re-seeds inside its update phase, the fused prologue
    with iteration = launch - 1"""
    from cniic_amd import _lib
    if route.startswith("persistent"):
        monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")   # (a silent hand-over to the launches would be an error)
    if route.endswith("unfused"):
        monkeypatch.setenv("CNIIC_KM_UNFUSED", "1")
    flags = {"brute": _lib.KM_BRUTE_FORCE, "no_skip": _lib.KM_NO_SKIP, "persistent_no_skip": _lib.KM_NO_SKIP}.get(route, 0)
    ctx.set_opt(_lib.OPT_KM_LOOP, None if route.startswith("persistent") else 1)
    try:
        _reseed_run(ctx, name, flags)
    finally:
        ctx.set_opt(_lib.OPT_KM_LOOP, None)


def _reseed_run(ctx, name, flags):
    K = int(G[name + "_K"][0])
    rc, r = ctx.kmeans_rgbw(G[name + "_keys"], G[name + "_weight"], K, flags=flags)
    it, res, moved = (int(v) for v in G[name + "_stats"])
    assert rc == 0
    assert r["stats"]["empty_reseeds"] == res and res >= 2
    assert r["stats"]["iterations"] == it and r["stats"]["moved_last"] == moved == 0
    assert np.array_equal(r["centroids"].astype(np.int32), G[name + "_centroids"])
    assert np.array_equal(r["labels"], G[name + "_labels"]) and np.array_equal(r["members"], G[name + "_members"])


@pytest.mark.gpu
@pytest.mark.parametrize("sp_min", ["0", str(1 << 40)])
@pytest.mark.parametrize("name", RGBW)
def test_hip_cluster_colors_codec_with_reseeds(ctx, monkeypatch, name, sp_min):
    """the codec (dense-table route and pixel-partition route: there the re-seed selects from the occupancy index)"""
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", sp_min)
    K = int(G[name + "_K"][0])
    img = image_of(G[name + "_keys"], G[name + "_weight"])
    rc, data, st = ctx.encode("cluster-colors(%d)" % K, img)
    rco, edata, est = O.encode("cluster-colors(%d)" % K, img, mode=O.MODE_L)
    assert rc == rco == 0 and data == edata
    assert st["empty_reseeds"] == est["empty_reseeds"] == int(G[name + "_stats"][1]) and st["iterations"] == est["iterations"]


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 1, 4])
@pytest.mark.parametrize("name", XY)
def test_hip_xyrgb_empty_cluster_reseed(ctx, name, flags):
    K = int(G[name + "_K"][0])
    rc, r = ctx.kmeans_xyrgb(G[name + "_img"], K, flags=flags)
    it, res, _ = (int(v) for v in G[name + "_stats"])
    assert rc == 0 and r["stats"]["empty_reseeds"] == res and r["stats"]["iterations"] == it
    c5 = np.concatenate([r["centroids"]["x"][:, None], r["centroids"]["y"][:, None], r["centroids"]["rgb"]], axis=1)
    assert np.array_equal(c5.astype(np.int32), G[name + "_centroids"])
    assert np.array_equal(r["labels"], G[name + "_labels"]) and np.array_equal(r["members"], G[name + "_members"])


@pytest.mark.gpu
def test_hip_few_active_error(ctx):
    """check_enough_active_clusters (kmeans.rs:41-57) -> CNIIC_ERR_FEW_ACTIVE, from the K-means ABI and from the codecs"""
    from cniic_amd import _lib
    K, mi = int(G["rgbw_few_K"][0]), int(G["rgbw_few_max_iters"][0])
    keys, w = G["rgbw_few_keys"], G["rgbw_few_weight"]
    for flags in (0, _lib.KM_BRUTE_FORCE):
        rc, r = ctx.kmeans_rgbw(keys, w, K, max_iters=mi, flags=flags, allow=(_lib.FEW_ACTIVE,))
        assert rc == _lib.FEW_ACTIVE and np.array_equal(r["members"], G["rgbw_few_members"])
        assert r["stats"]["active"] == int((G["rgbw_few_members"] > 0).sum())
    rc, r = ctx.kmeans_rgbw(keys, w, K)                                  # run to the end: every cluster is populated again
    assert rc == 0 and r["stats"]["active"] >= int(0.99 * K)
    rc, _, _ = ctx.encode("cluster-colors(%d)" % K, image_of(keys, w), max_iters=mi, allow=(_lib.FEW_ACTIVE,))
    assert rc == _lib.FEW_ACTIVE
    K, mi = int(G["xy_few_K"][0]), int(G["xy_few_max_iters"][0])
    rc, r = ctx.kmeans_xyrgb(G["xy_few_img"], K, max_iters=mi, allow=(_lib.FEW_ACTIVE,))
    assert rc == _lib.FEW_ACTIVE and np.array_equal(r["members"], G["xy_few_members"])
    rc, _, _ = ctx.encode("voronoi(%d)" % K, G["xy_few_img"], max_iters=mi, allow=(_lib.FEW_ACTIVE,))
    assert rc == _lib.FEW_ACTIVE
