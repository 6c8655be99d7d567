"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same inputs.
Bit-exact for every integer / byte / index result."""
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


def synth_img(h, w, seed=0, levels=256, noise=2):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, levels, (h // 4 + 1, w // 4 + 1, 3))
    img = np.kron(base, np.ones((4, 4, 1), np.int64))[:h, :w]
    img = img + rng.integers(-noise, noise + 1, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def keys_of(img):
    p = img.reshape(-1, 3).astype(np.uint32)
    return (p[:, 0] << 16) | (p[:, 1] << 8) | p[:, 2]


def pts_of_keys(keys):
    return np.stack([(keys >> 16) & 255, (keys >> 8) & 255, keys & 255], axis=1).astype(np.int32)


# ------------------------------------------------------------------ synthetic inputs
@pytest.mark.parametrize("kind", [0, 1])
def test_synth_matches_numpy(ctx, kind):
    from cniic_amd import synth
    w, h = 200, 131
    got = ctx.synth_image(kind, 0x636E696963 + 2, w, h)
    exp = (synth.uniform if kind == 0 else synth.photo)(w, h, 0x636E696963 + 2)
    assert np.array_equal(got, exp)


# ------------------------------------------------------------------ H1 count_freqs
@pytest.mark.parametrize("shape", [(1, 1), (1, 15), (3, 17), (64, 64), (97, 131), (256, 256)])
def test_hist_rgb24(ctx, shape):
    img = synth_img(*shape, seed=shape[0])
    keys, counts = ctx.hist_rgb24(img)
    ek, ec = O.count_freqs(keys_of(img))
    assert np.array_equal(keys, ek) and np.array_equal(counts, ec)
    assert counts.sum() == shape[0] * shape[1]


def test_hist_rgb24_single_colour_and_extremes(ctx):
    img = np.zeros((40, 40, 3), np.uint8)
    img[20:] = 255
    keys, counts = ctx.hist_rgb24(img)
    assert keys.tolist() == [0, 0xFFFFFF] and counts.tolist() == [800, 800]


def test_hist_syms_signed(ctx):
    rng = np.random.default_rng(3)
    syms = rng.integers(0, 1 << 27, 5000).astype(np.uint32)
    syms[:2000] = syms[0]
    keys, counts = ctx.hist_syms(2, syms)
    ek, ec = O.count_freqs(syms)
    assert np.array_equal(keys, ek) and np.array_equal(counts, ec)


# ------------------------------------------------------------------ K-means, colour form
@pytest.mark.parametrize("K", [1, 5, 64, 256, 300])
def test_kmeans_step_rgbw(ctx, K):
    rng = np.random.default_rng(K)
    img = synth_img(96, 96, seed=K)
    keys, counts = O.count_freqs(keys_of(img))
    w = counts.astype(np.uint32)
    cent = rng.integers(0, 256, (K, 3)).astype(np.uint8)
    cent[K // 2] = cent[0]                       # duplicate centroid: tie -> lowest id
    labels = rng.integers(0, K, keys.size).astype(np.uint32)
    got = ctx.kmeans_step_rgbw(keys, w, K, cent, labels)
    exp = O.kmeans_step(O.PT_RGBW, pts_of_keys(keys), w, K, cent.astype(np.int32), labels)
    for f in ("labels", "sums", "wsum", "members"):
        assert np.array_equal(got[f], exp[f]), f
    assert got["changed"] == exp["changed"]


def test_kmeans_step_rgbw_ties_stay(ctx):
    """kmeans.rs:375 strict '<': a point equidistant from its own and another centroid stays."""
    keys = np.array([0x000000, 0x020000, 0x010000], np.uint32)   # r = 0, 2, 1
    w = np.ones(3, np.uint32)
    cent = np.array([[0, 0, 0], [2, 0, 0]], np.uint8)
    for lab in (0, 1):
        labels = np.array([0, 1, lab], np.uint32)
        got = ctx.kmeans_step_rgbw(keys, w, 2, cent, labels)
        assert got["labels"].tolist() == [0, 1, lab] and got["changed"] == 0


@pytest.mark.parametrize("loop", ["persistent", "launches", "unfused"])
@pytest.mark.parametrize("K,shape", [(2, (32, 32)), (16, (64, 64)), (256, (128, 128)), (300, (128, 160)), (200, (300, 260))])
def test_kmeans_rgbw_run(ctx, monkeypatch, K, shape, loop):
    """`loop`: the whole run as ONE persistent launch (the default of a one-GPU run with K <= 256, k_kmeans_persist.hip;
    CNIIC_KM_PS_REQUIRE makes a silent hand-over an error) or one launch per iteration (CNIIC_OPT_KM_LOOP = 1: what several
    GPUs, K > 256 and the batch workers run; "unfused": with the centroid update as a kernel of its own, CNIIC_KM_UNFUSED)"""
    from cniic_amd import _lib
    if loop == "persistent" and K <= 256:
        monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")
    if loop == "unfused":
        monkeypatch.setenv("CNIIC_KM_UNFUSED", "1")
    ctx.set_opt(_lib.OPT_KM_LOOP, 1 if loop != "persistent" else None)
    try:
        _kmeans_rgbw_run(ctx, K, shape)
    finally:
        ctx.set_opt(_lib.OPT_KM_LOOP, None)


def _kmeans_rgbw_run(ctx, K, shape):
    img = synth_img(*shape, seed=7 + K)
    keys, counts = O.count_freqs(keys_of(img))
    w = counts.astype(np.uint32)
    rc, got = ctx.kmeans_rgbw(keys, w, K)
    rco, exp = O.kmeans(O.PT_RGBW, O.MODE_L, pts_of_keys(keys), w, K)
    assert rc == rco == 0
    assert got["stats"]["iterations"] == exp["stats"]["iterations"]
    assert np.array_equal(got["centroids"].astype(np.int32), exp["centroids"])
    assert np.array_equal(got["labels"], exp["labels"])
    assert np.array_equal(got["members"], exp["members"])


@pytest.mark.parametrize("K,flags", [(16, 1), (256, 1), (300, 1), (2048, 0)])
def test_kmeans_rgbw_brute_and_wide(ctx, K, flags):
    """brute-force kernel (flag 1) and the cell-pruned kernel give the oracle's run; K > 256 uses u16 labels"""
    img = synth_img(160, 160, seed=3 * K)
    keys, counts = O.count_freqs(keys_of(img))
    w = counts.astype(np.uint32)
    rc, got = ctx.kmeans_rgbw(keys, w, K, flags=flags)
    rco, exp = O.kmeans(O.PT_RGBW, O.MODE_L, pts_of_keys(keys), w, K)
    assert rc == rco == 0
    assert got["stats"]["iterations"] == exp["stats"]["iterations"]
    assert np.array_equal(got["centroids"].astype(np.int32), exp["centroids"])
    assert np.array_equal(got["labels"], exp["labels"])
    assert np.array_equal(got["members"], exp["members"])


def test_kmeans_rgbw_unsorted_input_order(ctx):
    """the point ORDER is the caller's (the reference clusters a HashMap-ordered Vec, clusterc.rs:21-24):
    a shuffled list must give the oracle's result for that same order"""
    img = synth_img(96, 96, seed=77)
    keys, counts = O.count_freqs(keys_of(img))
    perm = np.random.default_rng(5).permutation(keys.size)
    keys, w = keys[perm], counts[perm].astype(np.uint32)
    rc, got = ctx.kmeans_rgbw(keys, w, 32)
    rco, exp = O.kmeans(O.PT_RGBW, O.MODE_L, pts_of_keys(keys), w, 32)
    assert rc == rco == 0
    assert np.array_equal(got["centroids"].astype(np.int32), exp["centroids"])
    assert np.array_equal(got["labels"], exp["labels"])


def test_kmeans_rgbw_errors(ctx):
    from cniic_amd import _lib
    keys = np.arange(5, dtype=np.uint32)
    rc, _ = ctx.kmeans_rgbw(keys, np.ones(5, np.uint32), 8, allow=(_lib.TOO_FEW_POINTS,))
    assert rc == _lib.TOO_FEW_POINTS                         # kmeans.rs:68


def test_remap_rgb(ctx):
    img = synth_img(50, 70, seed=5)
    keys, counts = O.count_freqs(keys_of(img))
    rng = np.random.default_rng(0)
    K = 9
    labels = rng.integers(0, K, keys.size).astype(np.uint32)
    cent = rng.integers(0, 256, (K, 3)).astype(np.uint8)
    out = ctx.remap_rgb(img, keys, labels, cent)
    lut = dict(zip(keys.tolist(), labels.tolist()))
    exp = np.array([cent[lut[k]] for k in keys_of(img).tolist()], np.uint8).reshape(img.shape)
    assert np.array_equal(out, exp)


# ------------------------------------------------------------------ K-means, pixel (x,y,r,g,b) form
def xy_pts(img):
    h, w = img.shape[:2]
    y, x = np.mgrid[0:h, 0:w]
    return np.concatenate([x.reshape(-1, 1), y.reshape(-1, 1), img.reshape(-1, 3)], axis=1).astype(np.int32)


@pytest.mark.parametrize("K,shape", [(1, (8, 8)), (7, (40, 56)), (64, (70, 130)), (300, (64, 64))])
def test_kmeans_step_xyrgb(ctx, K, shape):
    from cniic_amd._lib import COLORPOS
    rng = np.random.default_rng(K)
    img = synth_img(*shape, seed=K)
    h, w = shape
    cent = np.zeros(K, COLORPOS)
    cent["x"] = rng.integers(0, w, K); cent["y"] = rng.integers(0, h, K)
    cent["rgb"] = rng.integers(0, 256, (K, 3))
    labels = rng.integers(0, K, h * w).astype(np.uint32)
    got = ctx.kmeans_step_xyrgb(img, K, cent, labels)
    c5 = np.concatenate([cent["x"][:, None], cent["y"][:, None], cent["rgb"]], axis=1).astype(np.int32)
    exp = O.kmeans_step(O.PT_XYRGB, xy_pts(img), None, K, c5, labels)
    for f in ("labels", "sums", "members"):
        assert np.array_equal(got[f], exp[f]), f
    assert got["changed"] == exp["changed"]


@pytest.mark.parametrize("K,shape,flags", [(4, (24, 32), 0), (16, (48, 80), 0), (16, (48, 80), 1), (50, (100, 70), 0)])
def test_kmeans_xyrgb_run(ctx, K, shape, flags):
    img = synth_img(*shape, seed=K + 1)
    rc, got = ctx.kmeans_xyrgb(img, K, flags=flags)
    rco, exp = O.kmeans(O.PT_XYRGB, O.MODE_L, xy_pts(img), None, K)
    assert rc == rco == 0
    assert got["stats"]["iterations"] == exp["stats"]["iterations"]
    c5 = np.concatenate([got["centroids"]["x"][:, None], got["centroids"]["y"][:, None], got["centroids"]["rgb"]], axis=1)
    assert np.array_equal(c5.astype(np.int32), exp["centroids"])
    assert np.array_equal(got["labels"], exp["labels"])
    assert np.array_equal(got["members"], exp["members"])


# ------------------------------------------------------------------ Hilbert + delta
@pytest.mark.parametrize("w,h", [(1, 1), (1, 9), (9, 1), (4, 4), (16, 16), (5, 3), (13, 8), (31, 10), (64, 48), (100, 37), (128, 128)])
def test_hilbert_xy(ctx, w, h):
    assert np.array_equal(ctx.hilbert_xy(w, h), O.hilbert_iter(w, h))


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (16, 16), (33, 20), (64, 64), (75, 130)])
def test_hilbert_linearize_and_delta(ctx, shape):
    img = synth_img(*shape, seed=11)
    lin = ctx.hilbert_linearize(img)
    elin = O.hilbert_linearize(img)
    assert np.array_equal(lin, elin)
    syms = ctx.hilbert_delta(img)
    esyms = O.delta_diff(elin)
    assert np.array_equal(syms, esyms)
    keys, counts, syms2 = ctx.hilbert_delta_hist(img, want_syms=True)
    ek, ec = O.count_freqs(esyms)
    assert np.array_equal(keys, ek) and np.array_equal(counts, ec) and np.array_equal(syms2, esyms)


@pytest.mark.parametrize("area", [2, 3, 16, 100, 4096])
@pytest.mark.parametrize("w,h", [(1, 9), (9, 1), (1, 5000), (5000, 1), (2, 700), (700, 3), (5, 3), (13, 8), (100, 37), (97, 131), (300, 200), (1000, 7), (255, 257), (640, 480),
                                 (1023, 517), (96, 64), (64, 96)])
def test_scan_leaves_equal_the_recursion(ctx, monkeypatch, w, h, area):
    """the scan of rectangles that are no 2^n square, cut into leaves (ScanLeavesDev: the recursion's upper levels walked once per
    image size, a table of offsets per class of leaf): every position as the oracle's iterator gives it, for leaves from two
    positions (a tree as deep as the recursion itself) to 4096 (the product's), lines and thin strips included; then the codecs
    that follow the scan, both ways"""
    monkeypatch.setenv("CNIIC_SCAN_LEAVES_MIN", "0")
    monkeypatch.setenv("CNIIC_SCAN_LEAF_AREA", str(area))
    assert np.array_equal(ctx.hilbert_xy(w, h), O.hilbert_iter(w, h))
    if area in (3, 4096):
        img = synth_img(h, w, seed=w + h)
        assert np.array_equal(ctx.hilbert_linearize(img), O.hilbert_linearize(img))
        for expr in ("delta", "hilbert(rle)"):
            rc, data, _ = ctx.encode(expr, img)
            assert rc == 0 and data == O.encode(expr, img)[1], expr
            rc, back = ctx.decode(expr, data)
            assert rc == 0 and np.array_equal(back, img), expr


def test_scan_leaves_of_several_sizes_are_kept_and_replaced(ctx, monkeypatch):
    """a context keeps the leaves of four image sizes; a fifth replaces the one used longest ago"""
    monkeypatch.setenv("CNIIC_SCAN_LEAVES_MIN", "0")
    sizes = [(33, 20), (75, 130), (100, 37), (64, 48), (31, 10), (33, 20), (200, 90), (75, 130)]
    for rep in range(2):
        for w, h in sizes:
            assert np.array_equal(ctx.hilbert_xy(w, h), O.hilbert_iter(w, h))


@pytest.mark.parametrize("move", ["", "any"])
@pytest.mark.parametrize("size", [64, 128, 512])
def test_hilbert_move_by_tiles_and_by_positions(ctx, monkeypatch, size, move):
    """2^n squares from 64 x 64: the tile kernel (rows <-> LDS <-> scan positions) and the per-position kernel give the oracle's
    linearised image, and `hilbert(rle)` / `delta` decode (the scatter) gives the image back"""
    if move:
        monkeypatch.setenv("CNIIC_HILBERT_MOVE", move)
    img = np.random.default_rng(size).integers(0, 256, (size, size, 3)).astype(np.uint8)
    assert np.array_equal(ctx.hilbert_linearize(img), O.hilbert_linearize(img))
    flat = img.copy()
    flat[:, : size // 2] = 9  # runs for the RLE
    for expr, im in (("hilbert(rle)", flat), ("delta", img)):
        rc, data, _ = ctx.encode(expr, im)
        assert rc == 0 and data == O.encode(expr, im)[1]
        rc, back = ctx.decode(expr, data)
        assert rc == 0 and np.array_equal(back, im)


@pytest.mark.parametrize("lane_scan", ["0", "1"])
@pytest.mark.parametrize("size", [8, 32, 256])
def test_hilbert_delta_both_kernels_on_pow2_squares(ctx, monkeypatch, lane_scan, size):
    """2^n squares: the one-position-per-lane kernel and the four-positions-per-thread kernel, with and without the
    fused histogram, give the oracle's symbols and counts"""
    monkeypatch.setenv("CNIIC_HILBERT_LANE_SCAN", lane_scan)
    img = synth_img(size, size, seed=size)
    esyms = O.delta_diff(O.hilbert_linearize(img))
    assert np.array_equal(ctx.hilbert_delta(img), esyms)
    keys, counts, syms2 = ctx.hilbert_delta_hist(img, want_syms=True)
    ek, ec = O.count_freqs(esyms)
    assert np.array_equal(keys, ek) and np.array_equal(counts, ec) and np.array_equal(syms2, esyms)


# ------------------------------------------------------------------ huf::encode_all
@pytest.mark.parametrize("n", [1, 2, 17, 4096, 4097, 50000])
def test_huf_encode_all_rgb(ctx, n):
    rng = np.random.default_rng(n)
    syms = (rng.integers(0, 40, n) ** 2 * 1031 % (1 << 24)).astype(np.uint32)
    got = ctx.huf_encode_all(1, syms)
    assert got == O.huf_encode_all(O.SYM_RGB, syms)


def test_huf_encode_all_long_codes(ctx):
    """Fibonacci-like counts give code lengths > 32 bits: exercises the 3-word straddle path."""
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    fib = [min(f, 3000) if i > 20 else f for i, f in enumerate(fib)]
    syms = np.concatenate([np.full(f, i * 7 + 1, np.uint32) for i, f in enumerate(fib)])
    np.random.default_rng(0).shuffle(syms)
    assert ctx.huf_encode_all(2, syms) == O.huf_encode_all(O.SYM_SIGNED, syms)


@pytest.mark.parametrize("runs", [False, True])
@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("n", [2, 3, 17, 4097, 60000])
def test_huf_codes_and_decoder_on_the_gpu(ctx, monkeypatch, kind, n, runs):
    """large alphabets: the host merges the tree (`runs`: merges RUNS of equally frequent leaves and the GPU expands them, as it
    does from 2^18 leaves on), every leaf's code and its place in the serialised decoder come from a walk to the root on the GPU
    (CNIIC_HUF_GPU_CODES_MIN=0: for these small ones too) -- the oracle's bytes for both symbol kinds, skewed and flat
    histograms, and through the `hufman` codec and the 32-bit route of `delta` on a noisy image"""
    monkeypatch.setenv("CNIIC_HUF_GPU_CODES_MIN", "0")
    if runs:
        monkeypatch.setenv("CNIIC_HUF_RUNS_MIN", "0")
    rng = np.random.default_rng(n * 3 + kind)
    top = (1 << 24) if kind == 1 else (1 << 27)
    distinct = rng.choice(top, size=min(n, 5000), replace=False).astype(np.uint32)
    if kind == 2:  # SignedColor keys: three 9-bit fields holding d + 255, d in -255 .. 255
        distinct = (distinct & ~np.uint32(0)) % np.uint32(511) + ((distinct >> 9) % np.uint32(511) << 9) + ((distinct >> 18) % np.uint32(511) << 18)
        distinct = np.unique(distinct.astype(np.uint32))
    p = rng.random(len(distinct)) ** 6
    syms = distinct[rng.choice(len(distinct), size=n, p=p / p.sum())].astype(np.uint32)
    assert ctx.huf_encode_all(kind, syms) == O.huf_encode_all(O.SYM_RGB if kind == 1 else O.SYM_SIGNED, syms)
    if n == 60000 and kind == 1:
        img = rng.integers(0, 256, (120, 130, 3)).astype(np.uint8)
        for expr in ("hufman", "delta"):
            rc, data, _ = ctx.encode(expr, img)
            assert rc == 0 and data == O.encode(expr, img)[1]
            rc, back = ctx.decode(expr, data)
            assert rc == 0 and np.array_equal(back, img)


@pytest.mark.parametrize("runs", [False, True])
def test_huf_long_codes_on_the_gpu(ctx, monkeypatch, runs):
    """Fibonacci-like counts: codes of up to 39 bits through the GPU's walk (`runs`: the tree from runs gives way to the host's
    merge when a code outgrows the 32 bits its sort by code keys on)"""
    monkeypatch.setenv("CNIIC_HUF_GPU_CODES_MIN", "0")
    if runs:
        monkeypatch.setenv("CNIIC_HUF_RUNS_MIN", "0")
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    fib = [min(f, 3000) if i > 20 else f for i, f in enumerate(fib)]
    syms = np.concatenate([np.full(f, i * 7 + 1, np.uint32) for i, f in enumerate(fib)])
    np.random.default_rng(0).shuffle(syms)
    assert ctx.huf_encode_all(2, syms) == O.huf_encode_all(O.SYM_SIGNED, syms)


# ------------------------------------------------------------------ codecs end to end
@pytest.mark.parametrize("expr", ["hufman", "delta", "hilbert(rle)", "cluster-colors(8)", "ccol(256)", "voronoi(6)", "voronoi(40)"])
@pytest.mark.parametrize("shape", [(48, 40), (64, 64), (100, 75)])
def test_codec_bytes_equal_oracle(ctx, expr, shape):
    img = synth_img(*shape, seed=21, levels=64)
    rc, data, st = ctx.encode(expr, img)
    rco, edata, est = O.encode(expr, img, mode=O.MODE_L)
    assert rc == rco == 0
    assert data == edata
    if "col" in expr or "voronoi" in expr:
        assert st["iterations"] == est["iterations"]
    rc, back = ctx.decode(expr, data)
    rco, eback = O.decode(expr, edata)
    assert rc == rco == 0 and np.array_equal(back, eback)
    if expr in ("hufman", "delta", "hilbert(rle)"):
        assert np.array_equal(back, img)
        assert ctx.mse(img, back) == 0.0
    else:
        assert abs(ctx.mse(img, back) - O.mse(img, back)) <= 1e-9 * max(1.0, O.mse(img, back))


def test_codec_edge_cases(ctx):
    from cniic_amd import _lib
    one = np.full((1, 1, 3), 9, np.uint8)
    for expr in ("hufman", "delta", "hilbert(rle)"):
        rc, data, _ = ctx.encode(expr, one)
        assert rc == 0 and data == O.encode(expr, one)[1]
        rc, back = ctx.decode(expr, data)
        assert rc == 0 and np.array_equal(back, one)
    flat = np.zeros((8, 8, 3), np.uint8)
    rc, _, _ = ctx.encode("cluster-colors(4)", flat, allow=(_lib.TOO_FEW_POINTS,))
    assert rc == _lib.TOO_FEW_POINTS
    rc, _ = ctx.decode("hufman", b"\x02\x00\x00\x00\x02\x00\x00\x00\x07", allow=(_lib.DECODE,))
    assert rc == _lib.DECODE



def _delta_img(kind, h, w):
    rng = np.random.default_rng(17)
    y, x = np.mgrid[0:h, 0:w]
    ramp = np.stack([x // 2 + y // 3, x // 3 + (2 * y) // 5, (x + y) // 4], axis=2) + rng.integers(-2, 3, (h, w, 3))
    if kind == "smooth":                      # differences inside the cube [-16, 15]^3 but where a ramp wraps from 255 to 0
        return (ramp & 255).astype(np.uint8)
    if kind == "mixed":                       # about one per cent of the differences outside it, spread thin
        hit = rng.random((h, w)) < 0.004
        ramp[hit] += rng.integers(-120, 120, (int(hit.sum()), 3))
        return (ramp & 255).astype(np.uint8)
    if kind == "noise":                       # nearly all of them outside: a chunk's 64 side entries overflow -> the 32-bit route
        return rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    return np.full((h, w, 3), 77, np.uint8)   # "flat": two symbols, the first pixel's and (0, 0, 0)


@pytest.mark.parametrize("knob", ["", "CNIIC_DELTA_ROUTE=32", "CNIIC_DELTA_GATHER=any", "CNIIC_TEST_PACK_IMG_WORDS=24", "CNIIC_TEST_INLINE_CODE_BITS=5",
                                  "CNIIC_HUF_GPU_CODES_MIN=0"])
@pytest.mark.parametrize("kind", ["smooth", "mixed", "noise", "flat"])
@pytest.mark.parametrize("shape", [(64, 64), (256, 256), (100, 75), (240, 321), (8, 8), (1, 700)])
def test_delta_routes_equal_oracle(ctx, monkeypatch, shape, kind, knob):
    """the `delta` encoder's 16-bit symbol stream (tile gather on 2^n squares from 64 x 64, per-position gather elsewhere; cold
    symbols through the side array; chunks that outgrow the bit image; codes that leave the inline word) and the 32-bit route it
    falls back to on noisy images: the same bytes as the oracle's, and they decode to the image"""
    if knob:
        k, v = knob.split("=")
        monkeypatch.setenv(k, v)
    img = _delta_img(kind, *shape)
    rc, data, _ = ctx.encode("delta", img)
    rco, edata, _ = O.encode("delta", img)
    assert rc == rco == 0 and data == edata
    rc, back = ctx.decode("delta", data)
    assert rc == 0 and np.array_equal(back, img)


def test_delta_single_symbol_and_back_to_back_calls(ctx):
    """an all-zero image is ONE symbol (zero-length code, no payload, huf.rs:140-142); the 2^27-bin table must be clean again for
    the call after a call, whichever route that took"""
    zero = np.zeros((64, 64, 3), np.uint8)
    imgs = [zero, _delta_img("noise", 64, 64), _delta_img("mixed", 128, 128), zero, _delta_img("smooth", 100, 75)]
    for img in imgs + imgs[::-1]:
        rc, data, _ = ctx.encode("delta", img)
        assert rc == 0 and data == O.encode("delta", img)[1]

def test_fuzz_lossless_codecs(ctx):
    """a few seconds of tests/fuzz_codecs.py: random images x route knobs through delta / hufman / hilbert(rle), bytes and round
    trip against the oracle (CNIIC_FUZZ_SECONDS for longer; `python tests/fuzz_codecs.py 150` ran 46 K cases)"""
    import os
    import fuzz_codecs
    assert fuzz_codecs.run(ctx, float(os.environ.get("CNIIC_FUZZ_SECONDS", "6"))) > 0


def test_fuzz_lossy_codecs(ctx):
    """a few seconds of tests/fuzz_kmeans.py: random images x K x route knobs through cluster-colors / voronoi -- return code, bytes,
    iteration count and decode against the oracle (`python tests/fuzz_kmeans.py 300` ran 3.4 K cases)"""
    import os
    import fuzz_kmeans
    assert fuzz_kmeans.run(ctx, float(os.environ.get("CNIIC_FUZZ_SECONDS", "6"))) > 0


@pytest.mark.parametrize("case", ["flat", "stripes", "levels2", "long_runs", "wide", "tall"])
def test_hilbert_rle_runs_equal_oracle(ctx, case):
    """run boundaries, the 255 cap across chunk borders (4096 positions per block) and ragged sizes"""
    rng = np.random.default_rng(5)
    if case == "flat":
        img = np.full((96, 80, 3), 200, np.uint8)                  # one segment of 7680: 30 full runs + 30
    elif case == "stripes":
        img = np.zeros((64, 64, 3), np.uint8); img[:, ::2] = 255
    elif case == "levels2":
        img = rng.integers(0, 2, (128, 128, 1)).astype(np.uint8).repeat(3, axis=2) * 255
    elif case == "long_runs":
        img = np.zeros((256, 256, 3), np.uint8); img[100:, :] = 7; img[5, 5] = 1
    elif case == "wide":
        img = (rng.integers(0, 3, (3, 700, 3)) * 100).astype(np.uint8)
    else:
        img = (rng.integers(0, 3, (513, 5, 3)) * 100).astype(np.uint8)
    rc, data, _ = ctx.encode("hilbert(rle)", img)
    rco, edata, _ = O.encode("hilbert(rle)", img)
    assert rc == rco == 0 and data == edata
    rc, back = ctx.decode("hilbert(rle)", data)
    assert rc == 0 and np.array_equal(back, img)


def test_hilbert_rle_decode_failure_points(ctx):
    from cniic_amd import _lib
    img = np.full((2, 2, 3), 9, np.uint8)
    rc, data, _ = ctx.encode("hilbert(rle)", img)
    rc, back = ctx.decode("hilbert(rle)", data[:8])
    assert rc == 0 and not back.any()
    bad = bytearray(data); bad[8] = 0
    assert ctx.decode("hilbert(rle)", bytes(bad), allow=(_lib.DECODE,))[0] == _lib.DECODE
    assert ctx.decode("hilbert(rle)", data[:15], allow=(_lib.DECODE,))[0] == _lib.DECODE
    rc, back = ctx.decode("hilbert(rle)", data + b"\x07")
    assert rc == 0 and np.array_equal(back, img)


# ------------------------------------------------------------------ cluster-colors through the super-cell partition
@pytest.fixture()
def sp_path(monkeypatch):
    """route every cluster-colors encode through the pixel partition of k_points.hip, whatever the size"""
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", "0")


@pytest.mark.parametrize("expr", ["cluster-colors(8)", "ccol(256)", "cluster-colors(300)"])
@pytest.mark.parametrize("shape", [(7, 5), (48, 40), (100, 75), (301, 299)])
def test_partition_path_equals_oracle(ctx, sp_path, expr, shape):
    """odd pixel counts (the 16-pixel groups end in a partial one), more than one 64 Ki-pixel chunk, u16 labels"""
    from cniic_amd import _lib
    img = synth_img(*shape, seed=23, levels=200, noise=3)
    rc, data, st = ctx.encode(expr, img, allow=(_lib.TOO_FEW_POINTS,))
    rco, edata, est = O.encode(expr, img, mode=O.MODE_L)
    assert rc == rco
    if rc == 0:
        assert data == edata and st["iterations"] == est["iterations"]


def test_partition_path_crowded_bucket_in_slices(ctx, sp_path):
    """560 K pixels in ONE super-cell (every colour below 32) and a second image with two crowded buckets: k_sp_partlab works a
    bucket of more than 2^17 entries in slices (grid.y), every slice with its own table; slice borders fall on entries that are
    not 4-aligned"""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 32, (700, 801, 3)).astype(np.uint8)
    rc, data, st = ctx.encode("cluster-colors(16)", img)
    rco, edata, est = O.encode("cluster-colors(16)", img, mode=O.MODE_L)
    assert rc == rco == 0 and data == edata and st["iterations"] == est["iterations"]
    img2 = rng.integers(0, 32, (611, 523, 3)).astype(np.uint8)
    img2[:, 200:] += 160        # the right part in another super-cell
    rc, data, st = ctx.encode("cluster-colors(8)", img2)
    rco, edata, est = O.encode("cluster-colors(8)", img2, mode=O.MODE_L)
    assert rc == rco == 0 and data == edata and st["iterations"] == est["iterations"]


@pytest.mark.parametrize("cap", ["3", "40"])
@pytest.mark.parametrize("expr", ["cluster-colors(8)", "ccol(256)", "cluster-colors(300)", "hufman", "delta"])
def test_label_pack_direct_route(ctx, monkeypatch, expr, cap):
    """every pack kernel writes a chunk whose bits outgrow its LDS image straight to memory (64-bit codes would: 16 bits per symbol
    fit); CNIIC_TEST_PACK_IMG_WORDS shrinks the image so that every chunk (3 words) or the denser ones (40) go that way -- the
    stream must not change (u8 and u16 labels, several chunks, a partial last one)"""
    monkeypatch.setenv("CNIIC_TEST_PACK_IMG_WORDS", cap)
    img = synth_img(301, 299, seed=29, levels=200, noise=3)
    rc, data, st = ctx.encode(expr, img)
    rco, edata, est = O.encode(expr, img, mode=O.MODE_L)
    assert rc == rco == 0 and data == edata
    if "col" in expr:
        assert st["iterations"] == est["iterations"]


@pytest.mark.parametrize("kind", ["flat2", "dark", "noise"])
def test_partition_path_skewed_colour_distributions(ctx, sp_path, monkeypatch, kind):
    """every pixel in one or two super-cells (one LDS bin takes a whole wave), or spread over all 512"""
    h, w = 260, 300
    rng = np.random.default_rng(5)
    if kind == "flat2":      # two colours: long runs of equal entries in one bucket
        img = np.zeros((h, w, 3), np.uint8)
        img[:, w // 3:] = (200, 10, 77)
        K = 2
    elif kind == "dark":     # all colours below 32: a single super-cell
        img = rng.integers(0, 32, (h, w, 3)).astype(np.uint8)
        K = 16
    else:
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        K = 64
    expr = "cluster-colors(%d)" % K
    rc, data, st = ctx.encode(expr, img)
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", str(1 << 40))   # the dense-table path (checked against the oracle above and elsewhere)
    rc2, data2, st2 = ctx.encode(expr, img)
    assert rc == rc2 == 0 and data == data2 and st["iterations"] == st2["iterations"]
    rco, edata, _ = O.encode(expr, img, mode=O.MODE_L)
    assert rco == 0 and data == edata


@pytest.fixture()
def gpu_decode(monkeypatch):
    """route every Huffman decode through the parallel GPU decoder, whatever the size"""
    monkeypatch.setenv("CNIIC_GPU_DECODE_MIN", "0")


@pytest.mark.parametrize("expr", ["hufman", "delta", "cluster-colors(8)", "ccol(256)"])
@pytest.mark.parametrize("shape", [(1, 1), (7, 3), (48, 40), (100, 75), (256, 192)])
def test_gpu_huffman_decode_equals_oracle(ctx, gpu_decode, expr, shape):
    from cniic_amd import _lib
    img = synth_img(*shape, seed=33, levels=64)
    rc, data, _ = ctx.encode(expr, img, allow=(_lib.TOO_FEW_POINTS,))
    if rc == _lib.TOO_FEW_POINTS:
        pytest.skip("fewer colours than clusters")
    rc, back = ctx.decode(expr, data)
    rco, eback = O.decode(expr, data)
    assert rc == rco == 0 and np.array_equal(back, eback)


def test_gpu_huffman_decode_long_codes_and_failures(ctx, gpu_decode):
    """codes far longer than the 12-bit table (Fibonacci counts), a one-symbol alphabet, streams cut short"""
    from cniic_amd import _lib
    fibs = [1, 1]
    while len(fibs) < 24:
        fibs.append(fibs[-1] + fibs[-2])
    px = np.concatenate([np.full((f, 3), [i * 10 % 256, i, 255 - i], np.uint8) for i, f in enumerate(fibs)])
    np.random.default_rng(1).shuffle(px)
    n = (len(px) // 64) * 64
    img = px[:n].reshape(-1, 64, 3)
    for expr in ("hufman", "delta"):
        rc, data, _ = ctx.encode(expr, img)
        rc, back = ctx.decode(expr, data)
        assert rc == 0 and np.array_equal(back, img)
        for cut in (1, 7, len(data) // 3):
            rc, _ = ctx.decode(expr, data[:-cut], allow=(_lib.DECODE,))
            rco, _ = O.decode(expr, data[:-cut])
            assert (rc == 0) == (rco == 0), (expr, cut)
    flat = np.full((40, 30, 3), 77, np.uint8)
    rc, data, _ = ctx.encode("hufman", flat)
    rc, back = ctx.decode("hufman", data)
    assert rc == 0 and np.array_equal(back, flat)
    # a delta stream whose colours leave 0..255 (hilbertc.rs:505): two pixels, differences +200 twice
    rc, data, _ = ctx.encode("delta", np.array([[[200, 0, 0], [200, 0, 0]]], np.uint8))
    # (the encoder cannot make one; the check is that a valid stream still passes through the range test)
    rc, back = ctx.decode("delta", data)
    assert rc == 0 and back[0, 0, 0] == 200


def test_codec_trait_surface(ctx):
    from cniic_amd import AnyCodec
    c = AnyCodec.from_str("cluster-colors(16)", ctx)
    assert c.name() == "cluster-colors_16" and not c.is_lossless()
    d = AnyCodec.from_str("delta", ctx)
    assert d.name() == "delta" and d.is_lossless()
    img = synth_img(32, 32, seed=2)
    assert np.array_equal(d.decode(d.encode(img)), img)
    assert d.decode(b"\x01\x00\x00\x00\x01\x00\x00\x00") is None
    r = AnyCodec.from_str("hilbert(rle)", ctx)
    assert r.name() == "hilbert-rle" and r.is_lossless()
    assert np.array_equal(r.decode(r.encode(img)), img)
    with pytest.raises(ValueError):
        AnyCodec.from_str("hilbert-rle")


@pytest.mark.parametrize("n_colours,K", [(8, 8), (9, 8), (8, 1), (300, 299), (40, 7)])
def test_partition_path_few_colours(ctx, sp_path, n_colours, K):
    """as many colours as clusters (every chunk of the initial assignment is one point), one more, a single cluster"""
    rng = np.random.default_rng(n_colours * 131 + K)
    pal = rng.choice(1 << 24, n_colours, replace=False).astype(np.uint32)
    idx = np.concatenate([np.arange(n_colours), rng.integers(0, n_colours, 2000 - n_colours)])
    rng.shuffle(idx)
    k = pal[idx]
    img = np.stack([(k >> 16) & 255, (k >> 8) & 255, k & 255], 1).astype(np.uint8).reshape(40, 50, 3)
    expr = "cluster-colors(%d)" % K
    rc, data, st = ctx.encode(expr, img)
    rco, edata, est = O.encode(expr, img, mode=O.MODE_L)
    assert rc == rco == 0 and data == edata and st["iterations"] == est["iterations"]


# ------------------------------------------------------------------ the reference's own known answers, through the HIP path
def _key(r, g, b):
    return (r << 16) | (g << 8) | b


def test_reference_kats_on_the_gpu(ctx):
    """clusterc.rs:304-312 rgb_mean ([0,0,0], [2,2,2] -> [1,1,1]); kmeans.rs:491-500 all_clusters (as many points as
    clusters: each is its own centroid); kmeans.rs:525-539 squares2 (two far squares, K = 2 -> exactly their centres),
    here as 3x3x3 cubes of colours; kmeans.rs:516-523 square1 (K = 1 -> the centre, all members)"""
    rc, r = ctx.kmeans_rgbw([_key(0, 0, 0), _key(2, 2, 2)], [1, 1], 1)
    assert rc == 0 and r["centroids"].tolist() == [[1, 1, 1]] and r["members"].tolist() == [2]
    rc, r = ctx.kmeans_rgbw([_key(0, 0, 0), _key(1, 1, 1)], [1, 1], 2)
    assert rc == 0 and sorted(r["centroids"].tolist()) == [[0, 0, 0], [1, 1, 1]] and sorted(r["labels"].tolist()) == [0, 1]
    cube = lambda c: [_key(c + dr, c + dg, c + db) for dr in (-1, 0, 1) for dg in (-1, 0, 1) for db in (-1, 0, 1)]
    keys = np.array(sorted(cube(50) + cube(200)), np.uint32)
    rc, r = ctx.kmeans_rgbw(keys, np.ones(keys.size, np.uint32), 2)
    assert rc == 0 and sorted(r["centroids"].tolist()) == [[50, 50, 50], [200, 200, 200]] and sorted(r["members"].tolist()) == [27, 27]
    rc, r = ctx.kmeans_rgbw(np.array(sorted(cube(100)), np.uint32), np.ones(27, np.uint32), 1)
    assert rc == 0 and r["centroids"].tolist() == [[100, 100, 100]] and r["members"].tolist() == [27]
    # the weighted mean truncates (clusterc.rs:96-112): (0*3 + 10*1) / 4 = 2
    rc, r = ctx.kmeans_rgbw([_key(0, 0, 0), _key(10, 10, 10)], [3, 1], 1)
    assert rc == 0 and r["centroids"].tolist() == [[2, 2, 2]]


def _voronoi_stream(w, h, cents):
    import struct
    b = struct.pack("<IIQ", w, h, len(cents))
    for (x, y, col) in cents:
        b += struct.pack("<IIQ", x, y, 3) + bytes(col)
    return b


@pytest.mark.parametrize("case", ["dense", "ties", "outside", "huge_coords", "many"])
def test_voronoi_repaint_pruned_equals_brute_force(ctx, case):
    """VoronoiCluster::decode (clusterc.rs:180-186): first minimum of the wrapping u32 key.  The tile-pruned repaint must
    agree with the oracle's brute force, including exact ties (duplicated and mirrored centroids) and centroids outside the
    image; coordinates of 2^14 and more take the brute-force kernel"""
    rng = np.random.default_rng(sum(map(ord, case)))
    w, h = 333, 217
    K = {"dense": 500, "ties": 64, "outside": 40, "huge_coords": 30, "many": 3000}[case]
    xs, ys = rng.integers(0, w, K), rng.integers(0, h, K)
    if case == "ties":
        xs[K // 2:], ys[K // 2:] = xs[:K - K // 2], ys[:K - K // 2]          # every site twice: the lower id must win
        xs[::7] = w - 1 - xs[::7]                                             # and mirrored pairs: equidistant columns
    if case == "outside":
        xs, ys = rng.integers(0, 3 * w, K), rng.integers(0, 3 * h, K)
    if case == "huge_coords":
        xs = rng.integers(0, 1 << 32, K, dtype=np.uint64)
        ys = rng.integers(0, 1 << 32, K, dtype=np.uint64)
    cents = [(int(xs[i]), int(ys[i]), rng.integers(0, 256, 3).astype(np.uint8).tolist()) for i in range(K)]
    data = _voronoi_stream(w, h, cents)
    rc, got = ctx.decode("voronoi(%d)" % K, data)
    rco, exp = O.decode("voronoi(%d)" % K, data)
    assert rc == rco == 0 and np.array_equal(got, exp)
