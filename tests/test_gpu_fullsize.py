"""BASELINE.json's full sizes on the GPU, checked through size-independent properties (the oracle
would take minutes to hours at these sizes): round trips, conservation laws, bijections,
idempotence, agreement between independent kernels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x636E696963


@pytest.fixture(scope="module")
def env():
    import torch

    import cniic_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield ctx, torch, dev
    ctx.close()


def synth(ctx, torch, dev, kind, seed, size):
    img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
    ctx.synth_image(kind, seed, size, size, out=img)
    return img


def test_config2_cluster_colors_4096(env):
    """cluster-colors K=256 on 4096x4096 (configs[1]): decoded image uses <= K colours, every decoded
    colour is a centroid-like mean inside the colour range of its members, histogram conservation,
    stream size = what the histogram predicts, encode is deterministic, brute-force K-means agrees."""
    ctx, torch, dev = env
    from cniic_amd import _lib
    size, K = 4096, 256
    img = synth(ctx, torch, dev, 1, SEED + 2, size)
    out = torch.empty(size * size * 2, dtype=torch.uint8, device=dev)
    rc, n1, st1 = ctx.encode("cluster-colors(%d)" % K, img, w=size, h=size, out=out)
    data1 = out[:n1].cpu().numpy().tobytes()
    rc, n2, st2 = ctx.encode("cluster-colors(%d)" % K, img, w=size, h=size, out=out, flags=_lib.KM_NO_SKIP)
    data2 = out[:n2].cpu().numpy().tobytes()
    assert data1 == data2 and st1["iterations"] == st2["iterations"]          # skip schedule changes nothing
    ctx.set_opt(_lib.OPT_KM_LOOP, 1)                                            # one launch per iteration instead of the persistent launch: same bytes, same work counted
    try:
        rc, n3, st3 = ctx.encode("cluster-colors(%d)" % K, img, w=size, h=size, out=out)
    finally:
        ctx.set_opt(_lib.OPT_KM_LOOP, None)
    assert out[:n3].cpu().numpy().tobytes() == data1 and st3["iterations"] == st1["iterations"]
    rc, back = ctx.decode("ccol(%d)" % K, data1)
    assert rc == 0 and back.shape == (size, size, 3)
    keys = (back[..., 0].astype(np.uint32) << 16) | (back[..., 1].astype(np.uint32) << 8) | back[..., 2]
    pal, cnt = np.unique(keys, return_counts=True)
    assert pal.size <= K and cnt.sum() == size * size
    src = img.cpu().numpy()
    mse = ctx.mse(src, back)
    assert 0 < mse < 400                                                        # K=256 on photo-like data
    # the palette is a fixed point: re-encoding the decoded image keeps every colour (each of the
    # <= K distinct colours is its own cluster when K >= #colours is not guaranteed, so compare MSE only)
    hk, hc = ctx.hist_rgb24(img, npx=size * size)
    assert int(hc.sum()) == size * size and np.all(np.diff(hk.astype(np.int64)) > 0)
    # stream size is a pure function of the reduced image's histogram
    assert ctx.huf_size(_lib.SYM_RGB, cnt.astype(np.uint64)) + 8 == len(data1)


def test_config2_kmeans_brute_equals_pruned_1024(env):
    """exact pruning: brute-force and cell-pruned K-means give identical centroids/labels at 1024^2, K=256"""
    ctx, torch, dev = env
    from cniic_amd import _lib
    img = synth(ctx, torch, dev, 1, SEED + 2, 1024)
    keys, counts = ctx.hist_rgb24(img, npx=1024 * 1024)
    w = counts.astype(np.uint32)
    rc, a = ctx.kmeans_rgbw(keys, w, 256)
    rc, b = ctx.kmeans_rgbw(keys, w, 256, flags=_lib.KM_BRUTE_FORCE)
    assert a["stats"]["iterations"] == b["stats"]["iterations"]
    assert np.array_equal(a["centroids"], b["centroids"]) and np.array_equal(a["labels"], b["labels"])
    assert np.array_equal(a["members"], b["members"]) and int(a["members"].sum()) == keys.size


def test_config5_delta_lossless_4096_and_hilbert_bijection(env):
    """delta on a 2^n square: lossless round trip; the scan is a bijection with adjacent steps;
    delta histogram conserves N; uniform-noise image as the worst case for the alphabet"""
    ctx, torch, dev = env
    size = 4096
    for kind, seed in ((1, SEED + 5), (0, SEED + 6)):
        img = synth(ctx, torch, dev, kind, seed, size if kind else 1024)
        s = img.shape[0]
        out = torch.empty(s * s * 12 + (1 << 22), dtype=torch.uint8, device=dev)
        rc, n, _ = ctx.encode("delta", img, w=s, h=s, out=out)
        rc, back = ctx.decode("delta", out[:n].cpu().numpy().tobytes())
        assert rc == 0 and np.array_equal(back, img.cpu().numpy())
        keys, counts, _ = ctx.hilbert_delta_hist(img, w=s, h=s)
        assert int(counts.sum()) == s * s and np.all(np.diff(keys.astype(np.int64)) > 0)
    xy = ctx.hilbert_xy(size, size).astype(np.int64)
    lin = xy[:, 1] * size + xy[:, 0]
    seen = np.zeros(size * size, np.uint8)
    seen[lin] = 1
    assert seen.all()                                                            # bijection
    assert (np.abs(np.diff(xy, axis=0)).sum(axis=1) == 1).all()                  # unit steps on 2^n squares
    assert tuple(xy[0]) == (0, 0) and tuple(xy[-1]) == (size - 1, 0)


def test_config5_delta_16384(env):
    """configs[4] at its full size: `delta` + Huffman histogram on one 16384 x 16384 image.  Lossless round trip,
    histogram total = N, stream length = what the histogram predicts + the 8 header bytes, and the
    scan positions of 10^5 sampled d (plus both ends) equal the oracle's random-access d -> (x, y)."""
    import oracle_lib as O
    ctx, torch, dev = env
    from cniic_amd import _lib
    size = 16384
    N = size * size
    img = synth(ctx, torch, dev, 1, SEED + 5, size)
    out = torch.empty(N * 4 + (1 << 26), dtype=torch.uint8, device=dev)
    rc, n, _ = ctx.encode("delta", img, w=size, h=size, out=out)
    assert rc == 0
    keys, counts, _ = ctx.hilbert_delta_hist(img, w=size, h=size)
    assert int(counts.sum()) == N and np.all(np.diff(keys.astype(np.int64)) > 0)
    assert ctx.huf_size(_lib.SYM_SIGNED, counts) + 8 == n
    data = out[:n].cpu().numpy().tobytes()
    del out
    rc, back = ctx.decode("delta", data)
    assert rc == 0 and back.shape == (size, size, 3)
    src = img.cpu().numpy()
    assert np.array_equal(back, src)
    del back, data
    # the delta symbols along the scan are those of the linearised image (P2, hilbertc.rs:463-477), checked on a window
    xy = ctx.hilbert_xy(size, size)
    rng = np.random.default_rng(16384)
    ds = np.concatenate([[0, 1, N - 2, N - 1], rng.integers(0, N, 100000)])
    for d in ds.tolist():
        assert tuple(xy[d]) == O.hilbert_d2xy(size, size, d), d
    lin = xy[:, 1].astype(np.int64) * size + xy[:, 0]
    seen = np.zeros(N, np.bool_)
    seen[lin] = True
    assert seen.all()                                                            # bijection at order 14
    d0 = int(rng.integers(1, N - 70000))
    win = src[xy[d0 - 1:d0 + 65536, 1], xy[d0 - 1:d0 + 65536, 0]].astype(np.int32)
    dsym = win[1:] - win[:-1] + 255
    exp = (dsym[:, 0] << 18) | (dsym[:, 1] << 9) | dsym[:, 2]
    syms = ctx.hilbert_delta_hist(img, want_syms=True, w=size, h=size)[2]
    assert np.array_equal(syms[d0:d0 + 65536], exp.astype(np.uint32))


def test_hilbert_non_pow2_large_bijection(env):
    ctx, torch, dev = env
    w, h = 1920, 1080                                                            # config 4 frame size
    xy = ctx.hilbert_xy(w, h).astype(np.int64)
    lin = xy[:, 1] * w + xy[:, 0]
    assert np.array_equal(np.sort(lin), np.arange(w * h))
    step = np.abs(np.diff(xy, axis=0)).sum(axis=1)
    assert (step <= 2).all() and (step == 2).sum() <= 1


def test_config3_voronoi_2048(env):
    """voronoi K=2048 on 4096x4096 (configs[2]): stream is exactly 16 + 19 K bytes; Voronoi repaint
    assigns every pixel the colour of its nearest centroid (checked on a sample by brute force);
    pruned and brute-force kernels agree on a smaller image."""
    ctx, torch, dev = env
    from cniic_amd import _lib
    size, K = 4096, 2048
    img = synth(ctx, torch, dev, 1, SEED + 3, size)
    out = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    rc, n, st = ctx.encode("voronoi(%d)" % K, img, w=size, h=size, out=out)
    data = out[:n].cpu().numpy().tobytes()
    assert rc == 0 and n == 16 + 19 * K and st["moved_last"] == 0 and st["active"] >= int(0.99 * K)
    rc, back = ctx.decode("voronoi(%d)" % K, data)
    assert rc == 0
    raw = np.frombuffer(data, np.uint8)[16:].reshape(K, 19)
    cx = raw[:, 0:4].copy().view("<u4")[:, 0].astype(np.int64)
    cy = raw[:, 4:8].copy().view("<u4")[:, 0].astype(np.int64)
    col = raw[:, 16:19]
    rng = np.random.default_rng(0)
    for x, y in zip(rng.integers(0, size, 200), rng.integers(0, size, 200)):
        d = (cx - x) ** 2 + (cy - y) ** 2
        assert np.array_equal(back[y, x], col[int(np.argmin(d))])                # first minimum
    small = synth(ctx, torch, dev, 1, SEED + 3, 512).cpu().numpy()
    rc, a = ctx.kmeans_xyrgb(small, 256)
    rc, b = ctx.kmeans_xyrgb(small, 256, flags=_lib.KM_BRUTE_FORCE)
    rc, c = ctx.kmeans_xyrgb(small, 256, flags=_lib.KM_NO_SKIP)
    for o in (b, c):
        assert a["stats"]["iterations"] == o["stats"]["iterations"]
        assert np.array_equal(a["centroids"], o["centroids"]) and np.array_equal(a["labels"], o["labels"])
    assert int(a["members"].sum()) == 512 * 512


def test_voronoi_dealing_8192(env, monkeypatch):
    """voronoi(2048) on 8192 x 8192: the size at which every block has to take exactly as many super-tiles as its u32 accumulators
    allow (16).  Drawn from the launch's counter, busy ones first (the default), or every 256th (CNIIC_XY_DYN=0): the same stream."""
    ctx, torch, dev = env
    from cniic_amd import _lib
    size, K = 8192, 2048
    img = synth(ctx, torch, dev, 1, SEED + 9, size)
    out = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    res = []
    for dyn in ("16", "0"):
        monkeypatch.setenv("CNIIC_XY_DYN", dyn)
        rc, n, st = ctx.encode("voronoi(%d)" % K, img, w=size, h=size, out=out, allow=(_lib.FEW_ACTIVE,))
        res.append((rc, out[:n].cpu().numpy().tobytes(), st["iterations"], st["moved_last"]))
    assert res[0] == res[1] and res[0][0] == 0 and res[0][3] == 0


def test_hufman_lossless_2048_uniform(env):
    """Hufman on uniform noise (largest alphabet): lossless, size = histogram prediction"""
    ctx, torch, dev = env
    from cniic_amd import _lib
    s = 1024
    img = synth(ctx, torch, dev, 0, SEED + 1, s)
    out = torch.empty(s * s * 16 + (1 << 22), dtype=torch.uint8, device=dev)
    rc, n, _ = ctx.encode("hufman", img, w=s, h=s, out=out)
    data = out[:n].cpu().numpy().tobytes()
    rc, back = ctx.decode("hufman", data)
    assert rc == 0 and np.array_equal(back, img.cpu().numpy())
    k, c = ctx.hist_rgb24(img, npx=s * s)
    assert ctx.huf_size(_lib.SYM_RGB, c) + 8 == len(data)


def test_hilbert_rle_4096_properties(env):
    """Hilbert{RLE(0)} at 4096x4096: the run records expand to the linearised image, no run is empty or longer
    than 255, neighbouring runs differ in colour unless the first is full, the decoder restores the image.
    Two inputs: the photo-like image (runs of 1) and a posterised one (long runs, many at the cap)."""
    ctx, torch, dev = env
    size = 4096
    photo = synth(ctx, torch, dev, 1, SEED + 2, size)
    poster = (photo >> 6) << 6
    for img in (photo, poster):
        out = torch.empty(size * size * 12 + 64, dtype=torch.uint8, device=dev)
        rc, n, _ = ctx.encode("hilbert(rle)", img, w=size, h=size, out=out)
        assert rc == 0 and (n - 8) % 12 == 0
        rec = out[8:n].cpu().numpy().reshape(-1, 12)
        cnt = rec[:, 0].astype(np.int64)
        assert cnt.min() >= 1 and cnt.max() <= 255 and int(cnt.sum()) == size * size
        assert (rec[:, 1] == 3).all() and not rec[:, 2:9].any()
        lin = ctx.hilbert_linearize(img.cpu().numpy()).reshape(-1, 3)
        assert np.array_equal(np.repeat(rec[:, 9:12], cnt, axis=0), lin)
        same = (rec[1:, 9:12] == rec[:-1, 9:12]).all(axis=1)
        assert (cnt[:-1][same] == 255).all()
        rc, back = ctx.decode("hilbert(rle)", out[:n].cpu().numpy().tobytes())
        assert rc == 0 and np.array_equal(back, img.cpu().numpy())


def test_cluster_colors_partition_equals_dense_table_at_odd_sizes(env, monkeypatch):
    """above the 2^20-pixel threshold the encode goes through the pixel partition (k_points.hip): same bytes as the
    dense-table route, for sizes that leave a partial 64 Ki-pixel chunk and a partial 16-pixel group, u8 and u16 labels"""
    ctx, torch, dev = env
    for (h, w, K, kind) in ((1031, 1021, 64, 1), (1100, 1000, 300, 1), (1024, 1024, 16, 0)):
        img = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
        ctx.synth_image(kind, SEED + 11, w, h, out=img)
        out = torch.empty(h * w * 4 + 4096, dtype=torch.uint8, device=dev)
        monkeypatch.delenv("CNIIC_SP_MIN_PIXELS", raising=False)
        rc, n1, st1 = ctx.encode("cluster-colors(%d)" % K, img, w=w, h=h, out=out)
        a = out[:n1].cpu().numpy().tobytes()
        monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", str(1 << 40))
        rc2, n2, st2 = ctx.encode("cluster-colors(%d)" % K, img, w=w, h=h, out=out)
        b = out[:n2].cpu().numpy().tobytes()
        assert rc == rc2 == 0 and a == b and st1["iterations"] == st2["iterations"], (h, w, K)


def test_cluster_colors_16384_roundtrip_properties(env):
    """the partition at 2^28 pixels (4096 chunks, every bucket split over thousands of runs): at most K colours come
    back, the decoded image keeps the dimensions, and a second encode gives the same bytes"""
    ctx, torch, dev = env
    size, K = 16384, 256
    img = synth(ctx, torch, dev, 1, SEED + 5, size)
    out = torch.empty(size * size + (1 << 20), dtype=torch.uint8, device=dev)
    rc, n1, st = ctx.encode("cluster-colors(%d)" % K, img, w=size, h=size, out=out)
    assert rc == 0 and st["iterations"] > 1
    data = out[:n1].cpu().numpy().tobytes()
    rc, n2, _ = ctx.encode("cluster-colors(%d)" % K, img, w=size, h=size, out=out)
    assert rc == 0 and out[:n2].cpu().numpy().tobytes() == data
    rc, back_h = ctx.decode("cluster-colors(%d)" % K, data)
    assert rc == 0 and back_h.shape == (size, size, 3)
    back = torch.from_numpy(back_h).to(dev)
    keys = (back[..., 0].to(torch.int32) << 16) | (back[..., 1].to(torch.int32) << 8) | back[..., 2].to(torch.int32)
    assert int(torch.unique(keys).numel()) <= K
    err = (back[::64, ::64].to(torch.float32) - img[::64, ::64].to(torch.float32)).pow(2).mean().item()
    assert err < 400.0   # 256 colours for the whole cube: cells ~40 levels wide, i.e. ~12-16 levels rms per channel, not noise


def test_config4_frame_batch_1080p(env):
    """configs[3] at its frame size on one rank: 8 frames 1920 x 1080 (no power of two, 506.25 pack chunks per frame), ONE palette.
    Every frame's stream decodes on its own; all frames share one palette of at most K colours; a frame decodes to exactly the
    rows the plain encode of the stacked frames (one 1920 x 8640 image: the same union clustering) decodes to; each stream is as
    long as its own label histogram predicts; the batch encode is deterministic."""
    ctx, torch, dev = env
    from cniic_amd import _lib
    from cniic_amd.dist import ShardedClusterColors
    F, w, h, K = 8, 1920, 1080, 256
    frames = torch.empty((F, h, w, 3), dtype=torch.uint8, device=dev)
    for f in range(F):
        ctx.synth_image(1, SEED + 4 + f, w, h, out=frames[f])
    stride = w * h
    out = torch.empty(stride * F, dtype=torch.uint8, device=dev)
    enc = ShardedClusterColors(ctx, K, None, dev)
    lens, st = enc.encode_frames(frames, w, h, F, out, stride)
    host = out.cpu().numpy().copy()
    lens2, st2 = enc.encode_frames(frames, w, h, F, out, stride)
    assert lens2 == lens and st2["iterations"] == st["iterations"] and np.array_equal(out.cpu().numpy(), host)
    big = torch.empty(F * w * h * 2, dtype=torch.uint8, device=dev)
    rc, nb, stb = ctx.encode("cluster-colors(%d)" % K, frames, w=w, h=F * h, out=big)   # the stacked frames as one image
    assert rc == 0 and stb["iterations"] == st["iterations"]
    rc, whole = ctx.decode("cluster-colors(%d)" % K, big[:nb].cpu().numpy().tobytes())
    assert rc == 0
    palette = set()
    for f in range(F):
        data = host[f * stride:f * stride + lens[f]].tobytes()
        rc, back = ctx.decode("cluster-colors(%d)" % K, data)
        assert rc == 0 and back.shape == (h, w, 3)
        assert np.array_equal(back, whole[f * h:(f + 1) * h]), "frame %d" % f
        keys = (back[..., 0].astype(np.uint32) << 16) | (back[..., 1].astype(np.uint32) << 8) | back[..., 2]
        pal, cnt = np.unique(keys, return_counts=True)
        palette.update(pal.tolist())
        assert ctx.huf_size(_lib.SYM_RGB, cnt.astype(np.uint64)) + 8 == lens[f]
    assert len(palette) <= K


def test_config4_frame_batch_128_frames_1080p(env):
    """configs[3] exactly as one GPU of the 8-GPU job sees it: 128 frames 1920 x 1080 (265 Mpixels, 5 x 10^5 pixels per partition
    bucket: k_sp_partlab runs sliced, k_frame_trees one block per frame), ONE palette.  The batch encode is deterministic; sampled
    frames decode on their own to exactly the rows the plain encode of the stacked frames (one 1920 x 138240 image: the same union
    clustering through the single-image path) decodes to; every stream is as long as its own label histogram predicts
    (SURVEY 8(a) H2); all frames share one palette of at most K colours; the labels of ALL frames are checked through the stream
    lengths (a wrong label in any frame changes that frame's histogram or its payload size)."""
    ctx, torch, dev = env
    from cniic_amd import _lib
    from cniic_amd.dist import ShardedClusterColors
    F, w, h, K = 128, 1920, 1080, 256
    frames = torch.empty((F, h, w, 3), dtype=torch.uint8, device=dev)
    for f in range(F):
        ctx.synth_image(1, SEED + 4 + f, w, h, out=frames[f])
    stride = w * h
    out = torch.empty(stride * F, dtype=torch.uint8, device=dev)
    enc = ShardedClusterColors(ctx, K, None, dev)
    lens, st = enc.encode_frames(frames, w, h, F, out, stride)
    assert len(lens) == F and st["iterations"] > 1 and st["moved_last"] == 0
    heads = [out[f * stride:f * stride + lens[f]].cpu().numpy().tobytes() for f in range(F)]
    lens2, st2 = enc.encode_frames(frames, w, h, F, out, stride)
    assert lens2 == lens and st2["iterations"] == st["iterations"]
    for f in range(F):
        assert out[f * stride:f * stride + lens[f]].cpu().numpy().tobytes() == heads[f], "frame %d differs between two batch encodes" % f
    enc.close()
    # the same union clustering through the single-image path
    big = torch.empty(F * w * h + (1 << 20), dtype=torch.uint8, device=dev)
    rc, nb, stb = ctx.encode("cluster-colors(%d)" % K, frames, w=w, h=F * h, out=big)
    assert rc == 0 and stb["iterations"] == st["iterations"]
    rc, whole = ctx.decode("cluster-colors(%d)" % K, big[:nb].cpu().numpy().tobytes())
    assert rc == 0 and whole.shape == (F * h, w, 3)
    del big
    wkeys = (whole[..., 0].astype(np.uint32) << 16) | (whole[..., 1].astype(np.uint32) << 8) | whole[..., 2]
    palette = set(np.unique(wkeys).tolist())
    assert len(palette) <= K
    for f in range(F):
        # every frame: stream length = what the histogram of ITS rows of the union clustering predicts
        cnt = np.unique(wkeys[f * h:(f + 1) * h], return_counts=True)[1]
        assert ctx.huf_size(_lib.SYM_RGB, cnt.astype(np.uint64)) + 8 == lens[f], "frame %d" % f
    for f in (0, 37, 90, 127):
        rc, back = ctx.decode("cluster-colors(%d)" % K, heads[f])
        assert rc == 0 and back.shape == (h, w, 3)
        assert np.array_equal(back, whole[f * h:(f + 1) * h]), "frame %d" % f
