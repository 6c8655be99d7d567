"""Property tests of the oracle the reference lacks: Hilbert bijection /
adjacency, delta and codec round trips, K-means mode L vs mode R agreement."""
import numpy as np
import pytest

import oracle_lib as O


def classic_d2xy(n, d):
    """Textbook Hilbert d2xy for an n x n (power of two) square."""
    x = y = 0
    t = d
    s = 1
    while s < n:
        rx = 1 & (t // 2)
        ry = 1 & (t ^ rx)
        if ry == 0:
            if rx == 1:
                x, y = s - 1 - x, s - 1 - y
            x, y = y, x
        x += s * rx
        y += s * ry
        t //= 4
        s *= 2
    return x, y


@pytest.mark.parametrize("w,h", [(1, 1), (1, 7), (7, 1), (2, 2), (4, 4), (8, 8), (16, 16), (5, 3), (3, 5),
                                 (13, 8), (8, 13), (17, 17), (31, 10), (10, 31), (64, 48), (100, 37), (6, 6)])
def test_hilbert_bijective_adjacent_and_d2xy(w, h):
    xy = O.hilbert_iter(w, h).astype(np.int64)
    assert xy.shape == (w * h, 2)
    lin = xy[:, 1] * w + xy[:, 0]
    assert np.array_equal(np.sort(lin), np.arange(w * h))          # bijection
    assert (xy[:, 0] < w).all() and (xy[:, 1] < h).all()
    assert tuple(xy[0]) == (0, 0)
    step = np.abs(np.diff(xy, axis=0)).sum(axis=1)
    # consecutive cells are neighbours; odd x odd rectangles may need one diagonal step
    assert (step <= 2).all() and (step == 2).sum() <= 1
    for d in range(w * h):                                           # random access == generator
        assert O.hilbert_d2xy(w, h, d) == tuple(xy[d])


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64])
def test_hilbert_pow2_is_classic(n):
    xy = O.hilbert_iter(n, n)
    for d in range(n * n):
        assert classic_d2xy(n, d) == tuple(xy[d])


def synth(h, w, seed=0, levels=256):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, levels, (h // 4 + 1, w // 4 + 1, 3))
    img = np.kron(base, np.ones((4, 4, 1), np.int64))[:h, :w]
    img = img + rng.integers(-2, 3, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("codec", ["hufman", "delta", "hilbert(rle)"])
@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (16, 16), (33, 20)])
def test_lossless_roundtrip(codec, shape):
    img = synth(*shape, seed=3)
    rc, data, _ = O.encode(codec, img)
    assert rc == 0
    rc, back = O.decode(codec, data)
    assert rc == 0 and np.array_equal(back, img)
    assert O.mse(img, back) == 0.0


def test_hilbert_rle_records_and_run_cap():
    """Hilbert{RLE(0.0)} (hilbertc.rs:26-39, 100-196): dims, then (count:u8, colour = u64 len 3 + 3 bytes) per run;
    a run stops at RepCount::MAX = 255 elements (:128-137) and the stream is the scan order of hilbert.rs"""
    flat = np.full((32, 32, 3), 5, np.uint8)                       # 1024 equal pixels
    rc, data, _ = O.encode("hilbert(rle)", flat)
    assert rc == 0 and data[:8] == (32).to_bytes(4, "little") * 2
    recs = [data[8 + 12 * i: 20 + 12 * i] for i in range((len(data) - 8) // 12)]
    assert [r[0] for r in recs] == [255, 255, 255, 255, 4]
    assert all(r[1:] == (3).to_bytes(8, "little") + bytes([5, 5, 5]) for r in recs)
    # runs follow the scan: the sequence of decoded colours is the linearised image
    img = synth(12, 20, seed=9, levels=3)
    rc, data, _ = O.encode("Hilbert(rle(0))", img)
    lin = O.hilbert_linearize(img).reshape(-1, 3)
    out = []
    for i in range((len(data) - 8) // 12):
        r = data[8 + 12 * i: 20 + 12 * i]
        out += [list(r[9:12])] * r[0]
    assert np.array_equal(np.array(out, np.uint8), lin)
    # no two consecutive runs share a colour unless the first one is full
    for i in range((len(data) - 8) // 12 - 1):
        a_, b_ = data[8 + 12 * i: 20 + 12 * i], data[20 + 12 * i: 32 + 12 * i]
        assert a_[9:12] != b_[9:12] or a_[0] == 255


def test_hilbert_rle_decoder_failure_points():
    """RleDecoder (hilbertc.rs:304-337): a short stream leaves black pixels, a zero count or a cut colour fail"""
    img = np.full((2, 2, 3), 9, np.uint8)
    rc, data, _ = O.encode("hilbert(rle)", img)
    assert data[8] == 4
    rc, back = O.decode("hilbert(rle)", data[:8])                   # no runs at all: ImageBuffer::new zeros
    assert rc == 0 and not back.any()
    short = bytearray(data); short[8] = 3                           # three of the four pixels
    rc, back = O.decode("hilbert(rle)", bytes(short))
    assert rc == 0 and int((back == 9).all(axis=2).sum()) == 3
    bad = bytearray(data); bad[8] = 0                               # assert!(self.count > 0)
    assert O.decode("hilbert(rle)", bytes(bad))[0] != 0
    assert O.decode("hilbert(rle)", data[:15])[0] != 0              # colour cut: unwrap()
    assert O.decode("hilbert(rle)", data + b"\x07")[0] == 0          # bytes after the last needed run are never read


def test_hufman_header_and_single_colour():
    img = np.full((4, 6, 3), 7, np.uint8)
    rc, data, _ = O.encode("Hufman", img)
    assert rc == 0
    assert data[:8] == (6).to_bytes(4, "little") + (4).to_bytes(4, "little")   # (w,h) hufc.rs:13
    assert len(data) == 8 + 12                                               # one leaf, no payload
    rc, back = O.decode("hufman", data)
    assert rc == 0 and np.array_equal(back, img)


@pytest.mark.parametrize("mode", [O.MODE_R, O.MODE_L])
def test_cluster_colors_roundtrip(mode):
    img = synth(48, 40, seed=5, levels=64)
    rc, data, st = O.encode("cluster-colors(8)", img, mode=mode)
    assert rc == 0 and st["iterations"] >= 1
    rc, back = O.decode("ccol(8)", data)
    assert rc == 0 and back.shape == img.shape
    assert len(np.unique(back.reshape(-1, 3), axis=0)) <= 8
    assert 0 < O.mse(img, back) < 2000


def test_cluster_colors_too_few_unique_colours():
    img = np.zeros((8, 8, 3), np.uint8)           # 1 unique colour < K
    rc, _, _ = O.encode("cluster-colors(4)", img)
    assert rc == O.TOO_FEW_POINTS                  # kmeans.rs:68


@pytest.mark.parametrize("mode", [O.MODE_R, O.MODE_L])
def test_voronoi_size_and_decode(mode):
    img = synth(24, 32, seed=9)
    K = 6
    rc, data, _ = O.encode("voronoi(%d)" % K, img, mode=mode)
    assert rc == 0
    assert len(data) == 16 + 19 * K                # clusterc.rs:155-165
    rc, back = O.decode("voronoi(%d)" % K, data)
    assert rc == 0 and back.shape == img.shape
    assert len(np.unique(back.reshape(-1, 3), axis=0)) <= K


def test_voronoi_decode_first_min_ties():
    """clusterc.rs:180-186: min_by_key keeps the FIRST of equal minima"""
    def cent(x, y, c):
        return x.to_bytes(4, "little") + y.to_bytes(4, "little") + (3).to_bytes(8, "little") + bytes(c)
    data = (3).to_bytes(4, "little") + (1).to_bytes(4, "little") + (2).to_bytes(8, "little")
    data += cent(0, 0, (10, 10, 10)) + cent(2, 0, (20, 20, 20))
    rc, back = O.decode("voronoi(2)", data)
    assert rc == 0
    assert back[0, :, 0].tolist() == [10, 10, 20]   # x=1 is equidistant -> first centroid


def test_modeL_modeR_agree_while_lists_are_full():
    """With K small the neighbour lists are never truncated below K-1 (floor sqrt(K) >= K-1 for K<=2,
    and 2*watermark grows), and without exact ties both modes are the same Lloyd iteration."""
    rng = np.random.default_rng(11)
    pts = np.concatenate([rng.normal(c, 6, (200, 3)) for c in ((40, 40, 40), (200, 60, 90))]).clip(0, 255).astype(np.int32)
    pts = np.unique(pts, axis=0)
    w = rng.integers(1, 50, len(pts)).astype(np.uint32)
    rcR, R = O.kmeans(O.PT_RGBW, O.MODE_R, pts, w, 2)
    rcL, L = O.kmeans(O.PT_RGBW, O.MODE_L, pts, w, 2)
    assert rcR == 0 and rcL == 0
    assert np.array_equal(R["centroids"], L["centroids"])
    assert np.array_equal(R["labels"], L["labels"])


def test_modeL_step_loop_equals_run():
    """orc_kmeans mode L == init + repeated (step, finalize)"""
    rng = np.random.default_rng(12)
    pts = np.unique(rng.integers(0, 256, (3000, 3)).astype(np.int32), axis=0)
    w = rng.integers(1, 9, len(pts)).astype(np.uint32)
    K = 16
    rc, run = O.kmeans(O.PT_RGBW, O.MODE_L, pts, w, K)
    assert rc == 0
    n = len(pts)
    labels = O.init_labels(n, K)
    ppc = n // K
    cent = np.stack([pts[n - (c + 1) * ppc] if c < K - 1 else pts[0] for c in range(K)])
    it = 0
    while True:
        st = O.kmeans_step(O.PT_RGBW, pts, w, K, cent, labels)
        labels = st["labels"]
        cent, _ = O.kmeans_finalize(O.PT_RGBW, pts, K, O.DEFAULT_SEED, it, st["sums"], st["wsum"], st["members"])
        it += 1
        if st["changed"] == 0:
            break
    assert it == run["stats"]["iterations"]
    assert np.array_equal(cent, run["centroids"]) and np.array_equal(labels, run["labels"])


def test_empty_cluster_reseed_is_deterministic():
    """kmeans.rs:117-134 with deviation D2: an empty cluster takes the point picked by splitmix64"""
    pts = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [250, 0, 0]], np.int32)
    w = np.ones(4, np.uint32)
    sums = np.zeros((2, 3), np.uint64); wsum = np.zeros(2, np.uint64); members = np.zeros(2, np.uint64)
    sums[0] = (253, 0, 0); wsum[0] = 4; members[0] = 4            # cluster 1 empty
    cent, nres = O.kmeans_finalize(O.PT_RGBW, pts, 2, 99, 5, sums, wsum, members)
    assert nres == 1
    idx = O.reseed_index(99, 5, 1, 4)
    assert list(cent[1]) == list(pts[idx]) and list(cent[0]) == [63, 0, 0]
