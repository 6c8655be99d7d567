"""Known-answer tests that PIN the oracle: every hot-path unit test the reference holds,
restated against oracle/liboracle.so (file:line of the reference test in each docstring)."""
import math

import numpy as np
import pytest

import oracle_lib as O


# ---------------------------------------------------------------- src/bit.rs:261-493
def test_write_x00():
    """bit.rs:279-287"""
    bw = O.BitWriter()
    for _ in range(8):
        bw.write(0)
    assert bw.into_inner() == bytes([0])


def test_write_xff():
    """bit.rs:289-297"""
    bw = O.BitWriter()
    for _ in range(8):
        bw.write(1)
    assert bw.into_inner() == bytes([0xFF])


def test_interleaved_byte():
    """bit.rs:299-322 -> [0x5e, 0x0c]"""
    bw = O.BitWriter()
    for b in (0, 1, 0):
        bw.write(b)
    bw.write_byte(0xF0)
    for b in (0, 1, 1, 0, 0):
        bw.write(b)
    assert bw.into_inner() == bytes([0x5E, 0x0C])


def test_bw_mask():
    """bit.rs:324-349 -> [0x0d, 0xfe]"""
    bw = O.BitWriter()
    for b in (0, 0, 0, 0, 1, 1, 0):
        bw.write(b)
    bw.write_byte(0xFF)
    bw.write(0)
    assert bw.into_inner() == bytes([0x0D, 0xFE])


@pytest.mark.parametrize("n,exp", [(0, 0), (1, 1), (2, 3), (3, 7), (4, 15), (5, 31), (6, 63), (7, 127), (8, 255), (9, 255)])
def test_bit_mask(n, exp):
    """bit.rs:351-399 bit_mask0..9"""
    assert O.bit_mask(n) == exp


def test_nth_lsb_msb():
    """bit.rs:401-429"""
    byte = 0b10110010
    assert [O.bit_nth(byte, i, msb_first=False) for i in range(8)] == [0, 1, 0, 0, 1, 1, 0, 1]
    assert [O.bit_nth(byte, i, msb_first=True) for i in range(8)] == [1, 0, 1, 1, 0, 0, 1, 0]


@pytest.mark.parametrize("bits,exp", [
    ([0], bytes([0x00])),                       # bit_packing0 (bit.rs:431-442)
    ([1], bytes([0x80])),                       # bit_packing1
    ([1, 1, 1, 1, 0, 0, 0, 0], bytes([0xF0])),  # bit_packing2
    ([1, 1, 1, 1, 0, 0, 0, 0, 1], bytes([0xF0, 0x80])),  # bit_packing3
])
def test_bit_packing_roundtrip(bits, exp):
    """bit.rs:431-493: BitArray::from_slice then write_arr + pad must reproduce the bits MSB-first."""
    bw = O.BitWriter()
    bw.write_code(bits)
    bw.pad_and_flush()
    assert bw.into_inner() == exp


def test_pad_and_flush_noop_when_aligned():
    """bit.rs:243-253"""
    bw = O.BitWriter()
    bw.write_byte(0xAB)
    bw.pad_and_flush()
    assert bw.into_inner() == bytes([0xAB])


# ---------------------------------------------------------------- src/ser.rs
def test_ser_primitives():
    """ser.rs:31-35,49-53,67-71,87-91 little-endian; ser.rs:210-214 Rgb = u64 len + 3 bytes"""
    assert O.ser("u8", 0x12) == b"\x12"
    assert O.ser("u16", 0x1234) == b"\x34\x12"
    assert O.ser("i16", -2) == b"\xfe\xff"
    assert O.ser("u32", 0x01020304) == b"\x04\x03\x02\x01"
    assert O.ser("u64", 3) == b"\x03" + b"\0" * 7
    assert O.ser("rgb", (1, 2, 3)) == b"\x03" + b"\0" * 7 + b"\x01\x02\x03"
    assert len(O.ser("rgb", (9, 9, 9))) == 11


# ---------------------------------------------------------------- src/huf.rs:376-540
ABC = dict(a=2, b=1, c=1)  # huf_abc() huf.rs:385-387


def _abc():
    keys = np.array([ord(c) for c in "abc"], np.uint32)
    counts = np.array([2, 1, 1], np.uint64)
    return keys, counts


def test_builder_is_sane_and_code_lens1():
    """huf.rs:389-394, 417-424: a:2,b:1,c:1 -> lengths 1,2,2"""
    _, counts = _abc()
    lens, codes = O.huf_build(counts)
    assert list(lens) == [1, 2, 2]
    # prefix-free
    cs = {format(int(c), "0%db" % l) for c, l in zip(codes, lens)}
    assert len(cs) == 3
    for a in cs:
        for b in cs:
            assert a == b or not b.startswith(a)


def test_bintrie_single_leaf():
    """huf.rs:396-402 + huf.rs:140-142: one symbol -> zero-length code, no payload"""
    lens, _ = O.huf_build(np.array([5], np.uint64))
    assert list(lens) == [0]
    data = O.huf_encode_all(O.SYM_CHAR, np.full(7, ord("x"), np.uint32))
    assert data == bytes([0, ord("x")])          # leaf tag + symbol, empty payload
    rc, syms = O.huf_decode_all(O.SYM_CHAR, data, 7)
    assert rc == 0 and all(syms == ord("x"))


def test_bintrie_iter2():
    """huf.rs:404-415: two leaves -> codes 0 and 1 in left-to-right order"""
    lens, codes = O.huf_build(np.array([1, 1], np.uint64))
    assert list(lens) == [1, 1]
    assert sorted(int(c) for c in codes) == [0, 1]


@pytest.mark.parametrize("text", ["a", "abcabcaabbcc"])
def test_enc_dec(text):
    """huf.rs:435-487 enc_dec1-3 (incl. decoder serialise/deserialise round trip) and ser1 (489-498)"""
    syms = np.array([ord(c) for c in text], np.uint32)
    data = O.huf_encode_all(O.SYM_CHAR, syms)
    rc, out = O.huf_decode_all(O.SYM_CHAR, data, len(text))
    assert rc == 0
    assert "".join(chr(c) for c in out) == text


def test_encode1_encode2():
    """huf.rs:500-539: explicit code table -> [0x5e,0x0c] and [0xf0]"""
    codes = {"a": [0, 1, 0], "b": [1, 1, 1, 1, 0, 0, 0, 0, 0, 1, 1], "c": [0, 0]}
    bw = O.BitWriter()
    for ch in "abc":
        bw.write_code(codes[ch])
    bw.pad_and_flush()
    assert bw.into_inner() == bytes([0x5E, 0x0C])
    bw = O.BitWriter()
    bw.write_code([1, 1, 1, 1, 0, 0, 0, 0])
    bw.pad_and_flush()
    assert bw.into_inner() == bytes([0xF0])


def test_huf_stream_layout_abc():
    """huf.rs:22-43 + 299-321: trie pre-order (1=branch,0=leaf+symbol) then MSB-first payload.
    a:2,b:1,c:1 -> heap pops b,c (or c,b) first, then a vs (bc): tree = Branch(x, y)."""
    data = O.huf_encode_all(O.SYM_CHAR, np.array([ord(c) for c in "abca"], np.uint32))
    # 2 branches + 3 leaves(1+1 byte each) = 8 bytes of trie, payload = 1+2+2+1 = 6 bits -> 1 byte
    assert len(data) == 2 + 3 * 2 + 1
    assert data[0] == 1
    assert O.huf_size(O.SYM_CHAR, np.array([2, 1, 1], np.uint64)) == len(data)


def test_huf_size_formula():
    """bytes = n(1+S) + (n-1) + ceil(sum f*len / 8)"""
    rng = np.random.default_rng(1)
    syms = rng.integers(0, 40, 5000).astype(np.uint32) * 65793  # grey RGB keys
    keys, counts = O.count_freqs(syms)
    data = O.huf_encode_all(O.SYM_RGB, syms)
    assert O.huf_size(O.SYM_RGB, counts) == len(data)
    lens, _ = O.huf_build(counts)
    n = len(keys)
    assert len(data) == n * 12 + (n - 1) + (int((counts * lens).sum()) + 7) // 8
    rc, out = O.huf_decode_all(O.SYM_RGB, data, syms.size)
    assert rc == 0 and np.array_equal(out, syms)


def test_decode_truncated_stream_fails():
    """huf.rs:198 EOF -> None"""
    syms = np.array([ord(c) for c in "abcabcaabbcc"], np.uint32)
    data = O.huf_encode_all(O.SYM_CHAR, syms)
    rc, _ = O.huf_decode_all(O.SYM_CHAR, data[:-1], len(syms))
    assert rc == O.DECODE
    rc, _ = O.huf_decode_all(O.SYM_CHAR, bytes([7]), 1)   # bad enum tag huf.rs:343-345
    assert rc == O.DECODE


def test_count_freqs():
    """utils.rs:4-16 (sorted by key: deviation D1)"""
    keys, counts = O.count_freqs(np.array([5, 3, 5, 5, 9, 3], np.uint32))
    assert list(keys) == [3, 5, 9] and list(counts) == [2, 3, 1]
    keys, counts = O.count_freqs(np.array([], np.uint32))
    assert keys.size == 0


# ---------------------------------------------------------------- src/kmeans.rs:446-581
def square(p):
    """kmeans.rs:506-514"""
    return [(p[0] + i, p[1] + j) for i in range(-1, 2) for j in range(-1, 2)]


@pytest.mark.parametrize("mode", [O.MODE_R, O.MODE_L])
def test_all_clusters(mode):
    """kmeans.rs:491-500"""
    data = [(0, 0), (1, 1)]
    rc, r = O.kmeans(O.PT_TOY2, mode, data, None, 2)
    assert rc == 0
    assert {tuple(c) for c in r["centroids"]} == set(data)
    for i, p in enumerate(data):
        assert tuple(r["centroids"][r["labels"][i]]) == p
    assert list(r["members"]) == [1, 1]


@pytest.mark.parametrize("mode", [O.MODE_R, O.MODE_L])
def test_square1(mode):
    """kmeans.rs:516-523"""
    rc, r = O.kmeans(O.PT_TOY2, mode, square((0, 0)), None, 1)
    assert rc == 0
    assert tuple(r["centroids"][0]) == (0, 0)
    assert r["members"][0] == 9


@pytest.mark.parametrize("mode", [O.MODE_R, O.MODE_L])
def test_squares2(mode):
    """kmeans.rs:525-539"""
    data = square((-100, 0)) + square((100, 0))
    rc, r = O.kmeans(O.PT_TOY2, mode, data, None, 2)
    assert rc == 0
    assert {tuple(c) for c in r["centroids"]} == {(-100, 0), (100, 0)}


def test_dist1_dist2():
    """kmeans.rs:541-557"""
    assert O.pt_dist(O.PT_TOY2, (0, 0), (0, 1)) == 1.0
    for p in square((-100, 0)):
        assert O.pt_dist(O.PT_TOY2, (-11, 0), p) < O.pt_dist(O.PT_TOY2, (11, 0), p)


def test_mean1():
    """kmeans.rs:559-563: mean of the square around (-100,0) is (-100,0) (truncating i64 division)"""
    pts = np.array(square((-100, 0)), np.int32)
    st = O.kmeans_step(O.PT_TOY2, pts, None, 1, [(0, 0)], np.zeros(9, np.uint32))
    cent, nres = O.kmeans_finalize(O.PT_TOY2, pts, 1, 0, 0, st["sums"], st["wsum"], st["members"])
    assert tuple(cent[0]) == (-100, 0) and nres == 0


def test_radii():
    """kmeans.rs:565-573: certainty radius = half the distance to the closest centroid"""
    rc, r = O.kmeans(O.PT_TOY2, O.MODE_R, [(0, 0), (1, 0)], None, 2)
    assert rc == 0
    assert list(r["radii"]) == [0.5, 0.5]


@pytest.mark.parametrize("mode", [O.MODE_R, O.MODE_L])
def test_proper_init_asg(mode):
    """kmeans.rs:575-580: must not trip the active-cluster assertion"""
    rc, _ = O.kmeans(O.PT_TOY2, mode, [(1000, 0), (1000, 1), (-1000, 0), (-1000, 1)], None, 3)
    assert rc == 0


def test_too_few_points():
    """kmeans.rs:67-68 assert!(points_per_cluster > 0)"""
    rc, _ = O.kmeans(O.PT_TOY2, O.MODE_R, [(0, 0)], None, 2)
    assert rc == O.TOO_FEW_POINTS


def test_init_assignment_chunks():
    """kmeans.rs:61-78: cluster i<K-1 = points[n-(i+1)ppc .. n-i*ppc], cluster K-1 = the head"""
    lab = O.init_labels(10, 3)  # ppc = 3
    assert list(lab) == [2, 2, 2, 2, 1, 1, 1, 0, 0, 0]


# ---------------------------------------------------------------- src/codec/clusterc.rs:299-338
def test_rgb_mean():
    """clusterc.rs:304-310"""
    pts = np.array([[0, 0, 0], [2, 2, 2]], np.int32)
    w = np.array([1, 1], np.uint32)
    st = O.kmeans_step(O.PT_RGBW, pts, w, 1, [[0, 0, 0]], np.zeros(2, np.uint32))
    cent, _ = O.kmeans_finalize(O.PT_RGBW, pts, 1, 0, 0, st["sums"], st["wsum"], st["members"])
    assert list(cent[0]) == [1, 1, 1]


def test_rgb_mean_weighted_truncates():
    """clusterc.rs:92-105: sum(c*count)/sum(count), u64 truncating"""
    pts = np.array([[0, 10, 255], [3, 0, 0]], np.int32)
    w = np.array([1, 2], np.uint32)
    st = O.kmeans_step(O.PT_RGBW, pts, w, 1, [[0, 0, 0]], np.zeros(2, np.uint32))
    cent, _ = O.kmeans_finalize(O.PT_RGBW, pts, 1, 0, 0, st["sums"], st["wsum"], st["members"])
    assert list(cent[0]) == [6 // 3, 10 // 3, 255 // 3]


def test_rgb_dist():
    """clusterc.rs:312-337 rgb_dist0..3"""
    assert O.pt_dist(O.PT_RGBW, (0, 10, 20), (0, 10, 20)) == 0.0
    assert O.pt_dist(O.PT_RGBW, (0, 0, 0), (1, 0, 0)) == 1.0
    assert O.pt_dist(O.PT_RGBW, (0, 0, 0), (1, 1, 0)) == math.sqrt(2.0)
    assert O.pt_dist(O.PT_RGBW, (0, 0, 0), (1, 1, 1)) == math.sqrt(3.0)


def test_colorpos_dist():
    """clusterc.rs:206-213: sqrt(dx^2 + dy^2 + dist(rgb)^2), wrapping u32 subtraction"""
    a, b = (3, 10, 1, 2, 3), (7, 2, 1, 2, 3)
    assert O.pt_dist(O.PT_XYRGB, a, b) == math.sqrt(16 + 64)
    assert O.pt_dist(O.PT_XYRGB, b, a) == math.sqrt(16 + 64)


# ---------------------------------------------------------------- README.md:150-175 delta example
def test_delta_readme_example():
    """README.md:166,173: 3 3 5 7 6 6 8 8 7 7 7 8 9 9 8 9 -> 3 0 2 2 -1 0 2 0 -1 0 0 1 1 0 -1 1
    (start value 0, hilbertc.rs:445)"""
    stream = [3, 3, 5, 7, 6, 6, 8, 8, 7, 7, 7, 8, 9, 9, 8, 9]
    expect = [3, 0, 2, 2, -1, 0, 2, 0, -1, 0, 0, 1, 1, 0, -1, 1]
    lin = np.repeat(np.array(stream, np.uint8)[:, None], 3, axis=1)
    d = O.unpack_signed(O.delta_diff(lin))
    assert d[:, 0].tolist() == expect and d[:, 1].tolist() == expect and d[:, 2].tolist() == expect
    rc, back = O.delta_undiff(O.delta_diff(lin))
    assert rc == 0 and np.array_equal(back, lin)


def test_readme_hilbert_sketch():
    """README.md:93-99: the 4x4 sketch (drawn with y upward) linearises the sample image to the
    stream of the delta example.  The frozen scan reproduces it."""
    img_rows_top_to_bottom = [[6, 8, 7, 7], [6, 8, 7, 8], [7, 5, 9, 9], [3, 3, 8, 9]]
    img = np.array(img_rows_top_to_bottom[::-1], np.uint8)  # y up -> row 0 = bottom row
    rgb = np.repeat(img[:, :, None], 3, axis=2)
    lin = O.hilbert_linearize(rgb)
    assert lin[:, 0].tolist() == [3, 3, 5, 7, 6, 6, 8, 8, 7, 7, 7, 8, 9, 9, 8, 9]
