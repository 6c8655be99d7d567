"""oracle/kmeans_fast.c (threads + vectorised arg-min) against the plain orc_kmeans_step, bit for bit.

The fast step exists to produce tests/golden/fullsize_digests.json (mode L at the BASELINE sizes); its
authority is this file: labels, u64 sums, weights, member counts and the changed count equal the plain
loop's on inputs made of ties (tiny coordinate ranges, duplicated centroids), for every thread count."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O


def _step_fast(kind, pts, weight, K, cent, labels, threads):
    D = 3 if kind == O.PT_RGBW else 5
    pts = np.ascontiguousarray(pts, np.int32).reshape(-1, D)
    n = pts.shape[0]
    w = None if weight is None else np.ascontiguousarray(weight, np.uint32)
    cent = np.ascontiguousarray(cent, np.int32).reshape(K, D)
    labels = np.array(labels, dtype=np.uint32, copy=True)
    sums = np.zeros((K, D), np.uint64)
    wsum = np.zeros(K, np.uint64)
    members = np.zeros(K, np.uint64)
    ch = C.c_uint64(0)
    rc = O.lib().orc_kmeans_step_fast(kind, O._p(pts), O._p(w), C.c_uint64(n), C.c_uint32(K), O._p(cent), O._p(labels),
                                      O._p(sums), O._p(wsum), O._p(members), C.byref(ch), C.c_int(threads))
    assert rc == 0, rc
    return dict(labels=labels, sums=sums, wsum=wsum, members=members, changed=ch.value)


def _case(kind, n, K, span, seed):
    rng = np.random.default_rng(seed)
    if kind == O.PT_RGBW:
        pts = rng.integers(0, span, (n, 3), dtype=np.int32)
        weight = rng.integers(1, 1 << 20, n, dtype=np.uint32)
        cent = rng.integers(0, span, (K, 3), dtype=np.int32)
    else:
        pts = np.concatenate([rng.integers(0, max(span, 2) * 40, (n, 2), dtype=np.int32),
                              rng.integers(0, span, (n, 3), dtype=np.int32)], axis=1)
        weight = None
        cent = np.concatenate([rng.integers(0, max(span, 2) * 40, (K, 2), dtype=np.int32),
                               rng.integers(0, span, (K, 3), dtype=np.int32)], axis=1)
    if K > 3:
        cent[K // 2] = cent[1]      # duplicated centroids: the lowest id must win, the current one must keep its point
        cent[K - 1] = cent[0]
    labels = rng.integers(0, K, n, dtype=np.uint32)
    return pts, weight, cent, labels


@pytest.mark.parametrize("kind", [O.PT_RGBW, O.PT_XYRGB])
@pytest.mark.parametrize("n,K,span", [(1, 1, 4), (7, 3, 2), (1000, 16, 3), (4099, 17, 6), (3000, 256, 256), (2500, 2048, 8),
                                      (777, 4096, 256)])
def test_fast_step_equals_plain_step(kind, n, K, span):
    pts, weight, cent, labels = _case(kind, n, K, span, 11 * n + K)
    want = O.kmeans_step(kind, pts, weight, K, cent, labels)
    assert O.lib().orc_kmeans_fast_ok(kind, O._p(pts), C.c_uint64(n), C.c_uint32(K), O._p(cent)) == 1
    for threads in (1, 2, 3, 8):
        got = _step_fast(kind, pts, weight, K, cent, labels, threads)
        for f in ("labels", "sums", "wsum", "members"):
            assert np.array_equal(got[f], want[f]), (f, threads)
        assert got["changed"] == want["changed"]


def test_fast_step_falls_back_outside_its_limits():
    # a coordinate of 16384 and more: squared distances may leave 31 bits, the plain loop must run (and does: same answer)
    pts, weight, cent, labels = _case(O.PT_XYRGB, 500, 8, 4, 5)
    pts[3, 0] = 70000
    assert O.lib().orc_kmeans_fast_ok(O.PT_XYRGB, O._p(pts), C.c_uint64(500), C.c_uint32(8), O._p(cent)) == 0
    want = O.kmeans_step(O.PT_XYRGB, pts, None, 8, cent, labels)
    got = _step_fast(O.PT_XYRGB, pts, None, 8, cent, labels, 4)
    assert np.array_equal(got["labels"], want["labels"]) and np.array_equal(got["sums"], want["sums"])


@pytest.mark.parametrize("expr,shape", [("cluster-colors(16)", (70, 90)), ("voronoi(12)", (48, 64)), ("cluster-colors(256)", (128, 160))])
def test_codecs_through_the_fast_step_give_the_same_stream(expr, shape):
    from cniic_amd import synth
    img = synth.photo(shape[1], shape[0], synth.SEED0 + 3)
    rc0, want, st0 = O.encode(expr, img, mode=O.MODE_L)
    O.lib().orc_set_lloyd_threads(4)
    try:
        rc1, got, st1 = O.encode(expr, img, mode=O.MODE_L)
    finally:
        O.lib().orc_set_lloyd_threads(0)
    assert rc0 == rc1 == 0 and got == want and st0["iterations"] == st1["iterations"]
