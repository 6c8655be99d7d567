"""An INDEPENDENT check of the Huffman trees: the oracle's build (oracle/huf.c) and the product's
(csrc/huff_host.cpp, the GPU sort + host merge + GPU codes path) follow the same two-queue merge rule, so comparing one
with the other proves nothing about optimality.  Here the yardstick is a plain binary heap (Python's heapq, the textbook
construction of src/huf.rs:58-117): ANY Huffman tree of a histogram has the same sum(count x length), and a full prefix
code has Kraft sum exactly 1.  Histograms: random, all-equal, Fibonacci-like (the deepest tree a given total allows),
counts >= 2^32, two symbols, heavy ties."""
import heapq
from fractions import Fraction

import numpy as np
import pytest

import oracle_lib as O


def heap_cost(counts):
    """sum over the internal nodes of their weight = sum(count x code length) of an optimal prefix code"""
    h = [int(c) for c in counts]
    if len(h) < 2:
        return 0
    heapq.heapify(h)
    total = 0
    while len(h) > 1:
        a = heapq.heappop(h)
        b = heapq.heappop(h)
        total += a + b
        heapq.heappush(h, a + b)
    return total


def histograms():
    rng = np.random.default_rng(20261004)
    yield "two", np.array([5, 9], np.uint64)
    yield "three_ties", np.array([1, 1, 1], np.uint64)
    yield "all_equal_257", np.full(257, 7, np.uint64)
    yield "all_equal_pow2", np.full(1024, 3, np.uint64)
    fib = [1, 1]
    while len(fib) < 60:
        fib.append(fib[-1] + fib[-2])
    yield "fibonacci_60", np.array(fib, np.uint64)            # code lengths up to 59 bits
    yield "fibonacci_shuffled", rng.permutation(np.array(fib[:48], np.uint64))
    yield "big_counts", (rng.integers(1, 1 << 20, 300).astype(np.uint64) << np.uint64(22)) + np.uint64(1 << 32)
    yield "random_small", rng.integers(1, 50, 5000).astype(np.uint64)
    yield "random_wide", rng.integers(1, 1 << 40, 3000).astype(np.uint64)
    yield "geometric", np.maximum(1, (1e9 * 0.97 ** np.arange(600))).astype(np.uint64)
    z = rng.zipf(1.3, 40000)
    yield "zipf_ties", np.bincount(np.minimum(z, 20000))[1:].astype(np.uint64)[np.bincount(np.minimum(z, 20000))[1:] > 0]


@pytest.mark.parametrize("name,counts", list(histograms()), ids=[n for n, _ in histograms()])
def test_oracle_tree_is_an_optimal_full_prefix_code(name, counts):
    lens, codes = O.huf_build(counts)
    assert int((counts.astype(object) * lens.astype(object)).sum()) == heap_cost(counts), name
    assert sum(Fraction(1, 1 << int(l)) for l in lens) == 1                 # Kraft: the code is full
    # prefix-free: sorted left-aligned, no code is a prefix of its successor (codes of up to 64 bits)
    if int(lens.max()) <= 64:
        items = sorted((int(c) << (64 - int(l)), int(l)) for c, l in zip(codes, lens))
        for (a, la), (b, lb) in zip(items, items[1:]):
            assert (a >> (64 - min(la, lb))) != (b >> (64 - min(la, lb))), name


@pytest.mark.parametrize("name,counts", list(histograms()), ids=[n for n, _ in histograms()])
def test_stream_size_is_the_heap_cost(name, counts):
    """orc_huf_size (what every GPU test compares stream lengths with) = header + leaves + branches + ceil(heap cost / 8)"""
    n = counts.size
    for kind, S in ((O.SYM_RGB, 11), (O.SYM_SIGNED, 6)):
        assert O.huf_size(kind, counts) == n * (1 + S) + (n - 1) + (heap_cost(counts) + 7) // 8


def test_round_trip_through_the_reference_faithful_decoder():
    """encode_all -> decode_all (the trie walk of huf.rs:187-206) on a tie-heavy stream"""
    rng = np.random.default_rng(7)
    syms = rng.integers(0, 300, 20000).astype(np.uint32)
    syms[::3] = 5
    data = O.huf_encode_all(O.SYM_RGB, syms)
    rc, back = O.huf_decode_all(O.SYM_RGB, data, syms.size)
    assert rc == 0 and np.array_equal(back, syms)
    k, c = O.count_freqs(syms)
    assert len(data) == O.huf_size(O.SYM_RGB, c)


@pytest.mark.parametrize("name,counts", list(histograms()), ids=[n for n, _ in histograms()])
def test_product_host_builder_is_optimal(name, counts):
    """cniic_huf_size is a pure host function of the histogram (csrc/huff_host.cpp: radix sort + two-queue merge):
    the same yardstick, without a GPU"""
    import ctypes as C
    import cniic_amd
    L = cniic_amd.lib()
    n = counts.size
    c = np.ascontiguousarray(counts, np.uint64)
    for kind, S in ((1, 11), (2, 6)):
        nb = C.c_uint64(0)
        rc = L.cniic_huf_size(C.c_int32(kind), c.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.byref(nb))
        assert rc == 0 and nb.value == n * (1 + S) + (n - 1) + (heap_cost(counts) + 7) // 8, name


@pytest.mark.gpu
@pytest.mark.parametrize("runs", [False, True])
@pytest.mark.parametrize("distinct,n", [(300, 50000), (40000, 400000), (150000, 600000)])
def test_gpu_encode_all_is_optimal_and_decodes_with_the_reference_faithful_decoder(distinct, n, runs, monkeypatch):
    """huf::encode_all on the GPU -- small alphabets (host tree) and >= 32768 distinct symbols (GPU (count, key) radix sort,
    host merge -- `runs`: of runs of equally frequent leaves, expanded on the GPU --, GPU codes + decoder): stream length = the
    heap's cost, and the oracle's trie-walk decoder reads it back"""
    import cniic_amd
    if runs:
        monkeypatch.setenv("CNIIC_HUF_RUNS_MIN", "0")
    rng = np.random.default_rng(distinct)
    z = np.minimum(rng.zipf(1.2, n), distinct).astype(np.uint32)           # heavy ties among the rare symbols
    syms = (z * np.uint32(2654435761)) & np.uint32(0xFFFFFF)               # spread over the 24-bit key space
    k, c = O.count_freqs(syms)
    with cniic_amd.Context(0) as ctx:
        data = ctx.huf_encode_all(O.SYM_RGB, syms)
    U = k.size
    assert len(data) == U * 12 + (U - 1) + (heap_cost(c) + 7) // 8
    rc, back = O.huf_decode_all(O.SYM_RGB, data, syms.size)
    assert rc == 0 and np.array_equal(back, syms)
