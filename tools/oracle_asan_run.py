"""Runs the oracle's codecs, Huffman build and K-means on small inputs against liboracle_asan.so (AddressSanitizer + UBSan build of
oracle/*.c); started by tests/test_oracle_asan.py with libasan preloaded.  Sanitizers are CPU-only here (race detection /
sanitizers row)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_lib as O
O._lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_asan.so"))
O._lib.orc_mse.restype = C.c_double
O._lib.orc_pt_dist.restype = C.c_double
O._lib.orc_kmeans_reseed_index.restype = C.c_uint64
O._lib.orc_kmeans_reseed_index.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64]
O._lib.orc_kmeans_init_label.restype = C.c_uint32
O._lib.orc_kmeans_init_label.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
O._lib.orc_bit_mask.restype = C.c_uint8
import numpy as np
from cniic_amd import synth
img = synth.photo(96, 64, synth.SEED0 + 1)
for expr in ("hufman", "delta", "hilbert(rle)", "cluster-colors(16)", "voronoi(8)"):
    rc, data, st = O.encode(expr, img)
    rc2, back = O.decode(expr, data)
    assert rc == 0 and rc2 == 0
one = np.full((1, 1, 3), 9, np.uint8)
for expr in ("hufman", "delta"):
    rc, data, _ = O.encode(expr, one); assert rc == 0
    rc, back = O.decode(expr, data); assert rc == 0
rng = np.random.default_rng(0)
for n in (1, 2, 3, 17, 1000):
    c = rng.integers(1, 50, n).astype(np.uint64)
    O.huf_build(c)
print("asan ok")
