"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference hot path.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (cniic_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

OK, BAD_ARG, TOO_FEW_POINTS, FEW_ACTIVE, DECODE, NOMEM, CAPACITY = 0, -1, -2, -3, -6, -7, -8
SYM_CHAR, SYM_RGB, SYM_SIGNED = 0, 1, 2
PT_TOY2, PT_RGBW, PT_XYRGB = 0, 1, 2
MODE_R, MODE_L = 0, 1
DEFAULT_SEED = 0x636E696963


class Buf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t)]


class Rd(C.Structure):
    _fields_ = [("p", C.c_void_p), ("n", C.c_size_t), ("pos", C.c_size_t)]


class BitW(C.Structure):
    _fields_ = [("out", C.POINTER(Buf)), ("curr_bits", C.c_uint8), ("bit_count", C.c_uint8)]


class KmStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("iterations", "moved_last", "obvious_stay", "neighbour_cutoff",
                                          "tested_neighbours", "empty_reseeds", "dist_evals")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(asan=False):
    target = "liboracle_asan.so" if asan else "liboracle.so"
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, target])
    return os.path.join(ORACLE_DIR, target)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            build()
        _lib = C.CDLL(path)
        _lib.orc_mse.restype = C.c_double
        _lib.orc_pt_dist.restype = C.c_double
        _lib.orc_kmeans_reseed_index.restype = C.c_uint64
        _lib.orc_kmeans_reseed_index.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64]
        _lib.orc_kmeans_init_label.restype = C.c_uint32
        _lib.orc_kmeans_init_label.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        _lib.orc_bit_mask.restype = C.c_uint8
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


# ---------------- bit / ser ----------------
class BitWriter:
    """IoBitWriter<Vec<u8>> (bit.rs:186-254)."""

    def __init__(self):
        self.buf = Buf()
        lib().orc_buf_init(C.byref(self.buf))
        self.w = BitW()
        lib().orc_bitw_init(C.byref(self.w), C.byref(self.buf))

    def write(self, bit):
        assert lib().orc_bitw_bit(C.byref(self.w), int(bit)) == 0

    def write_byte(self, n):
        assert lib().orc_bitw_byte(C.byref(self.w), C.c_uint8(n)) == 0

    def write_code(self, bits):
        """BitArray::from_slice + write_arr (bit.rs:123-149,164-178)."""
        nfull = len(bits) // 8
        full = bytes(int("".join(str(b) for b in bits[8 * i:8 * i + 8]), 2) for i in range(nfull))
        rest = bits[8 * nfull:]
        partial = int("".join(str(b) for b in rest), 2) if rest else 0
        fb = (C.c_uint8 * max(1, nfull)).from_buffer_copy(full.ljust(max(1, nfull), b"\0"))
        assert lib().orc_bitw_code(C.byref(self.w), fb, C.c_size_t(nfull), C.c_uint8(partial), C.c_uint8(len(rest))) == 0

    def pad_and_flush(self):
        assert lib().orc_bitw_pad_and_flush(C.byref(self.w)) == 0

    def into_inner(self):
        out = bytes(C.string_at(self.buf.data, self.buf.len)) if self.buf.len else b""
        lib().orc_buf_free(C.byref(self.buf))
        return out


def bit_mask(n):
    return int(lib().orc_bit_mask(C.c_uint8(n)))


def bit_nth(byte, idx, msb_first=True):
    return int(lib().orc_bit_nth(C.c_uint8(byte), C.c_uint8(idx), int(msb_first)))


def ser(kind, value):
    b = Buf()
    L = lib()
    L.orc_buf_init(C.byref(b))
    if kind == "u8":
        L.orc_ser_u8(C.byref(b), C.c_uint8(value))
    elif kind == "u16":
        L.orc_ser_u16(C.byref(b), C.c_uint16(value))
    elif kind == "i16":
        L.orc_ser_i16(C.byref(b), C.c_int16(value))
    elif kind == "u32":
        L.orc_ser_u32(C.byref(b), C.c_uint32(value))
    elif kind == "u64":
        L.orc_ser_u64(C.byref(b), C.c_uint64(value))
    elif kind == "rgb":
        L.orc_ser_rgb(C.byref(b), (C.c_uint8 * 3)(*value))
    else:
        raise ValueError(kind)
    out = bytes(C.string_at(b.data, b.len))
    L.orc_buf_free(C.byref(b))
    return out


# ---------------- histogram / huffman ----------------
def count_freqs(syms):
    syms = np.ascontiguousarray(syms, dtype=np.uint32)
    n = syms.size
    keys = np.empty(max(n, 1), np.uint32)
    counts = np.empty(max(n, 1), np.uint64)
    nu = C.c_uint64(0)
    rc = lib().orc_count_freqs(_p(syms), C.c_uint64(n), _p(keys), _p(counts), C.c_uint64(max(n, 1)), C.byref(nu))
    assert rc == 0, rc
    return keys[:nu.value].copy(), counts[:nu.value].copy()


def huf_build(counts):
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    n = counts.size
    lens = np.empty(n, np.uint32)
    codes = np.empty(n, np.uint64)
    rc = lib().orc_huf_build(_p(counts), C.c_uint64(n), _p(lens), _p(codes))
    assert rc == 0, rc
    return lens, codes


def huf_size(kind, counts):
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    nb = C.c_uint64(0)
    rc = lib().orc_huf_size(kind, _p(counts), C.c_uint64(counts.size), C.byref(nb))
    assert rc == 0, rc
    return nb.value


def huf_encode_all(kind, syms):
    syms = np.ascontiguousarray(syms, dtype=np.uint32)
    b = Buf()
    lib().orc_buf_init(C.byref(b))
    rc = lib().orc_huf_encode_all(kind, _p(syms), C.c_uint64(syms.size), C.byref(b))
    out = bytes(C.string_at(b.data, b.len)) if b.len else b""
    lib().orc_buf_free(C.byref(b))
    assert rc == 0, rc
    return out


def huf_decode_all(kind, data, nsyms):
    raw = np.frombuffer(data, dtype=np.uint8)
    r = Rd(raw.ctypes.data, raw.size, 0)
    syms = np.empty(nsyms, np.uint32)
    rc = lib().orc_huf_decode_all(kind, C.byref(r), _p(syms), C.c_uint64(nsyms))
    return rc, syms


# ---------------- k-means ----------------
_DIM = {PT_TOY2: 2, PT_RGBW: 3, PT_XYRGB: 5}


def kmeans(kind, mode, pts, weight, K, seed=DEFAULT_SEED, max_iters=0):
    D = _DIM[kind]
    pts = np.ascontiguousarray(pts, dtype=np.int32).reshape(-1, D)
    n = pts.shape[0]
    w = None if weight is None else np.ascontiguousarray(weight, dtype=np.uint32)
    cent = np.zeros((K, D), np.int32)
    labels = np.zeros(n, np.uint32)
    members = np.zeros(K, np.uint64)
    radii = np.zeros(K, np.float64)
    st = KmStats()
    rc = lib().orc_kmeans(kind, mode, _p(pts), _p(w), C.c_uint64(n), C.c_uint32(K), C.c_uint64(seed),
                          C.c_uint64(max_iters), _p(cent), _p(labels), _p(members), _p(radii), C.byref(st))
    return rc, dict(centroids=cent, labels=labels, members=members, radii=radii, stats=st.as_dict())


def kmeans_step(kind, pts, weight, K, centroids, labels):
    D = _DIM[kind]
    pts = np.ascontiguousarray(pts, dtype=np.int32).reshape(-1, D)
    n = pts.shape[0]
    w = None if weight is None else np.ascontiguousarray(weight, dtype=np.uint32)
    cent = np.ascontiguousarray(centroids, dtype=np.int32).reshape(K, D)
    labels = np.array(labels, dtype=np.uint32, copy=True)
    sums = np.zeros((K, D), np.uint64)
    wsum = np.zeros(K, np.uint64)
    members = np.zeros(K, np.uint64)
    ch = C.c_uint64(0)
    rc = lib().orc_kmeans_step(kind, _p(pts), _p(w), C.c_uint64(n), C.c_uint32(K), _p(cent), _p(labels),
                               _p(sums), _p(wsum), _p(members), C.byref(ch))
    assert rc == 0, rc
    return dict(labels=labels, sums=sums, wsum=wsum, members=members, changed=ch.value)


def kmeans_finalize(kind, pts, K, seed, it, sums, wsum, members):
    D = _DIM[kind]
    pts = np.ascontiguousarray(pts, dtype=np.int32).reshape(-1, D)
    cent = np.zeros((K, D), np.int32)
    nres = C.c_uint64(0)
    rc = lib().orc_kmeans_finalize(kind, _p(pts), C.c_uint64(pts.shape[0]), C.c_uint32(K), C.c_uint64(seed),
                                   C.c_uint64(it), _p(np.ascontiguousarray(sums, np.uint64)),
                                   _p(np.ascontiguousarray(wsum, np.uint64)),
                                   _p(np.ascontiguousarray(members, np.uint64)), _p(cent), C.byref(nres))
    assert rc == 0, rc
    return cent, nres.value


def init_labels(n, K):
    return np.array([lib().orc_kmeans_init_label(i, n, K) for i in range(n)], np.uint32)


def reseed_index(seed, it, c, n):
    return int(lib().orc_kmeans_reseed_index(seed, it, c, n))


def pt_dist(kind, a, b):
    a = np.ascontiguousarray(a, np.int32)
    b = np.ascontiguousarray(b, np.int32)
    return float(lib().orc_pt_dist(kind, _p(a), _p(b)))


# ---------------- hilbert / delta ----------------
def hilbert_iter(w, h):
    xy = np.zeros((max(w * h, 1), 2), np.uint32)
    rc = lib().orc_hilbert_iter(C.c_uint32(w), C.c_uint32(h), _p(xy))
    assert rc == 0, rc
    return xy[:w * h]


def hilbert_d2xy(w, h, d):
    x, y = C.c_uint32(0), C.c_uint32(0)
    lib().orc_hilbert_d2xy(C.c_uint32(w), C.c_uint32(h), C.c_uint64(d), C.byref(x), C.byref(y))
    return x.value, y.value


def hilbert_linearize(img):
    img = _u8(img)
    h, w = img.shape[:2]
    out = np.empty((h * w, 3), np.uint8)
    rc = lib().orc_hilbert_linearize(_p(img), C.c_uint32(w), C.c_uint32(h), _p(out))
    assert rc == 0, rc
    return out


def delta_diff(lin):
    lin = _u8(lin).reshape(-1, 3)
    s = np.empty(lin.shape[0], np.uint32)
    assert lib().orc_delta_diff(_p(lin), C.c_uint64(lin.shape[0]), _p(s)) == 0
    return s


def delta_undiff(syms):
    syms = np.ascontiguousarray(syms, np.uint32)
    out = np.empty((syms.size, 3), np.uint8)
    rc = lib().orc_delta_undiff(_p(syms), C.c_uint64(syms.size), _p(out))
    return rc, out


def unpack_signed(keys):
    keys = np.asarray(keys, np.uint32)
    return np.stack([((keys >> s) & 511).astype(np.int32) - 255 for s in (18, 9, 0)], axis=-1)


# ---------------- codecs ----------------
def encode(codec, img, mode=MODE_L, seed=DEFAULT_SEED):
    """Codec::encode (codec.rs:14-19) -> (rc, bytes, kmeans stats)."""
    img = _u8(img)
    h, w = img.shape[:2]
    cap = 64 + w * h * 16 + (1 << 16)
    out = np.empty(cap, np.uint8)
    ln = C.c_uint64(0)
    st = KmStats()
    rc = lib().orc_encode(codec.encode(), mode, C.c_uint64(seed), _p(img), C.c_uint32(w), C.c_uint32(h),
                          _p(out), C.c_uint64(cap), C.byref(ln), C.byref(st))
    return rc, (out[:ln.value].tobytes() if rc == 0 else b""), st.as_dict()


def decode(codec, data, max_px=1 << 26):
    """Codec::decode -> (rc, HxWx3 uint8 or None)."""
    raw = np.frombuffer(data, dtype=np.uint8)
    # peek dims to size the output
    if raw.size < 8:
        return DECODE, None
    w = int.from_bytes(raw[0:4].tobytes(), "little")
    h = int.from_bytes(raw[4:8].tobytes(), "little")
    if w * h > max_px:
        return CAPACITY, None
    out = np.zeros((max(h * w, 1), 3), np.uint8)
    cw, ch = C.c_uint32(0), C.c_uint32(0)
    rc = lib().orc_decode(codec.encode(), _p(raw), C.c_uint64(raw.size), _p(out), C.c_uint64(out.size),
                          C.byref(cw), C.byref(ch))
    if rc != 0:
        return rc, None
    return rc, out[:h * w].reshape(h, w, 3)


def mse(a, b):
    a, b = _u8(a), _u8(b)
    return float(lib().orc_mse(_p(a), _p(b), C.c_uint64(a.size // 3)))
