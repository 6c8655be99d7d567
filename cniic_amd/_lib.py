"""ctypes loader + thin wrappers of libcniic_hip.so (the C ABI of include/cniic_hip.h).

There is NO fallback: if the HIP extension is missing or no GPU is usable the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

OK = 0
ERR = {-1: "BAD_ARG", -2: "TOO_FEW_POINTS", -3: "FEW_ACTIVE", -4: "HIP", -5: "RCCL", -6: "DECODE", -7: "NOMEM",
       -8: "CAPACITY", -9: "UNSUPPORTED"}
BAD_ARG, TOO_FEW_POINTS, FEW_ACTIVE, HIP, RCCL, DECODE, NOMEM, CAPACITY, UNSUPPORTED = range(-1, -10, -1)
SYM_RGB, SYM_SIGNED = 1, 2
SYNTH_UNIFORM, SYNTH_PHOTO = 0, 1
KM_BRUTE_FORCE, KM_PROFILE, KM_NO_SKIP = 1, 2, 4
OPT_SP_MIN_PIXELS, OPT_HUF_GPU_CODES_MIN, OPT_GPU_DECODE_MIN, OPT_DELTA_ROUTE, OPT_STAGE_TIMERS, OPT_FRAME_TREES_HOST, OPT_BATCH_STREAMS, OPT_KM_MAX_BLOCKS, OPT_KM_LOOP = range(1, 10)

# every symbol include/cniic_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "cniic_ctx_create", "cniic_ctx_destroy", "cniic_last_error", "cniic_version", "cniic_is_testing_build", "cniic_sync", "cniic_dev_alloc",
    "cniic_dev_free", "cniic_memcpy", "cniic_ctx_set_opt", "cniic_ctx_unset_opt", "cniic_ctx_get_opt", "cniic_ctx_set_scan", "cniic_last_kernel_time", "cniic_hist_rgb24", "cniic_hist_syms",
    "cniic_kmeans_rgbw", "cniic_kmeans_xyrgb", "cniic_kmeans_step_rgbw", "cniic_kmeans_step_xyrgb",
    "cniic_km_create_rgbw", "cniic_km_partial_words", "cniic_km_partials", "cniic_km_begin",
    "cniic_km_labels_internal", "cniic_km_assign", "cniic_km_update",
    "cniic_km_result", "cniic_km_time_assign", "cniic_km_destroy", "cniic_hist_rgb24_dense", "cniic_cc_create",
    "cniic_cc_unique", "cniic_cc_label_bytes", "cniic_cc_partials", "cniic_cc_assign", "cniic_cc_update", "cniic_cc_poll", "cniic_cc_poll_lagged",
    "cniic_cc_export_labels", "cniic_cc_import_labels", "cniic_cc_finish", "cniic_cc_finish_frames", "cniic_cc_destroy", "cniic_comm_unique_id",
    "cniic_comm_create", "cniic_comm_create_host", "cniic_comm_create_mailbox", "cniic_comm_connect_mailbox", "cniic_comm_destroy", "cniic_comm_all_reduce", "cniic_comm_set_timeout", "cniic_cc_run", "cniic_occupancy_pack",
    "cniic_cc_create_local", "cniic_cc_image_begin", "cniic_cc_image_occupancy", "cniic_cc_image_create", "cniic_remap_rgb", "cniic_hilbert_xy",
    "cniic_hilbert_linearize", "cniic_hilbert_delta", "cniic_hilbert_delta_hist", "cniic_huf_encode_all",
    "cniic_huf_size", "cniic_codec_parse", "cniic_codec_name", "cniic_codec_is_lossless", "cniic_codec_encode",
    "cniic_codec_encode_opts", "cniic_codec_encode_batch", "cniic_codec_decode", "cniic_mse", "cniic_synth_image",
]


class CniicError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("cniic error %d (%s): %s" % (code, ERR.get(code, "?"), msg))


class KmOpts(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("max_iters", C.c_uint64), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class KmStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("iterations", "moved_last", "empty_reseeds", "active", "pair_evals")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


COLORPOS = np.dtype([("x", "<u4"), ("y", "<u4"), ("rgb", "u1", (3,)), ("pad", "u1")])


def lib_path():
    """libcniic_hip.so -- or, with CNIIC_USE_TESTING_LIB=1 (tests/conftest.py, tools/), libcniic_hip_testing.so: the same code built with
    -DCNIIC_TESTING, the only build in which the CNIIC_TEST_* / CNIIC_DBG_* / route-forcing environment knobs exist"""
    name = "libcniic_hip_testing.so" if os.environ.get("CNIIC_USE_TESTING_LIB") == "1" else "libcniic_hip.so"
    if os.environ.get("CNIIC_LIB_FILE"):   # a measuring build made by a tool under tools/ (e.g. -DCNIIC_PS_PHASES), never a product path
        name = os.environ["CNIIC_LIB_FILE"]
    return os.path.join(_HERE, name)


_lib = None


HOST_SUM_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32)   # cniic_host_sum_fn


def lib():
    """Load libcniic_hip.so; raise loudly if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C cniic_amd/csrc` (hipcc --offload-arch=gfx950)" % p)
        L = C.CDLL(p)
        L.cniic_last_error.restype = C.c_char_p
        L.cniic_last_error.argtypes = [C.c_void_p]
        L.cniic_km_partial_words.restype = C.c_uint64
        L.cniic_km_partial_words.argtypes = [C.c_uint32, C.c_uint32]
        L.cniic_ctx_destroy.restype = None
        L.cniic_ctx_destroy.argtypes = [C.c_void_p]
        L.cniic_km_destroy.restype = None
        L.cniic_km_destroy.argtypes = [C.c_void_p]
        L.cniic_cc_destroy.restype = None
        L.cniic_cc_destroy.argtypes = [C.c_void_p]
        L.cniic_comm_destroy.restype = None
        L.cniic_comm_destroy.argtypes = [C.c_void_p]
        L.cniic_cc_unique.restype = C.c_uint64
        L.cniic_cc_unique.argtypes = [C.c_void_p]
        L.cniic_cc_label_bytes.restype = C.c_uint32
        L.cniic_cc_label_bytes.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _ptr(x):
    """numpy array (host) / torch tensor (device or host) / int address / None -> c_void_p"""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return C.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous()
        return C.c_void_p(x.data_ptr())
    if isinstance(x, (bytes, bytearray)):
        return C.cast(C.c_char_p(bytes(x)), C.c_void_p)
    return C.c_void_p(int(x))


class Context:
    """One cniic_ctx: a HIP stream + scratch HBM on one GPU.  Not shared between threads."""

    def __init__(self, device=0, stream=None):
        """stream: a hipStream_t handle to enqueue on (e.g. torch.cuda.current_stream().cuda_stream of a
        NON-default torch stream), or None for a private stream.  The NULL/default stream has handle 0
        and cannot be shared this way: make a torch.cuda.Stream() current first."""
        if stream is not None and int(stream) == 0:
            raise ValueError("cannot share the default (NULL) stream: use torch.cuda.set_stream(torch.cuda.Stream()) first")
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.cniic_ctx_create(C.c_int32(device), C.c_void_p(stream or 0), C.byref(h))
        if rc != OK:
            raise CniicError(rc, "cniic_ctx_create(device=%d) failed: no usable gfx950 device?" % device)
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self._L.cniic_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, allow=()):
        if rc != OK and rc not in allow:
            raise CniicError(rc, (self._L.cniic_last_error(self.h) or b"").decode())
        return rc

    def set_opt(self, opt, value):
        """cniic_ctx_set_opt: a route switch / threshold for this context (value None: back to the environment's / the default)"""
        if value is None:
            self._check(self._L.cniic_ctx_unset_opt(self.h, C.c_int32(opt)))
        else:
            self._check(self._L.cniic_ctx_set_opt(self.h, C.c_int32(opt), C.c_uint64(value)))

    def set_scan(self, w, h, xy):
        """cniic_ctx_set_scan: inject the scan of w x h images (xy: (w h, 2) uint32 positions, or None for the built-in one)"""
        if xy is not None and isinstance(xy, np.ndarray):
            xy = np.ascontiguousarray(xy, np.uint32)
        self._check(self._L.cniic_ctx_set_scan(self.h, C.c_uint32(w), C.c_uint32(h), _ptr(xy)))

    def get_opt(self, opt):
        v = C.c_uint64(0)
        self._check(self._L.cniic_ctx_get_opt(self.h, C.c_int32(opt), C.byref(v)))
        return v.value

    def sync(self):
        self._check(self._L.cniic_sync(self.h))

    def kernel_time(self, which):
        ms, n = C.c_double(0), C.c_uint64(0)
        rc = self._L.cniic_last_kernel_time(self.h, which.encode(), C.byref(ms), C.byref(n))
        return (ms.value, n.value) if rc == OK else (0.0, 0)

    # ---- H1
    def hist_rgb24(self, rgb, npx=None):
        npx = int(npx if npx is not None else rgb.size // 3) if not hasattr(rgb, "numel") else int(npx or rgb.numel() // 3)
        nu = C.c_uint64(0)
        self._check(self._L.cniic_hist_rgb24(self.h, _ptr(rgb), C.c_uint64(npx), None, None, C.c_uint64(0), C.byref(nu)))
        keys = np.empty(max(nu.value, 1), np.uint32)
        counts = np.empty(max(nu.value, 1), np.uint64)
        self._check(self._L.cniic_hist_rgb24(self.h, _ptr(rgb), C.c_uint64(npx), _ptr(keys), _ptr(counts),
                                             C.c_uint64(keys.size), C.byref(nu)))
        return keys[:nu.value], counts[:nu.value]

    def hist_syms(self, kind, syms):
        syms = np.ascontiguousarray(syms, np.uint32)
        nu = C.c_uint64(0)
        self._check(self._L.cniic_hist_syms(self.h, kind, _ptr(syms), C.c_uint64(syms.size), None, None, C.c_uint64(0), C.byref(nu)))
        keys = np.empty(max(nu.value, 1), np.uint32)
        counts = np.empty(max(nu.value, 1), np.uint64)
        self._check(self._L.cniic_hist_syms(self.h, kind, _ptr(syms), C.c_uint64(syms.size), _ptr(keys), _ptr(counts),
                                            C.c_uint64(keys.size), C.byref(nu)))
        return keys[:nu.value], counts[:nu.value]

    # ---- K-means
    @staticmethod
    def _opts(seed=0, max_iters=0, flags=0):
        return KmOpts(seed, max_iters, flags, 0)

    def kmeans_rgbw(self, keys, weight, K, seed=0, max_iters=0, flags=0, allow=()):
        keys = np.ascontiguousarray(keys, np.uint32)
        weight = np.ascontiguousarray(weight, np.uint32)
        U = keys.size
        cent = np.zeros((K, 3), np.uint8)
        labels = np.zeros(U, np.uint32)
        members = np.zeros(K, np.uint64)
        st = KmStats()
        o = self._opts(seed, max_iters, flags)
        rc = self._check(self._L.cniic_kmeans_rgbw(self.h, _ptr(keys), _ptr(weight), C.c_uint64(U), C.c_uint32(K), C.byref(o),
                                                   _ptr(cent), _ptr(labels), _ptr(members), C.byref(st)), allow)
        return rc, dict(centroids=cent, labels=labels, members=members, stats=st.as_dict())

    def kmeans_step_rgbw(self, keys, weight, K, centroids, labels):
        keys = np.ascontiguousarray(keys, np.uint32)
        weight = np.ascontiguousarray(weight, np.uint32)
        cent = np.ascontiguousarray(centroids, np.uint8).reshape(K, 3)
        labels = np.array(labels, np.uint32, copy=True)
        sums = np.zeros((K, 3), np.uint64)
        wsum = np.zeros(K, np.uint64)
        members = np.zeros(K, np.uint64)
        ch = C.c_uint64(0)
        self._check(self._L.cniic_kmeans_step_rgbw(self.h, _ptr(keys), _ptr(weight), C.c_uint64(keys.size), C.c_uint32(K),
                                                   _ptr(cent), _ptr(labels), _ptr(sums), _ptr(wsum), _ptr(members), C.byref(ch)))
        return dict(labels=labels, sums=sums, wsum=wsum, members=members, changed=ch.value)

    def kmeans_xyrgb(self, img, K, seed=0, max_iters=0, flags=0, want_labels=True, allow=()):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        cent = np.zeros(K, COLORPOS)
        labels = np.zeros(h * w, np.uint32) if want_labels else None
        members = np.zeros(K, np.uint64)
        st = KmStats()
        o = self._opts(seed, max_iters, flags)
        rc = self._check(self._L.cniic_kmeans_xyrgb(self.h, _ptr(img), C.c_uint32(w), C.c_uint32(h), C.c_uint32(K), C.byref(o),
                                                    _ptr(cent), _ptr(labels), _ptr(members), C.byref(st)), allow)
        return rc, dict(centroids=cent, labels=labels, members=members, stats=st.as_dict())

    def kmeans_step_xyrgb(self, img, K, centroids, labels):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        cent = np.ascontiguousarray(centroids, COLORPOS)
        labels = np.array(labels, np.uint32, copy=True)
        sums = np.zeros((K, 5), np.uint64)
        wsum = np.zeros(K, np.uint64)
        members = np.zeros(K, np.uint64)
        ch = C.c_uint64(0)
        self._check(self._L.cniic_kmeans_step_xyrgb(self.h, _ptr(img), C.c_uint32(w), C.c_uint32(h), C.c_uint32(K), _ptr(cent),
                                                    _ptr(labels), _ptr(sums), _ptr(wsum), _ptr(members), C.byref(ch)))
        return dict(labels=labels, sums=sums, wsum=wsum, members=members, changed=ch.value)

    def remap_rgb(self, img, keys, labels, centroids):
        img = np.ascontiguousarray(img, np.uint8)
        keys = np.ascontiguousarray(keys, np.uint32)
        labels = np.ascontiguousarray(labels, np.uint32)
        cent = np.ascontiguousarray(centroids, np.uint8)
        out = np.empty_like(img)
        self._check(self._L.cniic_remap_rgb(self.h, _ptr(img), C.c_uint64(img.size // 3), _ptr(keys), _ptr(labels),
                                            C.c_uint64(keys.size), _ptr(cent), C.c_uint32(cent.shape[0]), _ptr(out)))
        return out

    # ---- Hilbert
    def hilbert_xy(self, w, h):
        xy = np.zeros((max(w * h, 1), 2), np.uint32)
        self._check(self._L.cniic_hilbert_xy(self.h, C.c_uint32(w), C.c_uint32(h), _ptr(xy)))
        return xy[:w * h]

    def hilbert_linearize(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        out = np.empty((h * w, 3), np.uint8)
        self._check(self._L.cniic_hilbert_linearize(self.h, _ptr(img), C.c_uint32(w), C.c_uint32(h), _ptr(out)))
        return out

    def hilbert_delta(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        syms = np.empty(h * w, np.uint32)
        self._check(self._L.cniic_hilbert_delta(self.h, _ptr(img), C.c_uint32(w), C.c_uint32(h), _ptr(syms)))
        return syms

    def hilbert_delta_hist(self, img, want_syms=False, w=None, h=None):
        if isinstance(img, np.ndarray):
            img = np.ascontiguousarray(img, np.uint8)
            h, w = img.shape[:2]
        nu = C.c_uint64(0)
        self._check(self._L.cniic_hilbert_delta_hist(self.h, _ptr(img), C.c_uint32(w), C.c_uint32(h), None, None, C.c_uint64(0),
                                                     C.byref(nu), None))
        keys = np.empty(max(nu.value, 1), np.uint32)
        counts = np.empty(max(nu.value, 1), np.uint64)
        syms = np.empty(h * w, np.uint32) if want_syms else None
        self._check(self._L.cniic_hilbert_delta_hist(self.h, _ptr(img), C.c_uint32(w), C.c_uint32(h), _ptr(keys), _ptr(counts),
                                                     C.c_uint64(keys.size), C.byref(nu), _ptr(syms)))
        return keys[:nu.value], counts[:nu.value], syms

    # ---- Huffman
    def huf_encode_all(self, kind, syms):
        syms = np.ascontiguousarray(syms, np.uint32)
        cap = 64 + syms.size * 20
        out = np.empty(cap, np.uint8)
        ln = C.c_uint64(0)
        self._check(self._L.cniic_huf_encode_all(self.h, kind, _ptr(syms), C.c_uint64(syms.size), _ptr(out), C.c_uint64(cap), C.byref(ln)))
        return out[:ln.value].tobytes()

    def huf_size(self, kind, counts):
        counts = np.ascontiguousarray(counts, np.uint64)
        nb = C.c_uint64(0)
        rc = self._L.cniic_huf_size(kind, _ptr(counts), C.c_uint64(counts.size), C.byref(nb))
        if rc != OK:
            raise CniicError(rc)
        return nb.value

    # ---- codecs
    def encode(self, expr, img, w=None, h=None, out=None, seed=0, max_iters=0, flags=0, allow=()):
        """Codec::encode.  img: HxWx3 uint8 numpy array, or a device tensor / address with w,h given."""
        if isinstance(img, np.ndarray):
            img = np.ascontiguousarray(img, np.uint8)
            h, w = img.shape[:2]
        own = out is None
        if own:
            cap = 64 + w * h * 16 + (1 << 16)
            out = np.empty(cap, np.uint8)
        else:
            cap = out.numel() if hasattr(out, "numel") else out.size
        ln = C.c_uint64(0)
        st = KmStats()
        o = self._opts(seed, max_iters, flags)
        rc = self._check(self._L.cniic_codec_encode_opts(self.h, expr.encode(), C.byref(o), _ptr(img), C.c_uint32(w), C.c_uint32(h),
                                                         _ptr(out), C.c_uint64(cap), C.byref(ln), C.byref(st)), allow)
        if own:
            return rc, (out[:ln.value].tobytes() if rc == OK else b""), st.as_dict()
        return rc, ln.value, st.as_dict()

    def encode_batch(self, expr, frames, w, h, F, out, stride, seed=0, max_iters=0, flags=0, allow=()):
        """cniic_codec_encode_batch: F images (one contiguous [F][h][w][3] buffer), each encoded on its own (its own palette), image f's
        stream at out[f * stride:].  -> (rc, list of F lengths, list of F per-image status codes, list of F stats dicts)"""
        lens = (C.c_uint64 * F)()
        rcs = (C.c_int32 * F)()
        sts = (KmStats * F)()
        o = self._opts(seed, max_iters, flags)
        rc = self._check(self._L.cniic_codec_encode_batch(self.h, expr.encode(), C.byref(o), _ptr(frames), C.c_uint32(w), C.c_uint32(h), C.c_uint32(F),
                                                          _ptr(out), C.c_uint64(stride), lens, rcs, sts), allow)
        return rc, [int(x) for x in lens], [int(x) for x in rcs], [s.as_dict() for s in sts]

    def decode(self, expr, data, allow=()):
        raw = np.frombuffer(bytes(data), np.uint8)
        if raw.size < 8:
            return DECODE, None
        w = int.from_bytes(raw[0:4].tobytes(), "little")
        h = int.from_bytes(raw[4:8].tobytes(), "little")
        if w * h > (1 << 28):
            return CAPACITY, None
        out = np.zeros((max(w * h, 1), 3), np.uint8)
        cw, ch = C.c_uint32(0), C.c_uint32(0)
        rc = self._check(self._L.cniic_codec_decode(self.h, expr.encode(), _ptr(raw), C.c_uint64(raw.size), _ptr(out),
                                                    C.c_uint64(out.size), C.byref(cw), C.byref(ch)), allow)
        if rc != OK:
            return rc, None
        return rc, out[:w * h].reshape(h, w, 3)

    def decode_into(self, expr, data, nbytes, out, allow=()):
        """Codec::decode with caller-owned buffers: data = the stream (device tensor / address / numpy array), nbytes of it;
        out = a uint8 buffer (device tensor or numpy array) that receives the w x h x 3 image.  -> (rc, w, h)"""
        cap = out.numel() if hasattr(out, "numel") else out.size
        cw, ch = C.c_uint32(0), C.c_uint32(0)
        rc = self._check(self._L.cniic_codec_decode(self.h, expr.encode(), _ptr(data), C.c_uint64(nbytes), _ptr(out), C.c_uint64(cap),
                                                    C.byref(cw), C.byref(ch)), allow)
        return rc, cw.value, ch.value

    def mse(self, a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        v = C.c_double(0)
        self._check(self._L.cniic_mse(self.h, _ptr(a), _ptr(b), C.c_uint64(a.size // 3), C.byref(v)))
        return v.value

    def synth_image(self, kind, seed, w, h, out=None):
        if out is None:
            out = np.empty((h, w, 3), np.uint8)
        self._check(self._L.cniic_synth_image(self.h, kind, C.c_uint64(seed), C.c_uint32(w), C.c_uint32(h), _ptr(out)))
        return out


def codec_parse(expr):
    kind, arg = C.c_int32(0), C.c_uint32(0)
    rc = lib().cniic_codec_parse(expr.encode(), C.byref(kind), C.byref(arg))
    if rc != OK:
        return None
    return kind.value, arg.value


def codec_name(expr):
    buf = C.create_string_buffer(64)
    rc = lib().cniic_codec_name(expr.encode(), buf, C.c_uint64(64))
    if rc != OK:
        raise CniicError(rc, "Malformed codec argument: %s" % expr)
    return buf.value.decode()


def codec_is_lossless(expr):
    rc = lib().cniic_codec_is_lossless(expr.encode())
    if rc < 0:
        raise CniicError(rc, "Malformed codec argument: %s" % expr)
    return bool(rc)
