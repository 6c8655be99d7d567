"""Shared-palette `cluster-colors` over several GPUs (north_star config 4), one process per GPU.

Every rank holds its own image; the ranks cluster the UNION of their pixels into one K-colour
palette and each encodes its own image with it:

    local dense colour counts --all-reduce--> global counts -> distinct colours (same on all ranks)
    K-means: every rank assigns its share of the colour-space cells, the K partial sums are
             all-reduced (RCCL, sum of int64 words) each iteration, every rank updates identically
    labels of the shards are merged (all-reduce of a byte array with one owner per element)
    each rank Huffman-codes its own colour-reduced image (clusterc.rs:31-52)

Sums are integers, so the palette is bit-identical for 1, 2, 4 or 8 ranks.  The collectives are
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests); the
compute is behind a small backend interface: HipBackend drives the C ABI (the product), the
tests plug a CPU checker in to exercise this driver without a GPU.
"""
import ctypes as C

from . import _lib


class HipBackend:
    """Everything on the GPU through libcniic_hip.so; tensors are torch device tensors."""

    def __init__(self, ctx, device):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.dev = device
        self.L = _lib.lib()

    def hist_dense(self, img, npx):
        t = self.torch.empty(1 << 24, dtype=self.torch.int32, device=self.dev)
        self.ctx._check(self.L.cniic_hist_rgb24_dense(self.ctx.h, C.c_void_p(img.data_ptr()), C.c_uint64(npx), C.c_void_p(t.data_ptr())))
        return t

    def new_partials(self, K):
        return self.torch.zeros(int(self.L.cniic_km_partial_words(K, 3)), dtype=self.torch.int64, device=self.dev)

    def cc_create(self, table, K, rank, world, partials, max_iters=0, seed=0):
        h = C.c_void_p()
        o = _lib.KmOpts(seed, max_iters, 0, 0)
        self.ctx._check(self.L.cniic_cc_create(self.ctx.h, C.c_void_p(table.data_ptr()), C.c_uint32(K), C.byref(o), C.c_uint32(rank),
                                               C.c_uint32(world), C.c_void_p(partials.data_ptr()), C.byref(h)))
        return h

    def assign(self, h):
        self.ctx._check(self.L.cniic_cc_assign(h))

    def update(self, h):
        """asynchronous centroid update (no host round trip)"""
        self.ctx._check(self.L.cniic_cc_update(h, None))

    def poll(self, h):
        """(iterations completed, converged?) -- synchronises the stream"""
        it, done = C.c_uint64(0), C.c_uint32(0)
        self.ctx._check(self.L.cniic_cc_poll(h, C.byref(it), C.byref(done)))
        return it.value, bool(done.value)

    def poll_lagged(self, h):
        """(valid?, iterations, converged?) as of the PREVIOUS call: no GPU stall"""
        it, done, valid = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
        self.ctx._check(self.L.cniic_cc_poll_lagged(h, C.byref(it), C.byref(done), C.byref(valid)))
        return bool(valid.value), it.value, bool(done.value)

    def export_labels(self, h):
        U = int(self.L.cniic_cc_unique(h))
        dt = self.torch.uint8 if self.L.cniic_cc_label_bytes(h) == 1 else self.torch.int16
        t = self.torch.empty(U, dtype=dt, device=self.dev)
        self.ctx._check(self.L.cniic_cc_export_labels(h, C.c_void_p(t.data_ptr())))
        return t

    def import_labels(self, h, t):
        self.ctx._check(self.L.cniic_cc_import_labels(h, C.c_void_p(t.data_ptr())))

    def finish(self, h, img, w, hh, local_table, out):
        ln = C.c_uint64(0)
        st = _lib.KmStats()
        cap = out.numel()
        lt = C.c_void_p(local_table.data_ptr()) if local_table is not None else None
        self.ctx._check(self.L.cniic_cc_finish(h, C.c_void_p(img.data_ptr()), C.c_uint32(w), C.c_uint32(hh), lt,
                                               C.c_void_p(out.data_ptr()), C.c_uint64(cap), C.byref(ln), C.byref(st)))
        return ln.value, st.as_dict()

    def destroy(self, h):
        self.L.cniic_cc_destroy(h)


class ShardedClusterColors:
    """encode(img, w, h, out) -> (stream bytes written to out, K-means stats), collectively on all ranks."""

    def __init__(self, ctx, K, dist, device, max_iters=0, backend=None, poll_every=4):
        self.K = K
        self.poll_every = poll_every
        self.dist = dist
        self.max_iters = max_iters
        self.be = backend if backend is not None else HipBackend(ctx, device)
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1

    def _all_reduce(self, t):
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(t)  # SUM

    def encode(self, img, w, h, out):
        be = self.be
        local = be.hist_dense(img, w * h)           # utils::count_freqs of this rank's pixels
        glob = local.clone()
        self._all_reduce(glob)                      # colour counts of the union
        partials = be.new_partials(self.K)
        handle = be.cc_create(glob, self.K, self.rank, self.world, partials, self.max_iters)
        try:
            while True:                             # kmeans.rs:26-32 `while changed_assignment`
                for _ in range(self.poll_every):    # no host round trip inside a batch; iterations issued after
                    be.assign(handle)               # convergence are no-ops on every rank (device-side flag)
                    self._all_reduce(partials)      # K partial centroid sums (+ moved count), identical on all ranks
                    be.update(handle)
                # the answer lags one batch (the state after the previous batch), the same on every rank, so the
                # GPU never waits for the host; a backend without it polls synchronously
                if hasattr(be, "poll_lagged"):
                    valid, it, done = be.poll_lagged(handle)
                    done = valid and done
                else:
                    it, done = be.poll(handle)
                if done:
                    break
            if self.world > 1:
                lab = be.export_labels(handle)
                if getattr(lab, "element_size", lambda: 1)() == 2:   # K > 256: RCCL has no 16-bit integer type
                    wide = lab.to(dtype=self.be.torch.int32) if hasattr(self.be, "torch") else lab.astype("int32")
                    self._all_reduce(wide)
                    lab = wide.to(dtype=lab.dtype) if hasattr(self.be, "torch") else wide.astype(lab.dtype)
                else:
                    self._all_reduce(lab)           # one owner per element, zeros elsewhere
                be.import_labels(handle, lab)
            return be.finish(handle, img, w, h, local if self.world > 1 else None, out)
        finally:
            be.destroy(handle)
