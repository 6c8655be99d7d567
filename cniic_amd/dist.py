"""Shared-palette `cluster-colors` over several GPUs (north_star config 4), one process per GPU.

Every rank holds its own image; the ranks cluster the UNION of their pixels into one K-colour
palette and each encodes its own image with it.  A rank keeps only ITS OWN image's distinct colours:

    local dense colour counts -> occupancy nibbles --all-reduce(sum, 8 MiB)--> which colours occur anywhere:
        the reference's point list (ascending distinct colours of all pixels) as a bitmap + popcount prefix,
        from which every rank initialises its own points (init_assignment / init_centroids, kmeans.rs:61-108)
    K-means: every rank assigns its own colours, the K partial sums are all-reduced (RCCL, sum of
             u64 words) each iteration, every rank updates identically
    each rank Huffman-codes its own colour-reduced image (clusterc.rs:31-52) with its own cluster weights

A colour held by several ranks is several points at one position: each copy takes the same decisions and
the integer sums equal those of the single merged point, so the palette is bit-identical for 1, 2, 4 or 8
ranks and to clustering the union in one process.  Per-rank work does not grow with the rank count and no
labels are exchanged.  (The first protocol all-reduced the 64 MiB count table and made every rank hold all
the union's colours; its entry points -- cniic_cc_create, export / import labels -- remain in the ABI.)

Collectives, two ways:
  * native (default on GPUs): the library's own RCCL communicator on the context's stream
    (cniic_comm_*); the whole K-means loop is one C call (cniic_cc_run) that enqueues
    assign -> ncclAllReduce -> update per iteration with no host round trip.  torch.distributed
    only carries the 128-byte RCCL id at set-up.
  * mailbox (CNIIC_COLLECTIVES=mailbox / collectives="mailbox"): the same C loop, the all-reduce a one-shot exchange
    over IPC-mapped mailboxes (cniic_comm_create_mailbox, k_mailbox.hip): every rank writes its sums into every peer's
    mailbox and adds the slots in rank order.  torch.distributed carries the 64-byte IPC handles at set-up.
  * torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests), driven from
    Python per iteration.  Used when RCCL cannot be bound, with CNIIC_COLLECTIVES=torch, and by the
    CPU tests, which plug a CPU checker in as compute backend to exercise this driver without a GPU.
"""
import os
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
        # (uninitialised: the session zeroes it on the context's stream when it is created -- torch's stream may be another)
        return self.torch.empty(int(self.L.cniic_km_partial_words(K, 3)), dtype=self.torch.int64, device=self.dev)

    def occupancy(self, table):
        """one nibble per colour, 1 where this image has it (u32[2^21]); summed over the ranks by an all-reduce"""
        occ = self.torch.empty(1 << 21, dtype=self.torch.int32, device=self.dev)
        self.ctx._check(self.L.cniic_occupancy_pack(self.ctx.h, C.c_void_p(table.data_ptr()), C.c_void_p(occ.data_ptr())))
        return occ

    def cc_create_local(self, table, occ, K, partials, max_iters=0, seed=0, flags=0):
        """K-means state over this rank's colours (table: its counts, overwritten), placed in the list of all occupied colours"""
        h = C.c_void_p()
        o = _lib.KmOpts(seed, max_iters, flags, 0)
        self.ctx._check(self.L.cniic_cc_create_local(self.ctx.h, C.c_void_p(table.data_ptr()), C.c_void_p(occ.data_ptr()), C.c_uint32(K),
                                                     C.byref(o), C.c_void_p(partials.data_ptr()), C.byref(h)))
        return h

    # ---- large images: the pixel partition instead of the dense table (same session handle afterwards)
    def image_begin(self, img, npx):
        """None when the image does not qualify (small, or not 16-byte aligned): the dense-table calls above apply"""
        if npx < int(os.environ.get("CNIIC_SP_MIN_PIXELS", 1 << 20)) or img.data_ptr() % 16:  # (the library's own threshold and knob)
            return None
        h = C.c_void_p()
        self.ctx._check(self.L.cniic_cc_image_begin(self.ctx.h, C.c_void_p(img.data_ptr()), C.c_uint64(npx), C.byref(h)))
        return h

    def image_occupancy(self, h):
        occ = self.torch.empty(1 << 21, dtype=self.torch.int32, device=self.dev)
        self.ctx._check(self.L.cniic_cc_image_occupancy(h, C.c_void_p(occ.data_ptr())))
        return occ

    def image_create(self, h, occ, K, partials, max_iters=0, seed=0, flags=0):
        o = _lib.KmOpts(seed, max_iters, flags, 0)
        self.ctx._check(self.L.cniic_cc_image_create(h, C.c_void_p(occ.data_ptr()), C.c_uint32(K), C.byref(o), C.c_void_p(partials.data_ptr())))

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

    def finish_frames(self, h, frames, w, hh, F, out, stride):
        """F streams, frame f's at out[f * stride:], lengths as a list"""
        lens = (C.c_uint64 * F)()
        st = _lib.KmStats()
        self.ctx._check(self.L.cniic_cc_finish_frames(h, C.c_void_p(frames.data_ptr()), C.c_uint32(w), C.c_uint32(hh), C.c_uint32(F),
                                                      C.c_void_p(out.data_ptr()), C.c_uint64(stride), lens, C.byref(st)))
        return list(lens), st.as_dict()

    def destroy(self, h):
        self.L.cniic_cc_destroy(h)

    # ---- native RCCL on the context's stream
    def comm_create(self, dist, rank, world):
        """the library's communicator, or None when RCCL cannot be bound (the id travels over torch.distributed)"""
        torch = self.torch
        idbuf = (C.c_uint8 * 128)()
        ok = 1
        if rank == 0:
            ok = 1 if self.L.cniic_comm_unique_id(idbuf) == _lib.OK else 0
        # one small CPU/GPU tensor broadcast; its first element says whether rank 0 could make an id
        backend = dist.get_backend() if dist is not None else "gloo"
        t = torch.zeros(129, dtype=torch.uint8, device=self.dev if backend == "nccl" else "cpu")
        if rank == 0:
            t[0] = ok
            t[1:] = torch.tensor(list(idbuf), dtype=torch.uint8)
        if dist is not None and world > 1:
            dist.broadcast(t, src=0)
        raw = bytes(t.cpu().tolist())
        if raw[0] != 1:
            return None
        idb = (C.c_uint8 * 128)(*raw[1:])
        h = C.c_void_p()
        rc = self.L.cniic_comm_create(self.ctx.h, idb, C.c_uint32(rank), C.c_uint32(world), C.byref(h))
        def agreed(mine):  # all ranks or none
            if dist is None or world == 1:
                return mine
            f = torch.tensor([mine], dtype=torch.int32, device=self.dev if backend == "nccl" else "cpu")
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            return int(f.item())

        mine = 1 if rc == _lib.OK else 0
        if not agreed(mine):
            if mine:
                self.L.cniic_comm_destroy(h)
            return None
        # known-answer all-reduce before the K-means loop depends on it: rank r gives r + 1 in every word
        probe = torch.full((5 * 256 + 2,), rank + 1, dtype=torch.int64, device=self.dev)
        torch.cuda.synchronize(self.dev)  # (filled on torch's stream, reduced on the context's)
        rc = self.L.cniic_comm_all_reduce(h, C.c_void_p(probe.data_ptr()), C.c_uint64(probe.numel()), C.c_int32(8))
        torch.cuda.synchronize(self.dev)
        good = 1 if rc == _lib.OK and bool((probe == world * (world + 1) // 2).all()) else 0
        if not agreed(good):
            self.L.cniic_comm_destroy(h)
            return None
        return h

    def comm_create_mailbox(self, dist, rank, world, max_bytes=0):
        """the one-shot exchange (cniic_comm_create_mailbox): every rank's 64-byte IPC handle is all-gathered over
        torch.distributed, then a known-answer all-reduce; None (on every rank) when any rank could not set it up"""
        torch = self.torch
        backend = dist.get_backend() if dist is not None else "gloo"
        where = self.dev if backend == "nccl" else "cpu"

        def agreed(mine):  # all ranks or none
            if dist is None or world == 1:
                return mine
            f = torch.tensor([mine], dtype=torch.int32, device=where)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            return int(f.item())

        hb = (C.c_uint8 * 64)()
        h = C.c_void_p()
        rc = self.L.cniic_comm_create_mailbox(self.ctx.h, C.c_uint32(rank), C.c_uint32(world), C.c_uint64(max_bytes), hb, C.byref(h))
        mine = 1 if rc == _lib.OK else 0
        if not agreed(mine):
            if mine:
                self.L.cniic_comm_destroy(h)
            return None
        t = torch.tensor(list(hb), dtype=torch.uint8, device=where)
        if dist is not None and world > 1:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            allh = torch.cat(parts).cpu()
        else:
            allh = t.cpu()
        raw = (C.c_uint8 * (64 * world))(*allh.tolist())
        rc = self.L.cniic_comm_connect_mailbox(h, raw)
        if not agreed(1 if rc == _lib.OK else 0):
            self.L.cniic_comm_destroy(h)
            return None
        probe = torch.full((5 * 256 + 2,), rank + 1, dtype=torch.int64, device=self.dev)
        torch.cuda.synchronize(self.dev)
        rc = self.L.cniic_comm_all_reduce(h, C.c_void_p(probe.data_ptr()), C.c_uint64(probe.numel()), C.c_int32(8))
        torch.cuda.synchronize(self.dev)
        good = 1 if rc == _lib.OK and bool((probe == world * (world + 1) // 2).all()) else 0
        if not agreed(good):
            self.L.cniic_comm_destroy(h)
            return None
        return h

    def comm_create_host(self, dist, rank, world):
        """the same communicator over torch.distributed's CPU path (gloo): the library's C loop runs unchanged, every
        all-reduce goes through host memory.  For hosts without RCCL, and for exercising the C loop with several ranks
        on one GPU (tests)."""
        torch = self.torch
        import numpy as np

        def host_sum(user, buf, count, elem_bytes):
            if elem_bytes < 0:       # this rank failed inside the library's loop: its peers must not wait for it
                self.host_aborted = True
                return 0
            try:
                dt = {1: np.uint8, 4: np.int32, 8: np.int64}[elem_bytes]   # (two's complement: the sum is the unsigned one)
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(count * elem_bytes,)).view(dt)
                t = torch.from_numpy(a)
                if elem_bytes == 1:
                    w = t.to(torch.int32)
                    dist.all_reduce(w)
                    t.copy_(w.to(torch.uint8))
                else:
                    dist.all_reduce(t)
                return 0
            except Exception:  # never unwind through the C frames
                return 1

        self._host_sum_cb = _lib.HOST_SUM_FN(host_sum)   # kept alive as long as the backend
        h = C.c_void_p()
        self.ctx._check(self.L.cniic_comm_create_host(self.ctx.h, C.c_uint32(rank), C.c_uint32(world), self._host_sum_cb, None, C.byref(h)))
        return h

    def comm_destroy(self, comm):
        if comm is not None:
            self.L.cniic_comm_destroy(comm)

    def comm_all_reduce(self, comm, t):
        self.ctx._check(self.L.cniic_comm_all_reduce(comm, C.c_void_p(t.data_ptr()), C.c_uint64(t.numel()), C.c_int32(t.element_size())))

    def run(self, h, comm):
        st = _lib.KmStats()
        self.ctx._check(self.L.cniic_cc_run(h, comm, C.byref(st)))
        return st.as_dict()


class ShardedClusterColors:
    """encode(img, w, h, out) -> (stream bytes written to out, K-means stats), collectively on all ranks.

    The context must run on torch's current stream (Context(dev, stream=torch.cuda.current_stream().cuda_stream)): the
    torch.distributed fallback reduces the library's buffers on torch's stream, between kernels of the context's."""

    def __init__(self, ctx, K, dist, device, max_iters=0, backend=None, poll_every=4, collectives=None):
        self.K = K
        self.poll_every = poll_every
        self.dist = dist
        self.max_iters = max_iters
        self.flags = 0   # cniic_kmeans_opts.flags of the sessions (bench.py: KM_PROFILE for the per-launch timers)
        self.be = backend if backend is not None else HipBackend(ctx, device)
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        assert self.world <= 15, "the occupancy nibbles are summed over the ranks: at most 15"
        # "native": the library's RCCL communicator; "torch": torch.distributed per iteration
        want = collectives or os.environ.get("CNIIC_COLLECTIVES", "native")
        self.comm = None
        if want == "native" and hasattr(self.be, "comm_create") and (self.world > 1 or collectives == "native"):
            self.comm = self.be.comm_create(dist, self.rank, self.world)
        elif want == "mailbox" and hasattr(self.be, "comm_create_mailbox"):
            self.comm = self.be.comm_create_mailbox(dist, self.rank, self.world)   # one-shot exchange over IPC-mapped mailboxes
        elif want == "host" and hasattr(self.be, "comm_create_host") and dist is not None:
            self.comm = self.be.comm_create_host(dist, self.rank, self.world)   # "host": the C loop over the caller's transport
        self.collectives = want if self.comm is not None else "torch"

    def close(self):
        if self.comm is not None:
            self.be.comm_destroy(self.comm)
            self.comm = None

    def _all_reduce(self, t):
        if self.comm is not None:
            self.be.comm_all_reduce(self.comm, t)  # in-stream, unsigned SUM
        elif self.dist is not None and self.world > 1:
            self.dist.all_reduce(t)  # SUM

    def encode_frames(self, frames, w, h, F, out, stride):
        """north_star config 4: this rank's F frames (one contiguous [F][h][w][3] buffer) and every other rank's are clustered
        into ONE palette; frame f's stream lands at out[f * stride:].  -> (list of F lengths, K-means stats)"""
        handle, partials = self._cluster(frames, w * h * F)
        try:
            return self.be.finish_frames(handle, frames, w, h, F, out, stride)
        finally:
            self.be.destroy(handle)

    def encode(self, img, w, h, out):
        handle, partials = self._cluster(img, w * h)
        try:
            return self.be.finish(handle, img, w, h, None, out)   # its own colours, labels and cluster weights: nothing to exchange
        finally:
            self.be.destroy(handle)

    def _cluster(self, img, npx):
        """the shared K-means over this rank's npx pixels and everybody else's: -> (session handle, partials buffer)"""
        be = self.be
        partials = be.new_partials(self.K)
        handle = be.image_begin(img, npx) if hasattr(be, "image_begin") else None
        if handle is not None:                      # large image: utils::count_freqs through the pixel partition (k_points.hip)
            try:
                occ = be.image_occupancy(handle)
                self._all_reduce(occ)               # which colours occur on any rank (nibble sums, <= 15 ranks)
                be.image_create(handle, occ, self.K, partials, self.max_iters, 0, self.flags) if self.flags else be.image_create(handle, occ, self.K, partials, self.max_iters)
            except Exception:
                be.destroy(handle)
                raise
        else:
            local = be.hist_dense(img, npx)         # utils::count_freqs of this rank's pixels, dense table
            occ = be.occupancy(local)
            self._all_reduce(occ)
            handle = be.cc_create_local(local, occ, self.K, partials, self.max_iters, 0, self.flags) if self.flags else be.cc_create_local(local, occ, self.K, partials, self.max_iters)
        try:
            in_library = self.comm is not None or (self.world == 1 and hasattr(be, "run"))
            if in_library:                          # one C call: assign -> ncclAllReduce -> update per iteration, in-stream
                be.run(handle, self.comm)           # (one rank and no communicator: the plain loop, nothing to reduce)
            while not in_library:                   # kmeans.rs:26-32 `while changed_assignment`
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
            return handle, partials
        except Exception:
            be.destroy(handle)
            raise
