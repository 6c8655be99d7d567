"""The one-shot exchange (cniic_comm_create_mailbox, k_mailbox.hip): an all-reduce as ONE kernel per rank that writes the
buffer into every peer's mailbox and adds the slots in rank order.  What one GPU can show: several ranks inside one process
(two or three contexts, each with its own stream and mailbox), and -- tests/test_dist.py -- several PROCESSES that share the
GPU and map each other's mailboxes through HIP IPC, driving the library's own K-means loop.  Known answers: unsigned sums
of 1-, 4- and 8-byte elements, odd lengths, buffers that go in pieces."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

pytestmark = pytest.mark.gpu


class Ranks:
    """`world` ranks in this process: a context with a stream of its own and a mailbox each.  (The contexts' PRIVATE streams, made
    here and destroyed in close(): streams from torch's pool are never released, a process has four hardware queues, and two
    ranks whose streams share a queue run their kernels in turn -- the first waits for the second until its timeout.)"""

    def __init__(self, world, max_bytes=0, timeout_ms=5000):
        import torch
        import cniic_amd
        from cniic_amd import _lib
        self.torch, self.L, self.world = torch, _lib.lib(), world
        self.dev = torch.device("cuda", 0)
        self.ctxs = [cniic_amd.Context(0) for _ in range(world)]
        self.comms, handles = [], []
        for r in range(world):
            hb, h = (C.c_uint8 * 64)(), C.c_void_p()
            self.ctxs[r]._check(self.L.cniic_comm_create_mailbox(self.ctxs[r].h, C.c_uint32(r), C.c_uint32(world), C.c_uint64(max_bytes), hb, C.byref(h)))
            self.comms.append(h)
            handles += list(hb)
        raw = (C.c_uint8 * len(handles))(*handles)
        for r in range(world):
            self.ctxs[r]._check(self.L.cniic_comm_connect_mailbox(self.comms[r], raw))
            self.ctxs[r]._check(self.L.cniic_comm_set_timeout(self.comms[r], C.c_uint64(timeout_ms)))

    def all_reduce(self, bufs):
        """enqueue every rank's kernel (each waits for the others inside), then wait for all"""
        self.torch.cuda.synchronize()
        rcs = [self.L.cniic_comm_all_reduce(self.comms[r], C.c_void_p(bufs[r].data_ptr()), C.c_uint64(bufs[r].numel()), C.c_int32(bufs[r].element_size()))
               for r in range(self.world)]
        self.torch.cuda.synchronize()
        return rcs

    def close(self):
        for h in self.comms:
            self.L.cniic_comm_destroy(h)
        for c in self.ctxs:
            c.close()


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("dtype,count", [("int64", 5 * 256 + 2), ("int64", 1), ("int32", 1000), ("int32", 4097), ("uint8", 4096 * 3 + 5), ("uint8", 3),
                                         ("int64", 40000), ("uint8", 100001)])
def test_all_reduce_known_answers(world, dtype, count):
    import torch
    R = Ranks(world, max_bytes=64 << 10)   # (the 40 000-word and the 100 001-byte buffers go in pieces)
    try:
        rng = np.random.default_rng(count * 7 + world)
        hi = {"int64": 1 << 62, "int32": 1 << 31, "uint8": 256 // max(world, 1)}[dtype]   # (bytes: lanes must not carry, as the nibble sums never do)
        host = [rng.integers(0, hi, count, dtype=np.int64).astype(dtype) for _ in range(world)]
        exp = host[0].copy()
        for h in host[1:]:
            exp = (exp + h).astype(dtype)   # wraps like the unsigned sum
        for rep in range(3):                # the same mailboxes again: both parities, and once more
            bufs = [torch.from_numpy(h.copy()).to(R.dev) for h in host]
            assert R.all_reduce(bufs) == [0] * world
            for r in range(world):
                assert np.array_equal(bufs[r].cpu().numpy(), exp), "rank %d, repeat %d" % (r, rep)
    finally:
        R.close()


def test_bytes_next_to_the_buffer_stay():
    """a length that is no multiple of four: the last word is read and written byte by byte"""
    import torch
    R = Ranks(2)
    try:
        bufs = [torch.full((64,), 9, dtype=torch.uint8, device=R.dev) for _ in range(2)]
        views = [b[:13] for b in bufs]
        for v in views:
            v.fill_(3)
        assert R.all_reduce(views) == [0, 0]
        for b in bufs:
            h = b.cpu().numpy()
            assert (h[:13] == 6).all() and (h[13:] == 9).all()
    finally:
        R.close()


def test_a_peer_that_never_comes_ends_as_an_error_not_a_hang():
    """rank 1 never enqueues its all-reduce: rank 0's kernel gives up after the communicator's timeout and the error
    surfaces on the next call"""
    import torch
    from cniic_amd import _lib
    R = Ranks(2, timeout_ms=300)
    try:
        b = torch.ones(100, dtype=torch.int64, device=R.dev)
        torch.cuda.synchronize()
        assert R.L.cniic_comm_all_reduce(R.comms[0], C.c_void_p(b.data_ptr()), C.c_uint64(100), C.c_int32(8)) == 0
        torch.cuda.synchronize()   # returns: the wait inside the kernel is bounded
        assert R.L.cniic_comm_all_reduce(R.comms[0], C.c_void_p(b.data_ptr()), C.c_uint64(100), C.c_int32(8)) == _lib.RCCL
        assert b"did not arrive" in R.L.cniic_last_error(R.ctxs[0].h)
    finally:
        R.close()


def test_misuse():
    import torch
    from cniic_amd import _lib
    import cniic_amd
    L = _lib.lib()
    ctx = cniic_amd.Context(0)
    hb, h = (C.c_uint8 * 64)(), C.c_void_p()
    assert L.cniic_comm_create_mailbox(ctx.h, C.c_uint32(3), C.c_uint32(2), C.c_uint64(0), hb, C.byref(h)) == _lib.BAD_ARG
    assert L.cniic_comm_create_mailbox(ctx.h, C.c_uint32(0), C.c_uint32(17), C.c_uint64(0), hb, C.byref(h)) == _lib.BAD_ARG
    assert L.cniic_comm_create_mailbox(ctx.h, C.c_uint32(0), C.c_uint32(1), C.c_uint64(0), hb, C.byref(h)) == 0
    b = torch.ones(8, dtype=torch.int64, device="cuda:0")
    assert L.cniic_comm_all_reduce(h, C.c_void_p(b.data_ptr()), C.c_uint64(8), C.c_int32(8)) == _lib.BAD_ARG   # not connected yet
    wrong = (C.c_uint8 * 64)(*([1] * 64))
    assert L.cniic_comm_connect_mailbox(h, wrong) == _lib.BAD_ARG   # the own entry must be the own handle
    assert L.cniic_comm_connect_mailbox(h, hb) == 0
    assert L.cniic_comm_all_reduce(h, C.c_void_p(b.data_ptr() + 1), C.c_uint64(7), C.c_int32(1)) == _lib.BAD_ARG  # alignment
    L.cniic_comm_destroy(h)
    ctx.close()
