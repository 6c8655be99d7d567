"""The ABI's stream-order contract (include/cniic_hip.h, "STREAM ORDER OF DEVICE BUFFERS"): a device buffer handed to a call must be
complete with respect to the context's stream.  Round 3's one wrong result in 40 000 fuzz cases (`delta` decode: "colour out of
range", gpurun_out/flake_fuzz_3.log) was a buffer filled on torch's stream and decoded on a context with a stream of its own.
This replays that input (tests/golden/stream_order_case.npz: the 512 x 512 noise image the fuzzer had drawn, knob
CNIIC_HUF_GPU_CODES_MIN=0) the way the contract says -- fill, synchronise, decode on a PRIVATE-stream context -- at every byte
alignment, and once more with the context on torch's own stream, where stream order alone suffices."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_saved_failing_case_decodes_after_an_explicit_sync(monkeypatch):
    import torch

    import cniic_amd
    img = np.load(os.path.join(HERE, "golden", "stream_order_case.npz"))["img"]
    assert img.shape == (512, 512, 3)
    monkeypatch.setenv("CNIIC_HUF_GPU_CODES_MIN", "0")
    dev = torch.device("cuda", 0)
    rco, want, _ = O.encode("delta", img)
    assert rco == 0
    with cniic_amd.Context(0) as ctx:                     # a stream of the context's own: nothing orders it against torch's
        rc, data, _ = ctx.encode("delta", img)
        assert rc == 0 and data == want
        for shift in (0, 1, 2, 3):
            buf = torch.zeros(len(data) + 8, dtype=torch.uint8, device=dev)
            buf[shift:shift + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
            out = torch.zeros(img.size, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()                      # the contract: torch's fills have landed before the context reads them
            rc, dw, dh = ctx.decode_into("delta", buf[shift:], len(data), out)
            assert rc == 0 and (dw, dh) == (512, 512)
            assert np.array_equal(out.cpu().numpy().reshape(img.shape), img), shift
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):                            # the context ON the producing stream: stream order is the synchronisation
        with cniic_amd.Context(0, stream=s.cuda_stream) as ctx:
            for shift in (0, 3):
                buf = torch.zeros(len(want) + 8, dtype=torch.uint8, device=dev)
                buf[shift:shift + len(want)] = torch.frombuffer(bytearray(want), dtype=torch.uint8).to(dev, non_blocking=True)
                out = torch.zeros(img.size, dtype=torch.uint8, device=dev)
                rc, dw, dh = ctx.decode_into("delta", buf[shift:], len(want), out)
                assert rc == 0 and np.array_equal(out.cpu().numpy().reshape(img.shape), img), shift
