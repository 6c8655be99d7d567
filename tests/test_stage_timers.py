"""The stage timers of a `delta` encode (CNIIC_KM_PROFILE in the call's options; bench.py --config c5 reports them as `stages`): the five
stages are consecutive and together the whole call, so their sum must account for the call's wall time -- and the stream of a timed
call (the pack in one piece, every stage synchronised) must be the stream of an untimed one."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STAGES = ("delta_gather", "delta_hist", "delta_tree", "huff_pack", "delta_finish")


def test_delta_stages_account_for_the_call():
    import torch

    import cniic_amd
    from cniic_amd import _lib, synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    with cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream) as ctx:
        W = H = 2048
        img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 5, W, H, out=img)
        out = torch.empty(W * H * 3 + (1 << 20), dtype=torch.uint8, device=dev)
        timed = torch.empty_like(out)
        rc, n, _ = ctx.encode("delta", img, w=W, h=H, out=out)   # (warm: tables, pools)
        assert rc == 0
        rc, n, _ = ctx.encode("delta", img, w=W, h=H, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc, nt, _ = ctx.encode("delta", img, w=W, h=H, out=timed, flags=_lib.KM_PROFILE)
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3
        assert rc == 0 and nt == n
        assert np.array_equal(timed[:n].cpu().numpy(), out[:n].cpu().numpy())
        ms = {}
        for k in STAGES:
            t, cnt = ctx.kernel_time(k)
            assert cnt >= 1 and t > 0.0, k
            ms[k] = t / cnt
        total = sum(ms.values())
        # consecutive stages on one stream, each closed by a wait: their sum is the call less the host's few microseconds between them
        # (the lower bound is loose on purpose: at this size the call is a quarter of a millisecond and the host's share between stages shows)
        assert 0.4 * wall_ms <= total <= 1.05 * wall_ms, (ms, wall_ms)
