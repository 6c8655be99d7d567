"""cniic_ctx_set_scan: the reference's scan comes from a crate this build cannot see (hilbert.rs:40-43 -> zhang_hilbert 0.1.1), so a
host that has the crate injects its order.  Tested with orders that are NOT the built-in one: with the scan s injected, the HIP
path must produce what the oracle produces for the image whose pixels are re-arranged so that the oracle's own scan meets them in
s's order -- same linear sequence, same dimensions, hence the same `delta` / `hilbert(rle)` bytes -- and decode must undo it."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def snake(w, h):
    xy = np.empty((h, w, 2), np.uint32)
    xs = np.arange(w, dtype=np.uint32)
    for y in range(h):
        xy[y, :, 0] = xs if y % 2 == 0 else xs[::-1]
        xy[y, :, 1] = y
    return xy.reshape(-1, 2)


def shuffled(w, h, seed):
    p = np.random.default_rng(seed).permutation(w * h)
    return np.stack([p % w, p // w], 1).astype(np.uint32)


@pytest.mark.parametrize("w,h,kind", [(37, 29, "snake"), (64, 64, "snake"), (64, 64, "shuffled"), (256, 256, "shuffled"), (5, 1, "shuffled"), (130, 70, "shuffled")])
def test_injected_scan_equals_oracle_on_the_rearranged_image(w, h, kind):
    import cniic_amd
    from cniic_amd import _lib, synth
    img = synth.photo(w, h, synth.SEED0 + 300 + w)
    s = snake(w, h) if kind == "snake" else shuffled(w, h, w * h)
    own = O.hilbert_iter(w, h)                                         # the oracle's scan
    perm = np.empty_like(img)
    perm[own[:, 1], own[:, 0]] = img[s[:, 1], s[:, 0]]                 # the oracle meets perm's pixels in s's order of img's
    with cniic_amd.Context(0) as ctx:
        builtin = {e: ctx.encode(e, img)[1] for e in ("delta", "hilbert(rle)")}
        ctx.set_scan(w, h, s)
        assert np.array_equal(ctx.hilbert_xy(w, h), s)
        assert np.array_equal(ctx.hilbert_linearize(img).reshape(-1, 3), img[s[:, 1], s[:, 0]])
        for expr in ("delta", "hilbert(rle)"):
            rco, exp, _ = O.encode(expr, perm)
            for route in (None, 32):
                ctx.set_opt(_lib.OPT_DELTA_ROUTE, route)
                rc, data, _ = ctx.encode(expr, img)
                assert rc == rco == 0 and data == exp, (expr, route)
            ctx.set_opt(_lib.OPT_DELTA_ROUTE, None)
            for gmin in (0, None):
                ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, gmin)
                rc, back = ctx.decode(expr, exp)
                assert rc == 0 and np.array_equal(back, img), expr
            ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, None)
        # codecs that do not scan are untouched; other dimensions keep the built-in scan
        assert ctx.encode("hufman", img)[1] == O.encode("hufman", img)[1]
        other = synth.photo(w + 1, h, 5)
        assert ctx.encode("delta", other)[1] == O.encode("delta", other)[1]
        # not a scan: a pixel twice / a position outside
        bad = s.copy()
        bad[-1] = bad[0]
        with pytest.raises(cniic_amd.CniicError):
            ctx.set_scan(w, h, bad)
        bad = s.copy()
        bad[0, 0] = w
        with pytest.raises(cniic_amd.CniicError):
            ctx.set_scan(w, h, bad)
        assert ctx.encode("delta", img)[1] == builtin["delta"]           # a refused scan leaves the built-in one
        ctx.set_scan(w, h, s)
        ctx.set_scan(w, h, None)
        for e in ("delta", "hilbert(rle)"):
            assert ctx.encode(e, img)[1] == builtin[e] == O.encode(e, img)[1]


def test_injected_scan_in_a_batch_and_from_device_memory():
    import torch
    import cniic_amd
    from cniic_amd import synth
    w, h, F = 64, 64, 5
    dev = torch.device("cuda", 0)
    s = shuffled(w, h, 7)
    own = O.hilbert_iter(w, h)
    frames = np.stack([synth.photo(w, h, synth.SEED0 + 400 + f) for f in range(F)])
    with cniic_amd.Context(0) as ctx:
        ctx.set_scan(w, h, torch.from_numpy(s.astype(np.int64)).to(torch.int32).to(dev))   # (u32 values as an int32 tensor)
        stride = w * h * 16 + 4096
        out = torch.zeros(stride * F, dtype=torch.uint8, device=dev)
        fr_d = torch.from_numpy(frames).to(dev)
        torch.cuda.synchronize()   # (the context runs on a stream of its own: torch's fills must have landed)
        rc, lens, rcs, _ = ctx.encode_batch("delta", fr_d, w, h, F, out, stride)
        host = out.cpu().numpy()
        for f in range(F):
            perm = np.empty_like(frames[f])
            perm[own[:, 1], own[:, 0]] = frames[f][s[:, 1], s[:, 0]]
            assert host[f * stride:f * stride + lens[f]].tobytes() == O.encode("delta", perm)[1], f
