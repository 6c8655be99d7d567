"""An image handed over at an address that is no multiple of 4 (a slice of a larger device buffer): every encoder must give the bytes it
gives for the same image at an aligned address.  Round 4 moved the K-means kernels' point and pixel loads to buffer loads, whose
descriptors take the caller's pointer as their base; `delta` reads 16-byte pieces only from 16-byte aligned images and takes the
per-position gather otherwise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("expr", ["voronoi(40)", "cluster-colors(16)", "delta", "hufman", "hilbert(rle)"])
@pytest.mark.parametrize("shift", [1, 2, 3, 5])
def test_encode_of_an_unaligned_device_image(expr, shift):
    import torch

    import cniic_amd
    from cniic_amd import _lib
    rng = np.random.default_rng(77)
    h, w = 97, 131   # (odd on purpose: the image's last pixel ends the buffer)
    img = (rng.integers(0, 40, (h, w, 3)) * 6 + rng.integers(0, 3, (h, w, 3))).astype(np.uint8)
    dev = torch.device("cuda", 0)
    with cniic_amd.Context(0) as ctx:
        rc0, want, st0 = ctx.encode(expr, img, allow=(_lib.FEW_ACTIVE,))
        assert rc0 in (0, _lib.FEW_ACTIVE)
        big = torch.zeros(img.size + 64, dtype=torch.uint8, device=dev)
        big[shift:shift + img.size] = torch.from_numpy(img.reshape(-1)).to(dev)
        out = torch.zeros(img.size * 4 + (1 << 16), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()   # (the context has a stream of its own: include/cniic_hip.h, stream order of device buffers)
        rc, n, st = ctx.encode(expr, big[shift:shift + img.size], w=w, h=h, out=out, allow=(_lib.FEW_ACTIVE,))
        assert rc == rc0 and n == len(want)
        assert out[:n].cpu().numpy().tobytes() == want
