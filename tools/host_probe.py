#!/usr/bin/env python3
"""The trait as the reference calls it -- host image in, host bytes out, and back (bench.rs:33-35, 45-46) -- for every codec at two sizes,
beside the HBM-resident call: what the PCIe legs and the host-side staging add.  Tools only."""
import os, sys, time, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
for w, h in ((1920, 1080), (4096, 4096)):
    img = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, w, h, out=img)
    himg = img.cpu().numpy()
    out = torch.empty(w * h * 13 + (1 << 22), dtype=torch.uint8, device=dev)
    hout = np.empty(w * h * 13 + (1 << 22), np.uint8)
    back = torch.empty(w * h * 3, dtype=torch.uint8, device=dev)
    hback = np.empty(w * h * 3, np.uint8)
    for expr in ("cluster-colors(256)", "hufman", "delta", "hilbert(rle)", "voronoi(64)"):
        ems, (rc, n, st) = t(lambda: ctx.encode(expr, img, w=w, h=h, out=out))
        hms, (rc2, n2, st2) = t(lambda: ctx.encode(expr, himg, out=hout))
        assert n2 == n
        dms, _ = t(lambda: ctx.decode_into(expr, out, n, back))
        hdms, _ = t(lambda: ctx.decode_into(expr, hout, n, hback))
        mb = (w * h * 3 + n) / 1e6
        print(json.dumps(dict(w=w, h=h, codec=expr, stream_MB=round(n / 1e6, 2), encode_hbm_ms=round(ems, 3), encode_host_ms=round(hms, 3), decode_hbm_ms=round(dms, 3),
                              decode_host_ms=round(hdms, 3), pcie_floor_ms=round(mb / 56e3 * 1e3, 3))), flush=True)
