#!/usr/bin/env python3
"""Every codec both ways on RECTANGLES that are no power of two (the reference's data set, DIV2K, is ~2040 x 1356; frames are 1920 x 1080):
the generalised Hilbert scan, the per-position gather, ragged tiles.  One line each (ms, Mpixels/s).  Tools only."""
import os, sys, time, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
shapes = [(2040, 1356), (1920, 1080), (4000, 3000), (2048, 2048), (4096, 4096), (4096, 2048), (8000, 6000)]
codecs = ["cluster-colors(256)", "voronoi(64)", "hufman", "delta", "hilbert(rle)"]
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
for w, h in shapes:
    img = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 11, w, h, out=img)
    out = torch.empty(w * h * 13 + (1 << 22), dtype=torch.uint8, device=dev)
    back = torch.empty(w * h * 3, dtype=torch.uint8, device=dev)
    for expr in codecs:
        if expr.startswith("voronoi") and w * h > 20e6: continue
        try:
            ems, (rc, n, st) = t(lambda: ctx.encode(expr, img, w=w, h=h, out=out))
            dms, _ = t(lambda: ctx.decode_into(expr, out, n, back))
            print(json.dumps(dict(w=w, h=h, codec=expr, encode_ms=round(ems, 3), decode_ms=round(dms, 3), enc_mpx_s=round(w * h / ems / 1e3), dec_mpx_s=round(w * h / dms / 1e3),
                                  iterations=st.get("iterations"))), flush=True)
        except Exception as e:
            print(json.dumps(dict(w=w, h=h, codec=expr, error=str(e)[:100])), flush=True)
