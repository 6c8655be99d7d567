#!/usr/bin/env python3
"""Every codec, encode and decode, device-resident, at a few sizes and both synthetic generators: one line each (ms).  Finds what nobody
timed -- `hilbert(rle)` decode was 1.2 s at 16384^2 until this existed.  Tools only."""
import os, sys, time, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["512", "4096"])]
codecs = ["cluster-colors(256)", "voronoi(64)", "hufman", "delta", "hilbert(rle)"]
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
for size in sizes:
    for kind, kname in ((_lib.SYNTH_PHOTO, "P"), (_lib.SYNTH_UNIFORM, "U")):
        img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
        ctx.synth_image(kind, synth.SEED0 + 11, size, size, out=img)
        out = torch.empty(size * size * 13 + (1 << 22), dtype=torch.uint8, device=dev)
        back = torch.empty(size * size * 3, dtype=torch.uint8, device=dev)
        for expr in codecs:
            if kname == "U" and expr.startswith("voronoi") and size > 2048: continue
            try:
                ems, (rc, n, st) = t(lambda: ctx.encode(expr, img, w=size, h=size, out=out, allow=(_lib.FEW_ACTIVE, _lib.TOO_FEW_POINTS)))
                if rc != 0: print(json.dumps(dict(size=size, image=kname, codec=expr, rc=rc))); continue
                dms, _ = t(lambda: ctx.decode_into(expr, out, n, back))
                print(json.dumps(dict(size=size, image=kname, codec=expr, bytes=int(n), encode_ms=round(ems, 3), decode_ms=round(dms, 3), iterations=st.get("iterations"))), flush=True)
            except Exception as e:
                print(json.dumps(dict(size=size, image=kname, codec=expr, error=str(e)[:100])), flush=True)
        t0 = time.perf_counter(); m = ctx.mse(img.cpu().numpy()[:min(size, 1024), :min(size, 1024)], img.cpu().numpy()[:min(size, 1024), :min(size, 1024)]); print(json.dumps(dict(size=size, image=kname, mse_1024_ms=round((time.perf_counter() - t0) * 1e3, 2))))
