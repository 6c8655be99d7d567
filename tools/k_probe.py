#!/usr/bin/env python3
"""The reference's K lists (README: cluster-colors 16 .. 256, voronoi 64 .. 2048) on one 4096^2 and one 1920x1080 photo-like image: encode ms,
iterations, ms per iteration.  Looks for a K that is slower than its neighbours.  Tools only."""
import os, sys, time, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
def t(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
for w, h in ((4096, 4096), (1920, 1080)):
    img = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, w, h, out=img)
    out = torch.empty(w * h * 4 + (1 << 22), dtype=torch.uint8, device=dev)
    for expr in ["cluster-colors(%d)" % k for k in (2, 8, 16, 32, 64, 128, 256, 512, 1024)] + ["voronoi(%d)" % k for k in (64, 128, 256, 512, 1024, 2048)]:
        ms, (rc, n, st) = t(lambda: ctx.encode(expr, img, w=w, h=h, out=out, allow=(_lib.FEW_ACTIVE, _lib.TOO_FEW_POINTS)))
        it = max(1, st.get("iterations") or 1)
        print(json.dumps(dict(w=w, h=h, codec=expr, rc=rc, ms=round(ms, 3), iterations=it, us_per_iteration=round(ms * 1e3 / it, 1), bytes_per_px=round(n / (w * h), 4))), flush=True)
