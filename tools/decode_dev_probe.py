"""Device-resident decode timings with the host-side stage marks (CNIIC_TRACE_HOST=1): tools only."""
import sys, os, time, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cniic_amd
from cniic_amd import _lib, synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
which = sys.argv[2:] or ["cluster-colors(256)", "delta", "hufman"]
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev); ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img)
out = torch.empty(size * size * 12 + (1 << 24), dtype=torch.uint8, device=dev)
back = torch.empty(size * size * 3, dtype=torch.uint8, device=dev)
for expr in which:
    rc, ln, st = ctx.encode(expr, img, w=size, h=size, out=out, allow=(_lib.FEW_ACTIVE,))
    torch.cuda.synchronize()
    for rep in range(3):
        t = time.perf_counter(); rc2, w, h = ctx.decode_into(expr, out, ln, back); torch.cuda.synchronize(); dt = time.perf_counter() - t
    ok = bool(torch.equal(back.view(size, size, 3), img)) if expr in ("delta", "hufman") else None
    print(json.dumps(dict(codec=expr, size=size, bytes=ln, decode_ms=round(dt * 1e3, 3), lossless_ok=ok, rc=rc2)), flush=True)
