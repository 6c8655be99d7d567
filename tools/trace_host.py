"""Host-side stage marks of one call (CNIIC_TRACE_HOST=1): where the host waits.  usage: CNIIC_TRACE_HOST=1 python3 tools/trace_host.py [codec] [w] [h] [decode]"""
import os, sys
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, cniic_amd
from cniic_amd import _lib, synth
expr = sys.argv[1] if len(sys.argv) > 1 else "cluster-colors(256)"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
h = int(sys.argv[3]) if len(sys.argv) > 3 else w
decode = len(sys.argv) > 4
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
img = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, w, h, out=img)
out = torch.empty(w * h * 13 + (1 << 20), dtype=torch.uint8, device=dev)
back = torch.empty(w * h * 3, dtype=torch.uint8, device=dev)
for _ in range(4):
    rc, n, st = ctx.encode(expr, img, w=w, h=h, out=out)
    if decode:
        sys.stderr.write("-- decode\n")
        ctx.decode_into(expr, out, n, back)
    sys.stderr.write("----\n")
ctx.close()
