import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")
import torch
import cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda:0")
F, W, H = 64, 1920, 1080
with cniic_amd.Context(0) as ctx:
    fr = torch.empty((F, H, W, 3), dtype=torch.uint8, device=dev)
    for f in range(F):
        ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 4 + f, W, H, out=fr[f])
    stride = W * H
    out = torch.empty(stride * F, dtype=torch.uint8, device=dev)
    for streams in (8, 4, 2, 1):
        ctx.set_opt(_lib.OPT_BATCH_STREAMS, streams)
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc, lens, rcs, sts = ctx.encode_batch("cluster-colors(256)", fr, W, H, F, out, stride)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("streams %d: %.3f ms per frame, %.0f Mpx/s, iterations mean %.1f, rc %s" % (streams, dt / F * 1e3, F * W * H / dt / 1e6, sum(s["iterations"] for s in sts) / F, set(rcs)))
    # one frame alone
    o1 = torch.empty(stride, dtype=torch.uint8, device=dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, n, st = ctx.encode("cluster-colors(256)", fr[0], w=W, h=H, out=o1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("one frame alone: %.3f ms, %s" % (dt * 1e3, st))
