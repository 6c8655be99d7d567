"""The reference's batch semantics (bench.rs:24-35: one palette per image) on 1080p frames: cniic_codec_encode_batch with its workers'
K-means as persistent launches inside a CU budget (CNIIC_KM_PS_BATCH_PCT, testing build) against one launch per iteration
(CNIIC_KM_PS_BATCH=0), over the number of worker streams.   python tools/batch_probe.py [frames]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")
import torch
import cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda:0")
F, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 1920, 1080
with cniic_amd.Context(0) as ctx:
    fr = torch.empty((F, H, W, 3), dtype=torch.uint8, device=dev)
    for f in range(F):
        ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 4 + f, W, H, out=fr[f])
    stride = W * H
    out = torch.empty(stride * F, dtype=torch.uint8, device=dev)
    ref = None
    for mode, pct in (("launches", 0),) + ((("persistent", 75), ("persistent", 100)) if os.environ.get("BATCH_PERSIST") else ()):   # (persistent: with profiles/r05_batch_persistent_budget.patch applied)
        os.environ["CNIIC_KM_PS_BATCH"] = "0" if mode == "launches" else "1"
        os.environ["CNIIC_KM_PS_BATCH_PCT"] = str(pct)
        for streams in (4, 6, 8, 12, 16):
            ctx.set_opt(_lib.OPT_BATCH_STREAMS, streams)
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                rc, lens, rcs, sts = ctx.encode_batch("cluster-colors(256)", fr, W, H, F, out, stride)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            got = [bytes(out[f * stride:f * stride + lens[f]].cpu().numpy()) for f in range(0, F, max(1, F // 4))]
            if ref is None:
                ref = got
            print("%-10s budget %3d %%, %2d streams: %.3f ms per frame, %6.0f Mpx/s, iterations mean %.1f, rc %s, same bytes %s" % (mode, pct, streams, dt / F * 1e3, F * W * H / dt / 1e6, sum(s["iterations"] for s in sts) / F, set(rcs), got == ref), flush=True)
    o1 = torch.empty(stride, dtype=torch.uint8, device=dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, n, st = ctx.encode("cluster-colors(256)", fr[0], w=W, h=H, out=o1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("one frame alone: %.3f ms, %s" % (dt * 1e3, st))
