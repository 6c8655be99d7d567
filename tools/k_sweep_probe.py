#!/usr/bin/env python3
"""Encode times of the two K-means codecs over K, on a photograph and on uniform noise (one MI355X): finds K-dependent pathologies.
tools/k_sweep_probe.py [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cniic_amd
from cniic_amd import _lib, synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(5)
photo = torch.empty((size, size, 3), dtype=torch.uint8, device=dev); ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=photo)
noise = torch.randint(0, 256, (size, size, 3), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty(size * size * 4 + (1 << 24), dtype=torch.uint8, device=dev)
for name, img in (("photo", photo), ("noise", noise)):
    for expr in ["cluster-colors(%d)" % k for k in (1, 2, 16, 256, 1024, 2048, 4096)] + ["voronoi(%d)" % k for k in (1, 16, 256, 2048, 4096)]:
        allow = (_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE, _lib.UNSUPPORTED) if hasattr(_lib, "UNSUPPORTED") else (_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE)
        try:
            rc, n, st = ctx.encode(expr, img, w=size, h=size, out=out, allow=allow); torch.cuda.synchronize()
            t = time.perf_counter(); rc, n, st = ctx.encode(expr, img, w=size, h=size, out=out, allow=allow); torch.cuda.synchronize()
            ms = (time.perf_counter() - t) * 1e3
            print("%-6s %-22s rc %2d  %9.2f ms  it %4d  %.3f ms/it  %.3f B/px" % (name, expr, rc, ms, st["iterations"], ms / max(1, st["iterations"]), n / (size * size)), flush=True)
        except Exception as e:
            print("%-6s %-22s %s" % (name, expr, str(e)[:100]), flush=True)
