#!/usr/bin/env python3
"""Encode / decode times of every codec on images that are NOT photographs (one MI355X): flat, two colours, a ramp, uniform noise, a photo with a
flat half.  Finds pathologies of data-dependent paths (one crowded colour bucket, one symbol, all-distinct colours).  tools/adversarial_probe.py [size]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cniic_amd
from cniic_amd import _lib, synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(5)
def photo():
    img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev); ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img); return img
def flat(): return torch.full((size, size, 3), 77, dtype=torch.uint8, device=dev)
def two():
    img = torch.zeros((size, size, 3), dtype=torch.uint8, device=dev); img[::2, 1::2] = 255; img[1::2, ::2] = 255; return img
def ramp():
    x = torch.arange(size, device=dev); y = torch.arange(size, device=dev)
    return torch.stack([(x[None, :] % 256).expand(size, size), (y[:, None] % 256).expand(size, size), ((x[None, :] + y[:, None]) // 32 % 256)], dim=2).to(torch.uint8).contiguous()
def noise(): return torch.randint(0, 256, (size, size, 3), dtype=torch.uint8, device=dev, generator=g)
def half():
    img = photo(); img[size // 2:] = 200; return img
def distinct():   # every pixel a colour of its own (as far as 2^24 colours go)
    i = torch.arange(size * size, device=dev, dtype=torch.int64)
    v = (i * 2654435761) % (1 << 24)
    return torch.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], dim=1).to(torch.uint8).reshape(size, size, 3).contiguous()
def stripes():
    img = torch.zeros((size, size, 3), dtype=torch.uint8, device=dev); img[:, ::2] = 255; return img
def noise6():    # noise in the low six bits: few hundred thousand colours of nearly equal counts
    return torch.randint(0, 64, (size, size, 3), dtype=torch.uint8, device=dev, generator=g) * 3
def blocks():    # 64 x 64 blocks of few colours
    x = torch.arange(size, device=dev) // 64; y = torch.arange(size, device=dev) // 64
    v = ((x[None, :] * 7 + y[:, None] * 13) % 5 * 50).to(torch.uint8)
    return torch.stack([v, v // 2, 255 - v], dim=2).contiguous()
out = torch.empty(size * size * 16 + (1 << 24), dtype=torch.uint8, device=dev)
back = torch.empty(size * size * 3, dtype=torch.uint8, device=dev)
which = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, mk in (("photo", photo), ("flat", flat), ("two colours", two), ("ramp", ramp), ("noise", noise), ("half flat", half), ("distinct", distinct),
                 ("stripes", stripes), ("blocks", blocks), ("noise6", noise6)):
    if which and name not in which: continue
    img = mk(); torch.cuda.synchronize()
    row = {}
    for expr in ("cluster-colors(256)", "voronoi(256)", "delta", "hufman", "hilbert(rle)"):
        allow = (_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE)
        rc, n, st = ctx.encode(expr, img, w=size, h=size, out=out, allow=allow); torch.cuda.synchronize()
        if rc not in (0, _lib.FEW_ACTIVE):
            row[expr] = "rc %d" % rc; continue
        t = time.perf_counter()
        for _ in range(3): rc, n, st = ctx.encode(expr, img, w=size, h=size, out=out, allow=allow)
        torch.cuda.synchronize(); enc = (time.perf_counter() - t) / 3 * 1e3
        rcd, dw, dh = ctx.decode_into(expr, out, n, back); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): rcd, dw, dh = ctx.decode_into(expr, out, n, back)
        torch.cuda.synchronize(); dec = (time.perf_counter() - t) / 3 * 1e3
        ok = rcd == 0 and (expr.startswith("cluster") or expr.startswith("voronoi") or bool(torch.equal(back[:size * size * 3], img.reshape(-1))))
        row[expr] = "enc %.2f ms  dec %.2f ms  %.3f B/px%s%s" % (enc, dec, n / (size * size), "" if ok else "  ROUND TRIP WRONG", "  it %d" % st["iterations"] if (expr.startswith("cluster") or expr.startswith("voronoi")) else "")
    print(name); [print("   %-20s %s" % (k, v)) for k, v in row.items()]
