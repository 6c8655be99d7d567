#!/usr/bin/env python3
"""Times the other hot-path rows of SURVEY 8 on one MI355X (configs 3 and 5 of BASELINE.json):
voronoi K-means per iteration, delta encode, Hufman encode, hilbert map.  Prints one line per row."""
import json
import os
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cniic_amd
from cniic_amd import _lib, synth

dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
which = sys.argv[1:] or ["voronoi", "delta", "hufman", "delta16k", "rle", "rle16k", "rect"]


def image(size, seed):
    img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
    ctx.synth_image(_lib.SYNTH_PHOTO, seed, size, size, out=img)
    return img


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


if "voronoi" in which:
    size, K, iters = 4096, 2048, 20
    img = image(size, synth.SEED0 + 3)
    out = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    dt, (rc, ln, st) = timed(lambda: ctx.encode("voronoi(%d)" % K, img, w=size, h=size, out=out, max_iters=iters,
                                                 allow=(_lib.FEW_ACTIVE,)), reps=2)
    ms, n = ctx.kernel_time("kmeans_xyrgb_iter")
    print(json.dumps({"row": "voronoi K-means (config 3)", "size": size, "K": K, "iterations": st["iterations"],
                      "ms_per_iteration": round(dt * 1e3 / max(1, st["iterations"]), 3),
                      "GBps_algorithmic_7B_per_px": round(7.0 * size * size / (dt / max(1, st["iterations"])) / 1e9, 1),
                      "centroids_tested_per_px": round(st["pair_evals"] / max(1, st["iterations"]) / (size * size), 1), "rc": rc}))

for name, size in (("delta", 4096), ("hufman", 4096), ("delta16k", 16384), ("rle", 4096), ("rle16k", 16384)):
    if name not in which:
        continue
    expr = "delta" if name.startswith("delta") else "hilbert(rle)" if name.startswith("rle") else "hufman"
    img = image(size, synth.SEED0 + 5)
    out = torch.empty(size * size * 12 + (1 << 24), dtype=torch.uint8, device=dev)
    dt, (rc, ln, st) = timed(lambda: ctx.encode(expr, img, w=size, h=size, out=out), reps=2)
    ctx.encode(expr, img, w=size, h=size, out=out, flags=_lib.KM_PROFILE)  # one more call with the stage timers on
    extra = {}
    for k in ("hilbert_delta", "delta_gather", "delta_hist", "huff_pack", "hist_rgb"):
        ms, n = ctx.kernel_time(k)
        if n:
            extra[k + "_ms"] = round(ms / n, 3)
    print(json.dumps({"row": "%s encode %dx%d" % (expr, size, size), "ms": round(dt * 1e3, 2), "Mpx_per_s": round(size * size / dt / 1e6, 1),
                      "bytes_per_px": round(ln / (size * size), 4), **extra}))
# rectangles that are no 2^n square (the reference's data set and frames): the scan by leaves + class tables
if "rect" in which:
    for w, h in ((1920, 1080), (4000, 3000), (8000, 6000)):
        img = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
        ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 5, w, h, out=img)
        out = torch.empty(w * h * 13 + (1 << 22), dtype=torch.uint8, device=dev)
        back = torch.empty(w * h * 3, dtype=torch.uint8, device=dev)
        for expr in ("delta", "hilbert(rle)"):
            dt, (rc, ln, st) = timed(lambda: ctx.encode(expr, img, w=w, h=h, out=out), reps=3)
            dd, _ = timed(lambda: ctx.decode_into(expr, out, ln, back), reps=3)
            print(json.dumps({"row": "%s %dx%d" % (expr, w, h), "encode_ms": round(dt * 1e3, 3), "decode_ms": round(dd * 1e3, 3),
                              "encode_Mpx_per_s": round(w * h / dt / 1e6, 1), "decode_Mpx_per_s": round(w * h / dd / 1e6, 1), "bytes_per_px": round(ln / (w * h), 4)}))
        del img, out, back
ctx.close()
