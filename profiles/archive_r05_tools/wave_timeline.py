#!/usr/bin/env python3
"""Per-wave timeline of ONE launch of the K-means assign kernel (a -DCNIIC_RGBW_PHASES build; CNIIC_DBG_TIMELINE): every wave
writes the 100 MHz clock at entry, after the prologue (folded-in update), after its first tests / first candidate build, at the
end of its cell loop, after the block barrier and at its end.  Prints where the launch's span goes.

    python tools/wave_timeline.py <launch> [<launch> ...]
"""
import os
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cniic_amd  # noqa: E402
from cniic_amd import _lib, synth  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
size, K = 4096, 256
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img)
out = torch.empty(size * size * 4 + (1 << 20), dtype=torch.uint8, device=dev)
expr = "cluster-colors(%d)" % K
ctx.encode(expr, img, w=size, h=size, out=out)
block_end = {}
for L in sys.argv[1:]:
    path = "/tmp/tl_%s.csv" % L
    os.environ["CNIIC_DBG_TIMELINE"] = L
    os.environ["CNIIC_DBG_TIMELINE_FILE"] = path
    ctx.encode(expr, img, w=size, h=size, out=out, flags=_lib.KM_PROFILE)
    a = np.loadtxt(path, delimiter=",", skiprows=1, dtype=np.int64)
    t = a[:, 1:7].astype(np.float64) / 100.0  # us
    t0 = t[:, 0].min()
    t -= t0
    dirty, cells = a[:, 7], a[:, 8]
    WPB = 12 if int(L) >= int(os.environ.get("CNIIC_KM_BIG_BLOCKS_FROM", "10")) else 8  # waves per block of that launch
    print("launch %s: %d waves, span %.2f us; dirty cells %d of %d tested" % (L, len(a), t[:, 5].max(), dirty.sum(), cells.sum()))
    names = ["entry", "prologue done", "first test/build done", "loop done", "barrier passed", "end"]
    for i, n in enumerate(names):
        v = t[:, i]
        v = v[a[:, 1 + i] > 0]
        if len(v):
            print("  %-22s min %6.2f  p50 %6.2f  p90 %6.2f  p99 %6.2f  max %6.2f" % (n, v.min(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max()))
    order = np.argsort(-t[:, 3])[:4]
    for i in order:
        print("    late wave %5d (block %4d): prologue done %.2f, first test/build %.2f, loop done %.2f, end %.2f | %d dirty of %d cells" %
              (a[i, 0], a[i, 0] // WPB, t[i, 1], t[i, 2], t[i, 3], t[i, 5], dirty[i], cells[i]))
    blk = (a[:, 0] // WPB).astype(np.int64)
    bend = np.zeros(blk.max() + 1)
    np.maximum.at(bend, blk, t[:, 3])
    block_end[L] = bend
    print("  blocks: loop done min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f us" % (bend.min(), np.percentile(bend, 10), np.percentile(bend, 50), np.percentile(bend, 90), bend.max()))
    if a.shape[1] >= 13 and not (dirty.sum() <= cells.sum()):
        # full schedule: what a block's loop time is made of (least squares over the blocks)
        nb = blk.max() + 1
        X = np.zeros((nb, 6))
        for col, src in enumerate((8, 7, 9, 10, 11, 12)):  # cells, sweeps, candidates, points, S builds, S lengths
            np.add.at(X[:, col], blk, a[:, src].astype(np.float64))
        bstart = np.full(nb, 1e9)
        np.minimum.at(bstart, blk, t[:, 1])
        y = bend - bstart
        A = np.column_stack([np.ones(nb), X])
        coef, *_ = np.linalg.lstsq(A, y, rcond=None)
        pred = A @ coef
        names6 = ["const", "per cell", "per sweep", "per candidate", "per point", "per S build", "per S entry"]
        print("  block loop time ~ " + ", ".join("%s %.4f" % (n, c) for n, c in zip(names6, coef)) + " (us); R^2 %.3f; block means: cells %.1f sweeps %.1f candidates %.0f points %.0f S builds %.1f" %
              (1 - ((y - pred) ** 2).sum() / ((y - y.mean()) ** 2).sum(), X[:, 0].mean(), X[:, 1].mean(), X[:, 2].mean(), X[:, 3].mean(), X[:, 4].mean()))
    is_skip = dirty.sum() <= cells.sum()   # (the full schedule counts sweeps in that slot: more than cells)
    if a.shape[1] >= 13 and is_skip and dirty.sum():  # skip schedule: the dirty cells' chain
        nd = float(dirty.sum())
        print("  per dirty cell: range + candidate build %.2f us, then waiting for its points %.2f us, sweeps %.2f us (%.1f candidates)" %
              (a[:, 9].sum() / 100.0 / nd, a[:, 10].sum() / 100.0 / nd, a[:, 11].sum() / 100.0 / nd, a[:, 12].sum() / nd))
    nb_ = len(bend)
    ids = np.arange(nb_)
    for name, key in (("block %% 8 (XCD if blocks are dealt round-robin)", ids % 8), ("block // 256 (dispatch round)", ids // 256), ("block %% 32 // 8", ids % 32 // 8)):
        ks = np.unique(key)
        print(("  end of loop by " + name + ": ") + " ".join("%.1f" % bend[key == k].mean() for k in ks))
    # how much of the variance is a smooth function of the block index (neighbouring blocks hold neighbouring cells)?
    sm = np.convolve(bend, np.ones(9) / 9, mode="same")
    print("  variance explained by a 9-block running mean over the block index: %.2f" % (1 - ((bend - sm)[8:-8] ** 2).sum() / ((bend - bend.mean())[8:-8] ** 2).sum()))
    loop = t[:, 3] - t[:, 1]
    for d in range(0, 9):
        sel = dirty == d
        if sel.sum():
            print("  waves with %d dirty cells: %5d, loop time mean %6.2f max %6.2f us" % (d, sel.sum(), loop[sel].mean(), loop[sel].max()))
Ls = list(block_end)
for i in range(len(Ls) - 1):
    x, y = block_end[Ls[i]], block_end[Ls[i + 1]]
    if len(x) == len(y):
        print("per-block end of loop, launch %s vs launch %s: correlation %.3f" % (Ls[i], Ls[i + 1], np.corrcoef(x, y)[0, 1]))
ctx.close()
