"""Per-iteration timeline of the persistent colour K-means launch (k_kmeans_persist.hip) on the headline workload:
    CNIIC_USE_TESTING_LIB=1 python tools/ps_trace.py [size] [K] [out.csv]
Block 0 stamps the 100 MHz clock when an iteration's centroids stand; the library writes the differences (CNIIC_KM_PS_TRACE)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "ps_trace.csv")
MAXIT = int(os.environ.get("PS_MAX_ITERS", "0"))
os.environ["CNIIC_KM_PS_TRACE"] = path
bpath = path.replace(".csv", "_blocks.csv")
if os.environ.get("PS_BLOCKS_TRACE", "1") == "1":
    os.environ["CNIIC_KM_PS_BLOCK_TRACE"] = bpath
os.environ.setdefault("CNIIC_KM_PS_REQUIRE", "1")

import torch

import cniic_amd
from cniic_amd import _lib, synth

dev = torch.device("cuda:0")
with cniic_amd.Context(0) as ctx:
    img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
    ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + int(os.environ.get('PS_SEED', '2')), size, size, out=img)
    out = torch.empty(size * size * 2, dtype=torch.uint8, device=dev)
    expr = "cluster-colors(%d)" % K
    for _ in range(3):
        rc, n, st = ctx.encode(expr, img, w=size, h=size, out=out, max_iters=MAXIT, allow=(-3,))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        rc, n, st = ctx.encode(expr, img, w=size, h=size, out=out, max_iters=MAXIT, allow=(-3,))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("encode %.3f ms, %d bytes, %s" % (dt * 1e3, n, st))
rows = [l.strip().split(",") for l in open(path)][1:]
us = [float(r[1]) for r in rows]
nm = [int(r[2]) for r in rows]
full = [u for u, m in zip(us[1:], nm[1:]) if m > 64 or m < 0]
skip = [u for u, m in zip(us[1:], nm[1:]) if 0 <= m <= 64]
print("iterations %d, loop %.1f us: first %.1f us | full x%d mean %.2f us | skip x%d mean %.2f us" % (
    len(us), sum(us), us[0], len(full), sum(full) / max(1, len(full)), len(skip), sum(skip) / max(1, len(skip))))
print(" ".join("%.1f" % u for u in us))

if os.path.exists(bpath):
    import collections
    per = collections.defaultdict(list)
    lines = list(open(bpath))
    info = {}
    if lines and lines[0].startswith("#"):
        for part in lines[0].split(":", 1)[1].split(";"):
            q = part.split()
            if len(q) == 4:
                info[int(q[0])] = (int(q[1]), int(q[2]), int(q[3]))
        lines = lines[1:]
    blk = collections.defaultdict(list)
    for l in lines[1:]:
        f = l.strip().split(",")
        blk[int(f[0])].append((int(f[1]), float(f[2]), float(f[9])))
        per[int(f[1])].append([float(x) for x in f[2:]])
    print("iteration: assign mean / max = lists + classify + sweeps (mean; cells swept mean / max) | flush mean | barrier min / mean | update mean   (us, over the blocks)")
    for i in sorted(per):
        v = per[i]
        n = len(v)
        m = lambda k: sum(x[k] for x in v) / n
        print("%3d: assign %6.2f / %6.2f = %5.2f + %5.2f + %5.2f (%5.1f / %3d) | flush %5.2f | barrier %5.2f / %5.2f | update %5.2f" % (
            i, m(0), max(x[0] for x in v), m(4), m(5), m(6), m(7), max(x[7] for x in v), m(1), min(x[2] for x in v), m(2), m(3)))

    if info:
        # what makes a block slow?  least squares of the mean assign time of iterations 10..24 (full schedule) on cells, points and lists
        import numpy as np
        ids = sorted(info)
        A = np.array([[info[g][0], info[g][1] / 256.0, info[g][2], 1.0] for g in ids])
        y = np.array([np.mean([a_ for (i, a_, n_) in blk[g] if 10 <= i <= 24]) for g in ids])
        coef, *_ = np.linalg.lstsq(A, y, rcond=None)
        pred = A @ coef
        print("blocks: cells %d..%d (mean %.1f), points %d..%d (mean %.0f), lists %d..%d" % (A[:, 0].min(), A[:, 0].max(), A[:, 0].mean(), A[:, 1].min() * 256, A[:, 1].max() * 256, A[:, 1].mean() * 256, A[:, 2].min(), A[:, 2].max()))
        print("full-schedule assign time of a block ~ %.3f us x cells + %.3f us x (points / 256) + %.3f us x lists + %.2f us; residual rms %.2f us; times %.1f..%.1f us (mean %.1f)" % (coef[0], coef[1], coef[2], coef[3], float(np.sqrt(np.mean((pred - y) ** 2))), y.min(), y.max(), y.mean()))
