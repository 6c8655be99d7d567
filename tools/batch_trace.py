"""One batch encode (64 frames 1920x1080, one palette each, 8 worker streams) for rocprofv3 --kernel-trace: how busy is the GPU?
    rocprofv3 --kernel-trace --output-format csv -d out -o bt -- python3 tools/batch_trace.py ; python3 tools/batch_trace.py --summarise out"""
import os, sys, time, glob, csv, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    rows = []
    for fn in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(fn)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last batch encode: from the marker kernel (k_synth... no: take the last 40 % of the timeline)
    t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
    lo = t1 - int(os.environ.get("WINDOW_NS", "40000000"))
    sel = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
    busy = collections.defaultdict(int)
    tot = 0
    ev = []
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy[r["Kernel_Name"][:60]] += e - s
        tot += e - s
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, last, any_busy, hist = 0, lo, 0, collections.defaultdict(int)
    for t, d in ev:
        hist[depth] += t - last
        if depth > 0:
            any_busy += t - last
        depth += d; last = t
    span = t1 - lo
    print("window %.2f ms: %d kernels, sum of durations %.2f ms (%.2f x the window), some kernel running %.1f %% of it" % (span / 1e6, len(sel), tot / 1e6, tot / span, 100.0 * any_busy / span))
    print("kernels in flight -> share of the window:", {k: round(100.0 * v / span, 1) for k, v in sorted(hist.items())})
    for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:14]:
        print("  %-60s %8.2f ms  %5d launches" % (k, v / 1e6, sum(1 for r in sel if r["Kernel_Name"][:60] == k)))
    sys.exit(0)
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
    ctx.set_opt(_lib.OPT_BATCH_STREAMS, int(os.environ.get("STREAMS", "8")))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, lens, rcs, sts = ctx.encode_batch("cluster-colors(256)", fr, W, H, F, out, stride)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("batch %d: %.3f ms per frame" % (rep, dt / F * 1e3), flush=True)
