#!/usr/bin/env python3
"""Per-launch trace of the K-means assign kernel over one cluster-colors encode (HIP start/stop events attached to every
dispatch, CNIIC_KM_PROFILE): one CSV line per launch -- number, duration in us, points its iteration moved, class
(first / iteration / final-update / no-op).  The table profiles/rNN_launch_trace.csv and bench.py's roofline.by_class
come from this.

    python tools/launch_trace.py out.csv [size] [K]
"""
import os
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out_csv = os.path.abspath(sys.argv[1])
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
os.environ["CNIIC_KM_LAUNCH_TRACE"] = out_csv

import torch  # noqa: E402

import cniic_amd  # noqa: E402
from cniic_amd import _lib, synth  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_opt(_lib.OPT_KM_LOOP, 1)   # (round 5: the loop as launches -- the default is ONE persistent launch, tools/ps_trace.py)
img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img)
out = torch.empty(size * size * 4 + (1 << 20), dtype=torch.uint8, device=dev)
expr = "cluster-colors(%d)" % K
for _ in range(2):
    ctx.encode(expr, img, w=size, h=size, out=out)
if os.environ.get("TRACE_DBG"):
    k, v = os.environ["TRACE_DBG"].split("=")
    os.environ[k] = v
try:
    rc, ln, st = ctx.encode(expr, img, w=size, h=size, out=out, flags=_lib.KM_PROFILE)
except Exception as ex:  # (an experiment that breaks the result on purpose: the trace is written before the codec notices)
    print("encode failed:", ex)
    st = {"iterations": 0}
ms, n = ctx.kernel_time("kmeans_rgbw_assign")
wms, wn = ctx.kernel_time("kmeans_rgbw_assign_working")
print("iterations %d, %d launches %.1f us total (%.2f us each); %d working launches %.1f us (%.2f us each)" %
      (st["iterations"], n, ms * 1e3, ms * 1e3 / max(1, n), wn, wms * 1e3, wms * 1e3 / max(1, wn)))
if os.path.exists(out_csv):
    rows = open(out_csv).read().strip().split("\n")[1:]
    print(" ".join("%s:%s" % (r.split(",")[0], r.split(",")[1]) for r in rows))
ctx.close()
