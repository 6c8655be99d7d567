#!/usr/bin/env python3
"""Timing experiment (a -DCNIIC_RGBW_PHASES build): the super-cell assign kernel cut short after a stage (CNIIC_SUP_STOP = 1 prologue,
2 S build, 3 classification, 4 everything but the sweeps), per-launch durations of launches 2..11."""
import os, sys
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
size = 4096
for stop in ("0", "1", "2", "3", "4"):
    os.environ["CNIIC_SUP_STOP"] = stop
    os.environ["CNIIC_KM_LAUNCH_TRACE"] = "/tmp/st.csv"
    ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
    ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img)
    out = torch.empty(size * size * 4 + (1 << 20), dtype=torch.uint8, device=dev)
    for _ in range(2):
        ctx.encode("cluster-colors(256)", img, w=size, h=size, out=out, flags=_lib.KM_PROFILE, max_iters=12, allow=(_lib.FEW_ACTIVE, _lib.HIP))
    rows = open("/tmp/st.csv").read().strip().split("\n")[1:]
    print("stop", stop, " ".join(r.split(",")[1] for r in rows[:14]))
    ctx.close()
