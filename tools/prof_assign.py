#!/usr/bin/env python3
"""Runs only the K-means assign kernel of the headline workload (for rocprofv3 --pmc passes).
usage: prof_assign.py [size] [K] [reps] [flags]"""
import ctypes as C
import os
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cniic_amd
from cniic_amd import _lib, synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img)
keys, counts = ctx.hist_rgb24(img, npx=size * size)
U = int(keys.size)
kd = torch.from_numpy(keys.astype(np.uint32).view(np.int32)).to(dev)
wd = torch.from_numpy(counts.astype(np.uint32).view(np.int32)).to(dev)
L = _lib.lib()
km = C.c_void_p()
o = _lib.KmOpts(0, 0, flags, 0)
ctx._check(L.cniic_km_create_rgbw(ctx.h, C.c_void_p(kd.data_ptr()), C.c_void_p(wd.data_ptr()), C.c_uint64(U), C.c_uint32(0),
                                  C.c_uint32(1), C.c_uint32(K), C.byref(o), None, C.byref(km)))
ctx._check(L.cniic_km_begin(km))
for _ in range(5):
    ctx._check(L.cniic_km_assign(km))
    ctx._check(L.cniic_km_update(km, None))
ms = C.c_double(0)
ctx._check(L.cniic_km_time_assign(km, C.c_int32(reps), C.byref(ms)))
print("U=%d K=%d assign %.2f us/launch  -> %.1f GB/s algorithmic (10 B/colour)" % (U, K, ms.value * 1e3, 10.0 * U / (ms.value * 1e-3) / 1e9))
L.cniic_km_destroy(km)
ctx.close()
