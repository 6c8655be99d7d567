import os, sys, ctypes as C
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
size = 16384
img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 5, size, size, out=img)
syms = torch.empty(size * size, dtype=torch.int32, device=dev)
L = _lib.lib()
for _ in range(3):
    ctx._check(L.cniic_hilbert_delta(ctx.h, C.c_void_p(img.data_ptr()), C.c_uint32(size), C.c_uint32(size), C.c_void_p(syms.data_ptr())))
out = torch.empty(size * size * 4 + (1 << 26), dtype=torch.uint8, device=dev)
for _ in range(2):
    ctx.encode("delta", img, w=size, h=size, out=out)
ctx.close()
