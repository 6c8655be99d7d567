import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
w, h = 8000, 6000
img = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 11, w, h, out=img)
out = torch.empty(w * h * 13 + (1 << 22), dtype=torch.uint8, device=dev)
back = torch.empty(w * h * 3, dtype=torch.uint8, device=dev)
for expr in sys.argv[1:]:
    for _ in range(4):
        rc, n, st = ctx.encode(expr, img, w=w, h=h, out=out)
        ctx.decode_into(expr, out, n, back)
    torch.cuda.synchronize()
