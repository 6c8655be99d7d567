import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
size = 4096
img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img)
out = torch.empty(size * size * 4 + (1 << 20), dtype=torch.uint8, device=dev)
for _ in range(4):
    ctx.encode("cluster-colors(256)", img, w=size, h=size, out=out)
    sys.stderr.write("----\n")
ctx.close()
