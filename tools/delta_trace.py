import sys, os
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/cniic_amd") else os.getcwd())
import torch, cniic_amd
from cniic_amd import _lib, synth
size=int(sys.argv[1])
dev=torch.device("cuda",0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx=cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
img=torch.empty((size,size,3),dtype=torch.uint8,device=dev); ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0+5, size,size,out=img)
out=torch.empty(size*size*4+(1<<24),dtype=torch.uint8,device=dev)
for i in range(3):
    ctx.encode("delta",img,w=size,h=size,out=out); torch.cuda.synchronize()
