import sys, os, time, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cniic_amd
from cniic_amd import _lib, synth
size=int(sys.argv[1]); K=int(sys.argv[2]); cap=int(sys.argv[3])
dev=torch.device("cuda",0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx=cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
img=torch.empty((size,size,3),dtype=torch.uint8,device=dev); ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0+3, size,size,out=img)
out=torch.empty(1<<20,dtype=torch.uint8,device=dev)
ctx.encode("voronoi(%d)"%K,img,w=size,h=size,out=out,max_iters=2,allow=(_lib.FEW_ACTIVE,)); torch.cuda.synchronize()
t=time.perf_counter(); rc,ln,st=ctx.encode("voronoi(%d)"%K,img,w=size,h=size,out=out,max_iters=cap,allow=(_lib.FEW_ACTIVE,)); torch.cuda.synchronize(); dt=time.perf_counter()-t
print(json.dumps(dict(size=size,K=K,rc=rc,iters=st["iterations"],moved_last=st["moved_last"],sec=round(dt,5),ms_per_iter=round(dt*1e3/max(1,st["iterations"]),3),cand_per_px=round(st["pair_evals"]/max(1,st["iterations"])/(size*size),1),reseeds=st["empty_reseeds"],active=st["active"])))
