import os, sys, time
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
size = 4096
img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img)
himg = img.cpu().numpy()
out_d = torch.empty(size*size*4, dtype=torch.uint8, device=dev)
out_h = np.empty(size*size*2, np.uint8)
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print("dev->dev   %.2f ms" % t(lambda: ctx.encode("cluster-colors(256)", img, w=size, h=size, out=out_d)))
print("host->dev  %.2f ms" % t(lambda: ctx.encode("cluster-colors(256)", himg, out=out_d)))
print("dev->host  %.2f ms" % t(lambda: ctx.encode("cluster-colors(256)", img, w=size, h=size, out=out_h)))
print("host->host %.2f ms" % t(lambda: ctx.encode("cluster-colors(256)", himg, out=out_h)))
print("host->host(own out) %.2f ms" % t(lambda: ctx.encode("cluster-colors(256)", himg)))
x = torch.empty(size*size*3, dtype=torch.uint8, device=dev)
hp = torch.from_numpy(himg.reshape(-1))
print("torch H2D pageable 48MB %.2f ms" % t(lambda: x.copy_(hp)))
pp = hp.pin_memory()
print("torch H2D pinned 48MB %.2f ms" % t(lambda: x.copy_(pp, non_blocking=True)))
t0=time.perf_counter(); b = himg.copy(); print("memcpy 48MB 1 thread %.2f ms" % ((time.perf_counter()-t0)*1e3))
