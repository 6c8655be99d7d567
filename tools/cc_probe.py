"""One cluster-colors encode of the headline image (for rocprofv3 passes; tools only)."""
import sys, os, time, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cniic_amd
from cniic_amd import _lib, synth
size = int(sys.argv[1]); K = int(sys.argv[2]); cap = int(sys.argv[3]); warm = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev); ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img)
out = torch.empty(size * size * 2 + (1 << 20), dtype=torch.uint8, device=dev)
for _ in range(warm):
    ctx.encode("cluster-colors(%d)" % K, img, w=size, h=size, out=out, max_iters=2); torch.cuda.synchronize()
t = time.perf_counter(); rc, ln, st = ctx.encode("cluster-colors(%d)" % K, img, w=size, h=size, out=out, max_iters=cap); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(json.dumps(dict(size=size, K=K, rc=rc, iters=st["iterations"], moved_last=st["moved_last"], sec=round(dt, 4), bytes=ln)))
