"""Times count_freqs on an image resident in HBM (tools only)."""
import sys, os, time, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cniic_amd
from cniic_amd import _lib, synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev)
ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img)
for rep in range(3):
    t = time.perf_counter(); keys, counts = ctx.hist_rgb24(img, npx=size * size); dt = time.perf_counter() - t
print(json.dumps(dict(size=size, unique=int(keys.size), total=int(counts.astype(np.uint64).sum()), ok=bool(int(counts.astype(np.uint64).sum()) == size * size), sec=round(dt, 4), sha=int(np.bitwise_xor.reduce((keys.astype(np.uint64) * 2654435761 + counts.astype(np.uint64)) & 0xffffffff)))))
