"""cluster-colors encode time at small sizes through both routes (tools only)."""
import sys, os, time, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cniic_amd
from cniic_amd import _lib, synth
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for size in (256, 512, 768, 1024, 2048):
    img = torch.empty((size, size, 3), dtype=torch.uint8, device=dev); ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 2, size, size, out=img)
    out = torch.empty(size * size * 4 + (1 << 20), dtype=torch.uint8, device=dev)
    row = {"size": size}
    for name, thr in (("partition", "0"), ("dense", str(1 << 40))):
        os.environ["CNIIC_SP_MIN_PIXELS"] = thr
        for _ in range(2):
            rc, ln, st = ctx.encode("cluster-colors(256)", img, w=size, h=size, out=out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            rc, ln, st = ctx.encode("cluster-colors(256)", img, w=size, h=size, out=out)
        torch.cuda.synchronize()
        row[name + "_ms"] = round((time.perf_counter() - t) / 5 * 1e3, 3)
        row["iters"] = st["iterations"]
    print(json.dumps(row))
