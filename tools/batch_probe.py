"""Throughput of cniic_codec_encode_batch (one palette per image) for several worker-stream counts / grid caps: tools only."""
import sys, os, time, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cniic_amd
from cniic_amd import _lib, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 1920, 1080
dev = torch.device("cuda", 0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
frames = torch.empty((F, H, W, 3), dtype=torch.uint8, device=dev)
for f in range(F):
    ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 4 + f, W, H, out=frames[f])
stride = W * H
out = torch.empty(stride * F, dtype=torch.uint8, device=dev)
k, c = ctx.hist_rgb24(frames[0], npx=W * H)
print("distinct colours of frame 0:", k.size, flush=True)
t = time.perf_counter(); ctx.encode("cluster-colors(256)", frames[0], w=W, h=H, out=out); ctx.encode("cluster-colors(256)", frames[0], w=W, h=H, out=out); torch.cuda.synchronize()
t = time.perf_counter(); rc, ln, st = ctx.encode("cluster-colors(256)", frames[0], w=W, h=H, out=out); torch.cuda.synchronize(); print("one frame alone: %.3f ms, %d iterations" % ((time.perf_counter() - t) * 1e3, st["iterations"]), flush=True)
for streams in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "4", "8", "16"])]:
    ctx.set_opt(_lib.OPT_BATCH_STREAMS, streams)
    ctx.encode_batch("cluster-colors(256)", frames, W, H, F, out, stride)
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.encode_batch("cluster-colors(256)", frames, W, H, F, out, stride)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(json.dumps(dict(streams=streams, max_blocks=os.environ.get("CNIIC_KM_MAX_BLOCKS"), ms_per_frame=round(dt / F * 1e3, 4), Mpx_s=round(F * W * H / dt / 1e6, 1))), flush=True)
