#!/usr/bin/env python3
# tests/fuzz_codecs.py [seconds] -- random images (sizes, smooth / ramps / noise / flat patches) through `delta`, `hufman` and
# `hilbert(rle)` on the GPU against the oracle (tests/oracle_lib.py), bytes and round trip; with the knobs that move the
# routes (16-bit / 32-bit delta stream, tile / per-position gather and linearise, host / GPU Huffman codes).
import os, sys, time
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import oracle_lib as O

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
KNOBS = [{}, {"CNIIC_DELTA_ROUTE": "32"}, {"CNIIC_DELTA_GATHER": "any", "CNIIC_HILBERT_MOVE": "any"}, {"CNIIC_HUF_GPU_CODES_MIN": "0"},
         {"CNIIC_HUF_GPU_CODES_MIN": "0", "CNIIC_HUF_RUNS_MIN": "0"},   # round 3: the tree from runs of equal count whatever the alphabet's size
         {"CNIIC_HUF_GPU_CODES_MIN": "0", "CNIIC_TEST_INLINE_CODE_BITS": "7"}, {"CNIIC_TEST_PACK_IMG_WORDS": "30"},
         {"CNIIC_GPU_DECODE_MIN": "0"}, {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_TEST_TRIE_GPU": "1"},   # round 3: the parallel decoder / the GPU trie parse whatever the size
         {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HILBERT_MOVE": "any"},
         {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HD_PHASES": "1"},                              # round 3: every stream through the phase maps
         {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_TEST_TRIE_GPU": "1", "CNIIC_HD_LUT2_BITS": "21"},  # the second table built from the leaves' side
         {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HD_LUT2_BITS": "13"},
         {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HD_HOPELESS_PCT": "1"}, {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HD_HOPELESS_PCT": "0"},   # pass 0's verdict: eager / never
         # round 5: the decoder with / without its kept symbols, third tables and first table, short and long warm-ups
         {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HD_KEEP": "1"}, {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HD_KEEP": "0", "CNIIC_HD_LUT3": "0"},
         {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HD_KEEP": "1", "CNIIC_HD_LUT1": "0", "CNIIC_HD_WARM": "32"}, {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HD_LUT1": "1", "CNIIC_HD_WARM": "480"},
         {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HD_KEEP": "1", "CNIIC_HD_LUT2_BITS": "13", "CNIIC_HD_LUT1": "0"}, {"CNIIC_GPU_DECODE_MIN": "0", "CNIIC_HD_KEEP": "1", "CNIIC_HD_PHASES": "1"},
         {"CNIIC_SCAN_LEAVES_MIN": "0"}, {"CNIIC_SCAN_LEAVES_MIN": "0", "CNIIC_SCAN_LEAF_AREA": "7"},   # the scan of rectangles from leaves + class tables
         {"CNIIC_SCAN_LEAVES_MIN": "0", "CNIIC_SCAN_LEAF_AREA": "64", "CNIIC_GPU_DECODE_MIN": "0"}]


def image():
    if rng.random() < 0.5:
        s = int(2 ** rng.integers(0, 10)); h = w = s
    else:
        h, w = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    y, x = np.mgrid[0:h, 0:w]
    kind = rng.integers(0, 5)
    if kind == 0:
        img = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:
        img = np.stack([x // 2 + y // 3, x // 3 + y, (x + y) // 4], axis=2) + rng.integers(-3, 4, (h, w, 3))
    elif kind == 2:
        img = np.stack([x * 3, y * 5, x ^ y], axis=2)
    elif kind == 3:
        img = np.full((h, w, 3), rng.integers(0, 256)) + (rng.random((h, w, 1)) < 0.02) * rng.integers(0, 255, (h, w, 3))
    else:
        img = (np.stack([x, y, x + y], axis=2) // int(rng.integers(1, 40))) * int(rng.integers(1, 9))
    return (img & 255).astype(np.uint8)


try:
    import torch as TORCH
    if not TORCH.cuda.is_available():
        TORCH = None
except Exception:
    TORCH = None


def run(ctx, budget):
    """`budget` seconds of random cases on ctx; returns how many were checked (an assertion stops at the first difference)"""
    t0, cases, said = time.time(), 0, time.time()
    while time.time() - t0 < budget:
        if time.time() - said > 60:   # (a long run says that it is alive)
            said = time.time()
            sys.stderr.write("fuzz: %d cases after %.0f s\n" % (cases, said - t0)); sys.stderr.flush()
        img = image()
        knob = KNOBS[int(rng.integers(0, len(KNOBS)))]
        saved = {k: os.environ.get(k) for k in knob}
        os.environ.update(knob)
        try:
          try:
            for expr in ("delta", "hufman", "hilbert(rle)"):
                rc, data, _ = ctx.encode(expr, img)
                erc, edata, _ = O.encode(expr, img)
                assert rc == erc == 0 and data == edata, (expr, img.shape, knob, "encode")
                rc, back = ctx.decode(expr, data)
                assert rc == 0 and np.array_equal(back, img), (expr, img.shape, knob, "decode")
                if TORCH is not None:   # the stream in HBM at a random byte alignment, the image into HBM; and a random cut of it
                    shift = int(rng.integers(0, 4))
                    buf = TORCH.zeros(len(data) + 8, dtype=TORCH.uint8, device="cuda")
                    buf[shift:shift + len(data)] = TORCH.frombuffer(bytearray(data), dtype=TORCH.uint8).cuda()
                    out = TORCH.zeros(max(img.size, 1), dtype=TORCH.uint8, device="cuda")
                    TORCH.cuda.synchronize()   # (ctx may run on a stream of its own: torch's fills must have landed)
                    rc, dw, dh = ctx.decode_into(expr, buf[shift:], len(data), out)
                    assert rc == 0 and np.array_equal(out[:img.size].cpu().numpy().reshape(img.shape), img), (expr, img.shape, knob, "decode (HBM)")
                    cut = int(rng.integers(1, max(2, len(data))))
                    rc, _, _ = ctx.decode_into(expr, buf[shift:], len(data) - cut, out, allow=(-6, -8))
                    erc, _ = O.decode(expr, data[:len(data) - cut])
                    assert (rc == 0) == (erc == 0), (expr, img.shape, knob, "cut", cut)
                cases += 1
          except Exception:
            sys.stderr.write("fuzz_codecs: failing case: codec %s, image %s kind-independent seed state, knobs %s\n" % (expr, img.shape, knob))
            np.save("/tmp/fuzz_fail.npy", img)
            raise
        finally:
            for k, v in saved.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
    return cases


if __name__ == "__main__":
    import cniic_amd
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    # (on torch's stream: the HBM decodes read buffers torch has just filled -- a context with a stream of its own raced with that copy
    # once in ~40 000 cases: "delta: colour out of range" on a stream that was fine)
    if TORCH is not None:
        TORCH.cuda.set_stream(TORCH.cuda.Stream())   # (a context does not share the NULL stream)
    ctx = cniic_amd.Context(0, stream=TORCH.cuda.current_stream().cuda_stream) if TORCH is not None else cniic_amd.Context(0)
    n = run(ctx, seconds)
    print("fuzz_codecs: %d cases in %.0f s, all equal to the oracle" % (n, seconds))
