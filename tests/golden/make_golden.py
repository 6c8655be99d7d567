#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- small input/expected-output vectors for the hot path.

The reference (Rust) cannot be built or run in the build image (no rustc/cargo, crates not
vendored), so these vectors are produced by the CPU oracle (oracle/liboracle.so), which is itself
pinned by the reference's own known-answer tests (tests/test_oracle_kat.py).  They freeze the
oracle's behaviour: test_golden.py checks the oracle still reproduces them (CPU) and that the HIP
library reproduces them byte for byte (GPU) without needing the oracle at run time.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from cniic_amd import synth  # noqa: E402


def keys_of(img):
    p = img.reshape(-1, 3).astype(np.uint32)
    return (p[:, 0] << 16) | (p[:, 1] << 8) | p[:, 2]


def main():
    out = {}
    # images: photo-like and uniform synthetic, odd sizes on purpose
    imgs = {"photo_96x64": synth.photo(96, 64, synth.SEED0 + 1), "uniform_40x33": synth.uniform(40, 33, synth.SEED0 + 1),
            "photo_64x64": synth.photo(64, 64, synth.SEED0 + 3)}
    for name, img in imgs.items():
        out["img_" + name] = img
        for expr in ("hufman", "delta", "hilbert(rle)", "cluster-colors(16)", "voronoi(8)"):
            rc, data, st = O.encode(expr, img, mode=O.MODE_L)
            assert rc == 0, (name, expr, rc)
            out["enc_%s_%s" % (name, expr)] = np.frombuffer(data, np.uint8)
            out["iters_%s_%s" % (name, expr)] = np.array([st["iterations"]], np.uint64)
            rc, back = O.decode(expr, data)
            assert rc == 0
            out["dec_%s_%s" % (name, expr)] = back
        k, c = O.count_freqs(keys_of(img))
        out["hist_keys_" + name] = k
        out["hist_counts_" + name] = c
        lin = O.hilbert_linearize(img)
        out["delta_syms_" + name] = O.delta_diff(lin)
    for (w, h) in ((16, 16), (13, 8), (5, 31)):
        out["hilbert_%dx%d" % (w, h)] = O.hilbert_iter(w, h)
    # K-means on the colours of one image, full run (mode L)
    img = imgs["photo_64x64"]
    k, c = O.count_freqs(keys_of(img))
    pts = np.stack([(k >> 16) & 255, (k >> 8) & 255, k & 255], axis=1).astype(np.int32)
    rc, r = O.kmeans(O.PT_RGBW, O.MODE_L, pts, c.astype(np.uint32), 32)
    assert rc == 0
    out["km_rgbw_centroids"] = r["centroids"].astype(np.uint8)
    out["km_rgbw_labels"] = r["labels"]
    out["km_rgbw_members"] = r["members"]
    out["km_rgbw_iters"] = np.array([r["stats"]["iterations"]], np.uint64)
    np.savez_compressed(os.path.join(HERE, "hotpath_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "hotpath_golden.npz"), "with", len(out), "arrays")


if __name__ == "__main__":
    main()
