#!/usr/bin/env python3
"""Writes tests/golden/fullsize_digests.json: the ORACLE's results at BASELINE.json's full sizes.

Run once in the build container (CPU only, no GPU, no product code in the loop except the numpy synthetic
generator that the tests pin against the GPU generator byte for byte):

    python tests/golden/make_fullsize_digests.py [--only c2,c2r,c5,c4,v1024,v2048,v4096] [--threads 8]

Every entry is the oracle's (oracle/*.c, mode L unless it says mode R) stream for one workload:
SHA-256, length, K-means iteration count.  `tests/test_gpu_fullsize_digests.py` (-m gpu) encodes the same
workload on the HIP path and compares digests: bit-exact parity at the sizes the bench reports on
(VERDICT r03 item 1), where until now only properties were checked.

The K-means of mode L runs through oracle/kmeans_fast.c (orc_set_lloyd_threads): the same step as
orc_kmeans_step, threaded and vectorised, held to the plain loop bit for bit by tests/test_oracle_fast.py.
Mode R (the reference's pruned search, kmeans.rs:330-416) runs as it is.

Workloads (generator: cniic_amd/synth.py "P", seeds as bench.py uses them):
  c2     configs[1]  cluster-colors(256), 4096 x 4096, seed S+2                    clusterc.rs:18-53
  c2r    the same image through mode R: bytes and MSE (the K7 band at full size)    kmeans.rs:150-323
  c5     configs[4]  delta, 16384 x 16384, seed S+5                                 hilbertc.rs:405-415
  c4     configs[3]  one GPU's share: 128 frames 1920 x 1080 (seed S+4+f), ONE palette: the stacked
         frames as one 1920 x 138240 image, then every frame's own Hufman stream of its rows of the
         reduced image (clusterc.rs:31-52 applied per frame)
  v1024, v2048, v4096   configs[2]  voronoi(2048), seed S+3, at 1024^2 / 2048^2 / 4096^2   clusterc.rs:148-166
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle_lib as O  # noqa: E402
from cniic_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "fullsize_digests.json")
S = synth.SEED0


def photo_rows(w, h, seed, y0, y1):
    """rows y0..y1 of synth.photo(w, h, seed) without the whole image's temporaries (16384^2 would need tens of GB)"""
    y, x = np.mgrid[y0:y1, 0:w]
    cx, fx, cy, fy = x >> 6, x & 63, y >> 6, y & 63
    idx = (y.astype(np.uint64) * np.uint64(w) + x.astype(np.uint64))
    seed2 = np.uint64(seed) ^ np.uint64(0xD1B54A32D192ED03)
    out = np.empty((y1 - y0, w, 3), np.uint8)
    for ch in range(3):
        a, b = synth._lattice(seed, cx, cy, ch), synth._lattice(seed, cx + 1, cy, ch)
        c, d = synth._lattice(seed, cx, cy + 1, ch), synth._lattice(seed, cx + 1, cy + 1, ch)
        v = ((a * (64 - fx) + b * fx) * (64 - fy) + (c * (64 - fx) + d * fx) * fy) >> 12
        with np.errstate(over="ignore"):
            noise = (synth._mix(seed2 + synth.GAMMA * (idx * np.uint64(4) + np.uint64(ch + 1))) & np.uint64(15)).astype(np.int64) - 8
        out[..., ch] = np.clip(v + noise, 0, 255).astype(np.uint8)
    return out


def photo(w, h, seed, strip=256):
    img = np.empty((h, w, 3), np.uint8)
    for y0 in range(0, h, strip):
        img[y0:min(h, y0 + strip)] = photo_rows(w, h, seed, y0, min(h, y0 + strip))
    return img


def sha(b):
    return hashlib.sha256(b).hexdigest()


def encode(expr, img, mode):
    """oracle_lib.encode with an output buffer sized by the codec (the wrapper's 16 B/px would be 4 GB at 16384^2)"""
    h, w = img.shape[:2]
    cap = 64 + w * h * (3 if expr == "delta" else 2) + (1 << 24)
    out = np.empty(cap, np.uint8)
    ln = C.c_uint64(0)
    st = O.KmStats()
    rc = O.lib().orc_encode(expr.encode(), mode, C.c_uint64(O.DEFAULT_SEED), O._p(img), C.c_uint32(w), C.c_uint32(h),
                            O._p(out), C.c_uint64(cap), C.byref(ln), C.byref(st))
    assert rc == 0, (expr, rc)
    return out[:ln.value], st.as_dict()


def load():
    if os.path.exists(OUT):
        with open(OUT) as f:
            return json.load(f)
    return {"_about": "oracle results at BASELINE sizes; made by tests/golden/make_fullsize_digests.py (see its docstring)", "cases": {}}


def save(d):
    with open(OUT + ".tmp", "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
    os.replace(OUT + ".tmp", OUT)


def case_c2(d, threads):
    img = photo(4096, 4096, S + 2)
    t = time.time()
    data, st = encode("cluster-colors(256)", img, O.MODE_L)
    d["cases"]["c2"] = dict(codec="cluster-colors(256)", w=4096, h=4096, generator="P", seed_offset=2, mode="L",
                            sha256=sha(data.tobytes()), length=int(data.size), iterations=st["iterations"],
                            image_sha256=sha(img.tobytes()), oracle_seconds=round(time.time() - t, 1))


def case_c2r(d, threads):
    img = photo(4096, 4096, S + 2)
    t = time.time()
    data, st = encode("cluster-colors(256)", img, O.MODE_R)
    rc, back = O.decode("cluster-colors(256)", data.tobytes())
    assert rc == 0
    d["cases"]["c2r"] = dict(codec="cluster-colors(256)", w=4096, h=4096, generator="P", seed_offset=2, mode="R",
                             length=int(data.size), bytes_per_px=data.size / (4096 * 4096), mse=O.mse(img, back),
                             iterations=st["iterations"], oracle_seconds=round(time.time() - t, 1))


def case_c5(d, threads):
    img = photo(16384, 16384, S + 5)
    t = time.time()
    data, st = encode("delta", img, O.MODE_L)
    d["cases"]["c5"] = dict(codec="delta", w=16384, h=16384, generator="P", seed_offset=5, sha256=sha(data.tobytes()),
                            length=int(data.size), image_sha256=sha(img.tobytes()), oracle_seconds=round(time.time() - t, 1),
                            note="scan = the build's frozen generalised Hilbert curve (parity unpinned against zhang_hilbert, DESIGN 2)")


def case_c4(d, threads):
    F, w, h, K = 128, 1920, 1080, 256
    frames = np.empty((F * h, w, 3), np.uint8)
    for f in range(F):
        frames[f * h:(f + 1) * h] = photo(w, h, S + 4 + f)
    t = time.time()
    data, st = encode("cluster-colors(%d)" % K, frames, O.MODE_L)        # the union clustering: the stacked frames as one image
    rc, whole = O.decode("cluster-colors(%d)" % K, data.tobytes(), max_px=F * w * h)
    assert rc == 0
    per = []
    for f in range(F):                                                   # every frame: dims + its OWN tree + its payload
        b = O.Buf()
        O.lib().orc_buf_init(C.byref(b))
        red = np.ascontiguousarray(whole[f * h:(f + 1) * h])
        assert O.lib().orc_hufman_encode(O._p(red), C.c_uint32(w), C.c_uint32(h), C.byref(b)) == 0
        s = bytes(C.string_at(b.data, b.len))
        O.lib().orc_buf_free(C.byref(b))
        per.append(dict(sha256=sha(s), length=len(s)))
    d["cases"]["c4"] = dict(codec="cluster-colors(256)", frames=F, w=w, h=h, generator="P", seed_offset="4+f", mode="L",
                            stacked_sha256=sha(data.tobytes()), stacked_length=int(data.size), iterations=st["iterations"],
                            frame_streams=per, all_frames_sha256=sha("".join(p["sha256"] for p in per).encode()),
                            oracle_seconds=round(time.time() - t, 1))


def case_voronoi(size):
    def run(d, threads):
        img = photo(size, size, S + 3)
        t = time.time()
        data, st = encode("voronoi(2048)", img, O.MODE_L)
        d["cases"]["v%d" % size] = dict(codec="voronoi(2048)", w=size, h=size, generator="P", seed_offset=3, mode="L",
                                        sha256=sha(data.tobytes()), length=int(data.size), iterations=st["iterations"],
                                        image_sha256=sha(img.tobytes()), oracle_seconds=round(time.time() - t, 1))
    return run


CASES = {"c2": case_c2, "c2r": case_c2r, "c5": case_c5, "c4": case_c4,
         "v1024": case_voronoi(1024), "v2048": case_voronoi(2048), "v4096": case_voronoi(4096)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=",".join(CASES))
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    small = synth.photo(200, 130, S + 9)
    assert np.array_equal(photo(200, 130, S + 9, strip=37), small), "strip generator differs from synth.photo"
    O.lib().orc_set_lloyd_threads(a.threads)
    for name in a.only.split(","):
        t = time.time()
        d = load()
        CASES[name](d, a.threads)
        save(d)
        print("%s done in %.0f s: %s" % (name, time.time() - t, {k: v for k, v in d["cases"][name].items() if k != "frame_streams"}), flush=True)


if __name__ == "__main__":
    main()
