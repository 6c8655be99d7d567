#!/usr/bin/env python3
# tests/fuzz_kmeans.py [seconds] -- the two lossy codecs against the oracle on random small images: `cluster-colors(K)` (dense
# table and pixel partition, the persistent launch and the launches) and `voronoi(K)` (pivot pruning / brute force, static and
# dynamic dealing of super-tiles): same return code, same bytes, same iteration count; the stream decodes to what the oracle
# decodes it to.  Test infrastructure: the oracle is the checker.  FUZZ_SEED picks the sequence.
import os, sys, time
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import oracle_lib as O

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
CC_KNOBS = [{}, {"CNIIC_SP_MIN_PIXELS": "0"}, {"CNIIC_SP_MIN_PIXELS": str(1 << 40)},
            {"CNIIC_KM_LOOP": "1"}, {"CNIIC_KM_UNFUSED": "1", "CNIIC_KM_LOOP": "1"}, {"CNIIC_KM_BATCH": "1"}, {"CNIIC_KM_MAXSKIP": "4"},
            # round 5, the persistent launch: odd grids, most points in memory, a launch that gives up under the encode that did not wait for it,
            # the full schedule without its clean-cell skip
            {"CNIIC_KM_PS_BLOCKS": "3", "CNIIC_SP_MIN_PIXELS": "0"}, {"CNIIC_KM_PS_BLOCKS": "37"}, {"CNIIC_KM_PS_BLOCKS": "8", "CNIIC_TEST_PS_LDS_BYTES": "60000"},
            {"CNIIC_TEST_PS_ABORT_AT": "2"}, {"CNIIC_TEST_PS_ABORT_AT": "1", "CNIIC_SP_MIN_PIXELS": "0"}, {"CNIIC_KM_PS_CLEANSKIP": "0"}, {"CNIIC_KM_MAXSKIP": "4", "CNIIC_KM_PS_BLOCKS": "5"}]
VOR_KNOBS = [{}, {"CNIIC_XY_DYN": "0"}, {"CNIIC_XY_DYN": "1"}, {"CNIIC_XY_DYN": "100000"}]


def image():
    big = rng.random() < 0.08   # now and then an image with several super-tiles / more than one block's worth of cells
    h, w = (int(rng.integers(200, 420)), int(rng.integers(200, 420))) if big else (int(rng.integers(1, 160)), int(rng.integers(1, 160)))
    y, x = np.mgrid[0:h, 0:w]
    style = int(rng.integers(0, 6))
    if style == 0:   img = rng.integers(0, 256, (h, w, 3))
    elif style == 1: img = rng.integers(0, int(rng.integers(2, 40)), (h, w, 3)) * int(rng.integers(1, 7))
    elif style == 2: img = (np.add.outer(np.arange(h), np.arange(w))[..., None] * np.array([1, 2, 3]) // int(rng.integers(1, 9))) % 256
    elif style == 3:
        img = np.full((h, w, 3), rng.integers(0, 256, 3)); img[: h // 2] = rng.integers(0, 256, 3)
    elif style == 4: img = np.clip(rng.normal(128, int(rng.integers(1, 60)), (h, w, 3)), 0, 255)
    else:            img = np.stack([x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1), (x ^ y) & 255], 2) + rng.integers(-2, 3, (h, w, 3))
    return np.ascontiguousarray(np.clip(img, 0, 255), np.uint8)


def run(ctx, budget):
    """`budget` seconds of random cases; returns how many were checked (an assertion stops at the first difference)"""
    from cniic_amd import _lib
    t0, cases, said = time.time(), 0, time.time()
    while time.time() - t0 < budget:
        if time.time() - said > 60:   # (a long run says that it is alive)
            said = time.time()
            sys.stderr.write("fuzz: %d cases after %.0f s\n" % (cases, said - t0)); sys.stderr.flush()
        img = image()
        vor = rng.random() < 0.4
        K = int(rng.choice([1, 2, 3, 7, 16, 40, 64] if vor else [1, 2, 3, 7, 16, 64, 255, 256, 257, 600]))
        expr = ("voronoi(%d)" if vor else "cluster-colors(%d)") % K
        knobs = VOR_KNOBS if vor else CC_KNOBS
        knob = knobs[int(rng.integers(0, len(knobs)))]
        saved = {k: os.environ.get(k) for k in knob}
        os.environ.update(knob)
        try:
            erc, edata, est = O.encode(expr, img)
            rc, data, st = ctx.encode(expr, img, allow=(_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE))
            what = (expr, img.shape, knob)
            assert rc == erc, what + ("return code", rc, erc)
            if rc == 0:
                assert data == edata, what + ("bytes",)
                assert st["iterations"] == est["iterations"], what + ("iterations", st["iterations"], est["iterations"])
                rc2, back = ctx.decode(expr, data)
                erc2, eback = O.decode(expr, data)
                assert rc2 == erc2 == 0 and np.array_equal(back, eback), what + ("decode",)
            cases += 1
        except Exception:
            np.save("/tmp/fuzz_kmeans_fail.npy", img)
            raise
        finally:
            for k, v in saved.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
    return cases


if __name__ == "__main__":
    import cniic_amd
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    n = run(cniic_amd.Context(0), seconds)
    print("fuzz_kmeans: %d cases in %.0f s, all equal to the oracle" % (n, seconds))
