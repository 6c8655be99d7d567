#!/usr/bin/env python3
"""Generates tests/golden/reseed_golden.npz -- inputs that force the rare branches of kmeans::cluster:

  * update_centroids' empty-cluster branch (src/kmeans.rs:117-134; deviation D2: the stolen point is
    reseed_index(seed, iteration, cluster) instead of thread_rng), for ColorCount and ColorPos points;
  * check_enough_active_clusters (src/kmeans.rs:41-57) failing.

The cases were found by a random search over small inputs with the CPU oracle (mode L); the search is
re-run here for the fixed generator seeds below, so the script documents exactly how each input is made.
A run that ENDS with two empty clusters did not turn up in 3*10^5 random inputs (an empty cluster is
re-seeded onto a point, which then joins it unless that point already sits on its own centroid), so the
FEW_ACTIVE cases stop the loop early with max_iters -- same check, same members array.

    python tests/golden/make_reseed_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402

RGBW_SEEDS = [38, 621, 1268, 2355]          # 3, 3, 3, 2 empty-cluster reseeds over the run
RGBW_FEW = (1473, 2)                         # (generator seed, max_iters): two clusters empty after iteration 1
XY_SEEDS = [10, 31, 43, 4]                   # 4, 7, 2, 1 reseeds
XY_FEW = (31, 2)


def rgbw_case(seed):
    rng = np.random.default_rng(seed)
    U = int(rng.integers(12, 120)); K = int(rng.integers(3, min(U, 40)))
    span = int(rng.choice([4, 8, 16, 64, 256]))
    keys = np.unique(rng.integers(0, span, (U, 3)) @ np.array([65536, 256, 1])).astype(np.uint32)
    w = rng.integers(1, 50, keys.size).astype(np.uint32)
    return keys, w, K


def xy_case(seed):
    rng = np.random.default_rng(seed)
    h = int(rng.integers(3, 12)); w = int(rng.integers(3, 12))
    K = int(rng.integers(3, max(4, h * w // 2)))
    lev = int(rng.choice([2, 4, 256]))
    img = (rng.integers(0, lev, (h, w, 3)) * (255 // (lev - 1))).astype(np.uint8)
    return img, K


def pts_of_keys(keys):
    return np.stack([(keys >> 16) & 255, (keys >> 8) & 255, keys & 255], axis=1).astype(np.int32)


def xy_pts(img):
    h, w = img.shape[:2]
    y, x = np.mgrid[0:h, 0:w]
    return np.concatenate([x.reshape(-1, 1), y.reshape(-1, 1), img.reshape(-1, 3)], axis=1).astype(np.int32)


def put(out, name, r, extra):
    out[name + "_centroids"] = r["centroids"]
    out[name + "_labels"] = r["labels"]
    out[name + "_members"] = r["members"]
    out[name + "_stats"] = np.array([r["stats"]["iterations"], r["stats"]["empty_reseeds"], r["stats"]["moved_last"]], np.uint64)
    out.update({name + "_" + k: v for k, v in extra.items()})


def main():
    out = {}
    for s in RGBW_SEEDS:
        keys, w, K = rgbw_case(s)
        rc, r = O.kmeans(O.PT_RGBW, O.MODE_L, pts_of_keys(keys), w, K)
        assert rc == 0 and r["stats"]["empty_reseeds"] >= 2, (s, rc, r["stats"])
        put(out, "rgbw%d" % s, r, dict(keys=keys, weight=w, K=np.array([K], np.uint32)))
    keys, w, K = rgbw_case(RGBW_FEW[0])
    rc, r = O.kmeans(O.PT_RGBW, O.MODE_L, pts_of_keys(keys), w, K, max_iters=RGBW_FEW[1])
    assert rc == O.FEW_ACTIVE
    put(out, "rgbw_few", r, dict(keys=keys, weight=w, K=np.array([K], np.uint32), max_iters=np.array([RGBW_FEW[1]], np.uint64)))
    for s in XY_SEEDS:
        img, K = xy_case(s)
        rc, r = O.kmeans(O.PT_XYRGB, O.MODE_L, xy_pts(img), None, K)
        assert rc == 0 and r["stats"]["empty_reseeds"] >= 1, (s, rc, r["stats"])
        put(out, "xy%d" % s, r, dict(img=img, K=np.array([K], np.uint32)))
    img, K = xy_case(XY_FEW[0])
    rc, r = O.kmeans(O.PT_XYRGB, O.MODE_L, xy_pts(img), None, K, max_iters=XY_FEW[1])
    assert rc == O.FEW_ACTIVE
    put(out, "xy_few", r, dict(img=img, K=np.array([K], np.uint32), max_iters=np.array([XY_FEW[1]], np.uint64)))
    np.savez_compressed(os.path.join(HERE, "reseed_golden.npz"), **out)
    print("wrote reseed_golden.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
