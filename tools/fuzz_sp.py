"""Differential fuzz: cluster-colors through the pixel partition vs through the dense table, random images / sizes / K.
usage: fuzz_sp.py [cases] [seed]   (tools only)"""
import os, sys, json
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")   # the probes' knobs exist in the testing build of the library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, cniic_amd
from cniic_amd import _lib
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = cniic_amd.Context(0)
bad = 0
for i in range(cases):
    h, w = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    style = int(rng.integers(0, 5))
    if style == 0:   img = rng.integers(0, 256, (h, w, 3))
    elif style == 1: img = rng.integers(0, int(rng.integers(2, 40)), (h, w, 3)) * int(rng.integers(1, 7))
    elif style == 2: img = (np.add.outer(np.arange(h), np.arange(w))[..., None] * np.array([1, 2, 3]) // int(rng.integers(1, 9))) % 256
    elif style == 3: img = np.full((h, w, 3), rng.integers(0, 256, 3)); img[: h // 2] = rng.integers(0, 256, 3)
    else:            img = np.clip(rng.normal(128, int(rng.integers(1, 60)), (h, w, 3)), 0, 255)
    img = np.ascontiguousarray(img, np.uint8)
    K = int(rng.choice([1, 2, 3, 7, 16, 64, 255, 256, 257, 600]))
    expr = "cluster-colors(%d)" % K
    os.environ["CNIIC_SP_MIN_PIXELS"] = "0"
    rc1, d1, s1 = ctx.encode(expr, img, allow=(_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE))
    os.environ["CNIIC_SP_MIN_PIXELS"] = str(1 << 40)
    rc2, d2, s2 = ctx.encode(expr, img, allow=(_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE))
    ok = rc1 == rc2 and (rc1 != 0 or (d1 == d2 and s1["iterations"] == s2["iterations"]))
    if not ok:
        bad += 1
        print(json.dumps(dict(case=i, h=h, w=w, style=style, K=K, rc1=rc1, rc2=rc2)))
print(json.dumps(dict(cases=cases, mismatches=bad)))
sys.exit(1 if bad else 0)
