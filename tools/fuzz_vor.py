"""Differential fuzz: voronoi encode with pivot pruning + tile skipping vs the brute-force flag, and its decode vs the
brute-force repaint route (coordinates forced large via a no-op: the oracle is the checker in tests; here GPU vs GPU).
usage: fuzz_vor.py [cases] [seed]   (tools only)"""
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
    h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    style = int(rng.integers(0, 4))
    if style == 0:   img = rng.integers(0, 256, (h, w, 3))
    elif style == 1: img = rng.integers(0, int(rng.integers(2, 30)), (h, w, 3)) * 8
    elif style == 2: img = (np.add.outer(np.arange(h), np.arange(w))[..., None] * np.array([1, 2, 3])) % 256
    else:            img = np.full((h, w, 3), 77)
    img = np.ascontiguousarray(img, np.uint8)
    K = int(rng.choice([1, 2, 5, 16, 40, 100, 300]))
    expr = "voronoi(%d)" % K
    allow = (_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE)
    rc1, d1, s1 = ctx.encode(expr, img, allow=allow)
    rc2, d2, s2 = ctx.encode(expr, img, flags=_lib.KM_BRUTE_FORCE, allow=allow)
    rc3, d3, s3 = ctx.encode(expr, img, flags=_lib.KM_NO_SKIP, allow=allow)
    ok = rc1 == rc2 == rc3 and (rc1 != 0 or (d1 == d2 == d3 and s1["iterations"] == s2["iterations"] == s3["iterations"]))
    if not ok:
        bad += 1
        print(json.dumps(dict(case=i, h=h, w=w, style=style, K=K, rc=(rc1, rc2, rc3))))
print(json.dumps(dict(cases=cases, mismatches=bad)))
sys.exit(1 if bad else 0)
