"""cluster-colors with K > 256 keeps a candidate strip of 256 entries and a super-cell list of 512 per wave (km_ccap / km_scap, k_kmeans_rgbw.hip,
round 4: whole-table strips left a block two waves at K = 2048).  A cell with more candidates than the strip holds sweeps against the whole table
instead.  Here that happens: 4096 distinct colours packed into a 16^3 cube and K = 1500 / 2048 centroids among them -- several hundred candidates
per 8^3 cell -- against the oracle, stream for stream."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [1500, 2048])
@pytest.mark.parametrize("reps", [1, 3])
def test_more_candidates_than_the_strip_holds(K, reps):
    import cniic_amd
    rng = np.random.default_rng(K + reps)
    r, g, b = np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij")
    cols = np.stack([r.ravel() + 100, g.ravel() + 60, b.ravel() + 30], axis=1).astype(np.uint8)   # 4096 colours, two 8^3 cells a side
    px = np.repeat(cols, reps, axis=0)
    px = px[rng.permutation(len(px))]
    w = 128
    h = len(px) // w
    img = np.ascontiguousarray(px[:w * h].reshape(h, w, 3))
    expr = "cluster-colors(%d)" % K
    rco, want, ost = O.encode(expr, img)
    with cniic_amd.Context(0) as ctx:
        rc, data, st = ctx.encode(expr, img)
    assert rc == rco
    if rc == 0:
        assert st["iterations"] == ost["iterations"]
        assert data == want
