#!/bin/bash
# voronoi(2048) at 4096^2 over the number of moved centroids below which a block takes its super-tiles statically instead of drawing them
# (CNIIC_XY_DYN, testing build)
R=$(cd "$(dirname "$0")/.." && pwd)
for d in 16 8 4 2 1; do
  CNIIC_USE_TESTING_LIB=1 CNIIC_XY_DYN=$d python3 $R/bench.py --config c3 --cpu-sample 0 --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('xy dyn $d:', d['ms_per_step'], 'ms', d['config']['kmeans_iterations'], 'iterations', d.get('parity'))"
done
