#!/bin/bash
# the persistent K-means loop on the headline image over the number of iterations whose movers are booked by aggregated rounds
# (CNIIC_KM_AGG_LAUNCHES, testing build)
R=$(cd "$(dirname "$0")/.." && pwd)
for a in 0 3 8 16 26 64; do
  echo "== agg iterations $a"
  CNIIC_KM_AGG_LAUNCHES=$a PS_BLOCKS_TRACE=0 python3 $R/tools/ps_trace.py 4096 256 $R/gpurun_out/ps_agg.csv 2>&1 | grep -E "loop|encode" 
done
