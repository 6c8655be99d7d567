#!/bin/bash
# the persistent K-means loop over the number of moved centroids up to which an iteration takes the skip schedule (CNIIC_KM_MAXSKIP <= 64, testing build)
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do for m in 64 48 32 16; do
  echo -n "max skip $m: "
  CNIIC_KM_MAXSKIP=$m PS_BLOCKS_TRACE=0 python3 $R/tools/ps_trace.py 4096 256 $R/gpurun_out/ps_ms.csv 2>&1 | grep -E "loop"
done; done
