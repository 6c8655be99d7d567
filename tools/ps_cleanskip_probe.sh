#!/bin/bash
# the persistent K-means loop on the headline image with and without the full schedule's clean-cell skip (CNIIC_KM_PS_CLEANSKIP, testing build), three runs each, alternating
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2 3; do for v in 1 0; do
  echo "== clean skip $v"
  CNIIC_KM_PS_CLEANSKIP=$v PS_BLOCKS_TRACE=0 python3 $R/tools/ps_trace.py 4096 256 $R/gpurun_out/ps_cs.csv 2>&1 | grep -E "loop|encode"
done; done
