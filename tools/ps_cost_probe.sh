#!/bin/bash
# the persistent K-means loop over the cost model its cell chunks are balanced by (CNIIC_CELL_COST per cell + CNIIC_CELL_SWEEP_COST per sweep of 256 points; testing build)
R=$(cd "$(dirname "$0")/.." && pwd)
for cs in "512 256" "1024 256" "2048 256" "256 256" "512 128" "512 512" "1024 64" "4096 256"; do
  set -- $cs
  echo "== cell cost $1, sweep cost $2"
  CNIIC_CELL_COST=$1 CNIIC_CELL_SWEEP_COST=$2 PS_BLOCKS_TRACE=0 python3 $R/tools/ps_trace.py 4096 256 $R/gpurun_out/ps_cost.csv 2>&1 | grep -E "loop"
done
