#!/bin/bash
# A/B of one testing-build knob of the persistent K-means launch on the headline image:  bash tools/ps_ab.sh CNIIC_KM_PS_PIVOT_REUSE 1 0
R=$(cd "$(dirname "$0")/.." && pwd)
knob=$1; shift
for rep in 1 2 3; do for v in "$@"; do
  echo -n "$knob=$v: "
  env $knob=$v PS_BLOCKS_TRACE=0 python3 $R/tools/ps_trace.py 4096 256 $R/gpurun_out/ps_ab.csv 2>&1 | grep -E "loop"
done; done
