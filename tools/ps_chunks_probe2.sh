#!/bin/bash
# chunk counts on OTHER images than the headline's (seeds, sizes): is 24 better than 16 in general?
R=$(cd "$(dirname "$0")/.." && pwd)
for cfg in "4096 3" "4096 7" "4096 11" "2048 2" "2048 5" "8192 2"; do
  set -- $cfg
  for lib in libcniic_hip_testing.so libcniic_hip_ch24.so libcniic_hip_ch24s40.so libcniic_hip_ch32s48.so; do
    echo -n "size $1 seed $2 $lib: "
    PS_SEED=$2 CNIIC_LIB_FILE=$lib PS_BLOCKS_TRACE=0 python3 $R/tools/ps_trace.py $1 256 $R/gpurun_out/ps_ch.csv 2>&1 | grep -E "loop"
  done
done
