#!/bin/bash
# the persistent K-means loop over the number of interleaved chunks a block owns (measuring builds libcniic_hip_ch<n>.so: tools/build_variant.sh ch<n> "-DCNIIC_PS_CHUNKS=<n>" k_kmeans_persist.hip)
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do for lib in ${PS_LIBS:-libcniic_hip_testing.so libcniic_hip_ch24.so libcniic_hip_ch32.so libcniic_hip_ch48.so}; do
  echo "== $lib"
  CNIIC_LIB_FILE=$lib PS_BLOCKS_TRACE=0 python3 $R/tools/ps_trace.py 4096 256 $R/gpurun_out/ps_ch.csv 2>&1 | grep -E "loop|encode"
done; done
