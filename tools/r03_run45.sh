#!/bin/bash
set -e
R=$(pwd); O=$R/gpurun_out/r45; mkdir -p $O
CNIIC_TRACE_HOST=1 timeout -k 10 200 python tools/bench_others.py hufman 2>&1 | tail -22
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o s -- python3 $R/tools/bench_others.py hufman > /dev/null 2>&1
cp $(find $O/s -name '*kernel_stats.csv' | head -1) $O/hufman_kernel_stats.csv; rm -rf $O/s
cut -d, -f1-4 $O/hufman_kernel_stats.csv | cut -c1-150 | head -30
