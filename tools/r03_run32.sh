#!/bin/bash
# round 3, run 32: voronoi per-launch durations, static vs dynamic dealing of the super-tiles
set -e
R=$(pwd); O=$R/gpurun_out/r32; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for d in 0 1; do
CNIIC_XY_DYN=$d rocprofv3 --kernel-trace --output-format csv -d $O/t$d -o t -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
python3 $R/tools/trace_iters.py $(find $O/t$d -name '*kernel_trace.csv' | head -1) k_xy_assign > $O/iters_dyn$d.txt
rm -rf $O/t$d
done
paste $O/iters_dyn0.txt $O/iters_dyn1.txt | cut -c1-250 | tail -22
