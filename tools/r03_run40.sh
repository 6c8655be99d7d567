#!/bin/bash
set -e
R=$(pwd); O=$R/gpurun_out/r40; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
python3 $R/tools/trace_iters.py $(find $O/t -name '*kernel_trace.csv' | head -1) k_xy_assign > $O/iters_group.txt
rm -rf $O/t
sed -n 19,26p $O/iters_group.txt
