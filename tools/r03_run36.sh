#!/bin/bash
set -e
R=$(pwd); O=$R/gpurun_out/r36; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
python3 $R/tools/trace_iters.py $(find $O/t -name '*kernel_trace.csv' | head -1) k_xy_assign > $O/iters_ordered.txt
rm -rf $O/t
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p -o p -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
python3 $R/tools/pmc_kernel_mean.py $O/p k_xy_assign > $O/pmc_ordered.txt 2>&1
rm -rf $O/p
tail -20 $O/iters_ordered.txt; cat $O/pmc_ordered.txt
