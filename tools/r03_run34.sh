#!/bin/bash
set -e
R=$(pwd); O=$R/gpurun_out/r34; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for d in 0 16; do
CNIIC_XY_DYN=$d rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p$d -o p -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
python3 $R/tools/pmc_kernel_mean.py $O/p$d k_xy_assign > $O/pmc_dyn$d.txt 2>&1
rm -rf $O/p$d
cat $O/pmc_dyn$d.txt
done
