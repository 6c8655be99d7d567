#!/bin/bash
# round 3, run 21: SQ counters per assign launch with the block-wide candidate build
set -e
R=$(pwd); O=$R/gpurun_out/r21; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python3 $R/tools/launch_trace.py /tmp/x.csv > /dev/null 2>&1
python3 $R/tools/pmc_rows.py $O/pmc_sq k_rgbw_assign > $O/pmc_sq_assign_per_launch.txt
rm -rf $O/pmc_sq
head -16 $O/pmc_sq_assign_per_launch.txt
