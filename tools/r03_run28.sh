#!/bin/bash
# round 3, run 28: what the LDS does in an assign launch (per-launch counters)
set -e
R=$(pwd); O=$R/gpurun_out/r28; mkdir -p $O
export TMPDIR=/tmp
export CNIIC_KM_POOL=0
cd /tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_a -o p -- python3 $R/tools/launch_trace.py /tmp/x.csv > /dev/null 2>&1
python3 $R/tools/pmc_rows.py $O/pmc_a k_rgbw_assign > $O/pmc_lds_per_launch.txt
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_b -o p -- python3 $R/tools/launch_trace.py /tmp/x.csv > /dev/null 2>&1
python3 $R/tools/pmc_rows.py $O/pmc_b k_rgbw_assign > $O/pmc_lds2_per_launch.txt
rm -rf $O/pmc_a $O/pmc_b
head -14 $O/pmc_lds_per_launch.txt; head -14 $O/pmc_lds2_per_launch.txt
