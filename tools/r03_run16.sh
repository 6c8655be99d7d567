mkdir -p gpurun_out/r03n
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/r03n/sq_counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/r03n/sq_counters.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03n/pmc1 -o p -- python3 $GRAFT_REPO_ROOT/tools/launch_trace.py /tmp/x.csv > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03n/pmc2 -o p -- python3 $GRAFT_REPO_ROOT/tools/launch_trace.py /tmp/x.csv > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_rows.py gpurun_out/r03n/pmc1 k_rgbw_assign 0,5,10,20,30,45,58 > gpurun_out/r03n/pmc1.txt 2>&1
python3 tools/pmc_rows.py gpurun_out/r03n/pmc2 k_rgbw_assign 0,5,10,20,30,45,58 > gpurun_out/r03n/pmc2.txt 2>&1
cat gpurun_out/r03n/pmc1.txt gpurun_out/r03n/pmc2.txt
rm -rf gpurun_out/r03n/pmc1 gpurun_out/r03n/pmc2
