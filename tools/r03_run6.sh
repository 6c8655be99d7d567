mkdir -p gpurun_out/r03f
CNIIC_HD_STATS=1 timeout -k 10 300 python tools/decode_dev_probe.py 4096 "cluster-colors(256)" delta 2>&1 | grep "^\[hd\]\|codec" | tail -16
CNIIC_HD_STATS=1 timeout -k 10 300 python tools/decode_dev_probe.py 16384 delta 2>&1 | grep "^\[hd\]\|codec" | tail -5
CNIIC_TRACE_HOST=1 timeout -k 10 300 python tools/decode_probe.py 4096 hufman > gpurun_out/r03f/probe_huf_host.txt 2>&1
grep -v "^\[host\] \(huf\|delta:\|km\|map\|build\|tree\|so\.\|pack\)" gpurun_out/r03f/probe_huf_host.txt | tail -8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03f/prof -o dec -- python3 $GRAFT_REPO_ROOT/tools/decode_dev_probe.py 4096 hufman > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_stats.py gpurun_out/r03f/prof > gpurun_out/r03f/kstats.txt; grep "k_tp\|k_hd\|copyBuffer\|fillBuffer" gpurun_out/r03f/kstats.txt
