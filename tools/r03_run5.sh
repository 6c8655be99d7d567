set -x
mkdir -p gpurun_out/r03e
timeout -k 10 600 python -m pytest tests/test_decode_device.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03e/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03e/pytest.log
tail -8 gpurun_out/r03e/pytest.log
grep -q "pytest rc 0" gpurun_out/r03e/pytest.log || exit 1
for c in "" "--config c5 --c5-size 4096" "--config c5"; do timeout -k 10 300 python bench.py --decode --cpu-sample 0 $c 2>gpurun_out/r03e/err.txt | python tools/show_dec.py; done
CNIIC_TRACE_HOST=1 timeout -k 10 300 python tools/decode_dev_probe.py 4096 hufman > gpurun_out/r03e/probe_huf.txt 2>&1
grep -v "^\[host\] \(huf\|delta:\|km\|map\|build\|tree\|so\.\|pack\)" gpurun_out/r03e/probe_huf.txt | tail -14
CNIIC_TRACE_HOST=1 timeout -k 10 300 python tools/decode_probe.py 4096 hufman > gpurun_out/r03e/probe_huf_host.txt 2>&1
grep -v "^\[host\] \(huf\|delta:\|km\|map\|build\|tree\|so\.\|pack\)" gpurun_out/r03e/probe_huf_host.txt | tail -8
