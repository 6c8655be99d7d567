set -x
mkdir -p gpurun_out/r03d
python -m pytest tests/test_decode_device.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r03d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03d/pytest.log
tail -6 gpurun_out/r03d/pytest.log
for c in "" "--config c5 --c5-size 4096" "--config c5"; do python bench.py --decode --cpu-sample 0 $c 2>gpurun_out/r03d/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['config']['pixels_per_gpu'], 'ms', d['ms_per_step'], 'stages', d['stages'], 'host_io', d['host_io_ms_per_step'], 'roof', d['roofline']['frac'])"; done
CNIIC_TRACE_HOST=1 python tools/decode_dev_probe.py 4096 > gpurun_out/r03d/probe4k.txt 2>&1
CNIIC_TRACE_HOST=1 python tools/decode_dev_probe.py 16384 delta > gpurun_out/r03d/probe16k.txt 2>&1
grep -v "^\[host\] \(huf\|delta:\|km\|map\|build\|tree\|so\.\|pack\)" gpurun_out/r03d/probe4k.txt | tail -40
tail -12 gpurun_out/r03d/probe16k.txt
