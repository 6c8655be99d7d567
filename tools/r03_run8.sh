mkdir -p gpurun_out/r03g
timeout -k 10 600 python -m pytest tests/test_decode_device.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03g/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03g/pytest.log
tail -4 gpurun_out/r03g/pytest.log
grep -q "pytest rc 0" gpurun_out/r03g/pytest.log || exit 1
CNIIC_HD_STATS=1 timeout -k 10 300 python tools/decode_dev_probe.py 4096 "cluster-colors(256)" delta 2>&1 | grep "^\[hd\]\|codec" | tail -12
CNIIC_HD_STATS=1 timeout -k 10 300 python tools/decode_dev_probe.py 16384 delta 2>&1 | grep "^\[hd\]\|codec" | tail -5
for c in "" "--config c5 --c5-size 4096" "--config c5"; do timeout -k 10 300 python bench.py --decode --cpu-sample 0 $c 2>gpurun_out/r03g/err.txt | python tools/show_dec.py; done
timeout -k 10 300 python tools/decode_dev_probe.py 4096 hufman 2>&1 | grep codec
