mkdir -p gpurun_out/r03l
timeout -k 10 900 python -m pytest tests/test_decode_device.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_injected_scan.py -m gpu -x -q > gpurun_out/r03l/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03l/pytest.log
tail -3 gpurun_out/r03l/pytest.log
grep -q "pytest rc 0" gpurun_out/r03l/pytest.log || exit 1
for c in "" "--config c5 --c5-size 4096" "--config c5"; do timeout -k 10 300 python bench.py --decode --cpu-sample 0 $c 2>gpurun_out/r03l/err.txt | python tools/show_dec.py; done
timeout -k 10 300 python tools/decode_dev_probe.py 4096 hufman 2>&1 | grep codec
