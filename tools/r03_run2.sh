set -x
mkdir -p gpurun_out/r03b
python -m pytest tests/test_decode_device.py tests/test_gpu_parity.py tests/test_huffman_optimal.py -m gpu -x -q > gpurun_out/r03b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03b/pytest.log
tail -15 gpurun_out/r03b/pytest.log
python bench.py --decode --cpu-sample 0 > gpurun_out/r03b/dec_c2.json 2> gpurun_out/r03b/dec_c2.err; echo "rc $?"; tail -3 gpurun_out/r03b/dec_c2.err
python bench.py --decode --config c5 --c5-size 4096 --cpu-sample 0 > gpurun_out/r03b/dec_d4k.json 2> gpurun_out/r03b/dec_d4k.err; echo "rc $?"; tail -3 gpurun_out/r03b/dec_d4k.err
python bench.py --decode --config c5 --cpu-sample 0 > gpurun_out/r03b/dec_c5.json 2> gpurun_out/r03b/dec_c5.err; echo "rc $?"; tail -3 gpurun_out/r03b/dec_c5.err
CNIIC_TRACE_HOST=1 python tools/decode_probe.py 4096 hufman > gpurun_out/r03b/decode_probe.txt 2>&1
cat gpurun_out/r03b/dec_c2.json gpurun_out/r03b/dec_d4k.json gpurun_out/r03b/dec_c5.json
tail -12 gpurun_out/r03b/decode_probe.txt
