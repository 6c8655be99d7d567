mkdir -p gpurun_out/r03k
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_rare_branches.py tests/test_golden.py tests/test_boundary.py tests/test_decode_device.py -m gpu -x -q > gpurun_out/r03k/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03k/pytest.log
tail -4 gpurun_out/r03k/pytest.log
grep -q "pytest rc 0" gpurun_out/r03k/pytest.log || exit 1
timeout -k 10 300 python tools/fuzz_vor.py 600 5 2>&1 | tail -3
timeout -k 10 600 python bench.py --config c3 --steps 3 --cpu-sample 0 2>gpurun_out/r03k/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['kmeans_iterations'])"
