mkdir -p gpurun_out/r03m
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_rare_branches.py tests/test_golden.py tests/test_dist.py tests/test_mode_r_band.py -m gpu -x -q > gpurun_out/r03m/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03m/pytest.log
tail -3 gpurun_out/r03m/pytest.log
grep -q "pytest rc 0" gpurun_out/r03m/pytest.log || exit 1
timeout -k 10 600 python bench.py --cpu-sample 0 --no-extras > gpurun_out/r03m/bench.json 2> gpurun_out/r03m/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03m/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], d['roofline']['by_class'])
PY
timeout -k 10 300 python tools/fuzz_sp.py 500 99 2>&1 | tail -2
