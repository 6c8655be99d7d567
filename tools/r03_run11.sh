mkdir -p gpurun_out/r03i
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03i/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03i/pytest.log
tail -4 gpurun_out/r03i/pytest.log
grep -q "pytest rc 0" gpurun_out/r03i/pytest.log || exit 1
timeout -k 10 600 python bench.py --cpu-sample 0 > gpurun_out/r03i/bench.json 2> gpurun_out/r03i/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03i/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], d['roofline']['by_class'])
print('c4', d['c4_one_gpu']['value'], d['c4_one_gpu']['ms_per_step'], d['c4_one_gpu']['roofline']['frac'])
print('batch', d['batch_own_palettes']['value'], d['batch_own_palettes']['by_streams'])
PY
