mkdir -p gpurun_out/r03h
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03h/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03h/pytest.log
tail -6 gpurun_out/r03h/pytest.log
grep -q "pytest rc 0" gpurun_out/r03h/pytest.log || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r03h/bench.json 2> gpurun_out/r03h/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03h/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'host_io', d['host_io_ms_per_step'])
print('c4', d['c4_one_gpu']['value'], d['c4_one_gpu']['ms_per_step'])
print('batch', d['batch_own_palettes'])
print('cpu all cores', d['c4_one_gpu'].get('cpu_baseline'))
PY
