#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r48_tests.log 2>&1 || { tail -30 gpurun_out/r48_tests.log; exit 1; }
tail -2 gpurun_out/r48_tests.log
timeout -k 10 600 python bench.py > gpurun_out/r48_bench.json 2> gpurun_out/r48_bench.err || { tail -20 gpurun_out/r48_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r48_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('ms_per_step','value','Mpixels_per_s','error')}) for k,v in d.items() if k in ('c4_one_gpu','batch_own_palettes','host_io_ms_per_step','extras_error')})
PY
