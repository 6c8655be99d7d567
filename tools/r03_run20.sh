#!/bin/bash
# round 3, run 20: block-wide candidate build in the full schedule
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_rare_branches.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r20_tests.log 2>&1 || { tail -30 gpurun_out/r20_tests.log; exit 1; }
tail -2 gpurun_out/r20_tests.log
timeout -k 10 300 python bench.py --cpu-sample 0 --no-extras > gpurun_out/r20_bench.json 2>gpurun_out/r20_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r20_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], d['value'], r['frac'], {k:(v['us'],v.get('n')) for k,v in r['by_class'].items()})
PY
