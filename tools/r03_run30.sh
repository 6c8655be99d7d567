#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r30_tests.log 2>&1 || { tail -30 gpurun_out/r30_tests.log; exit 1; }
tail -2 gpurun_out/r30_tests.log
timeout -k 10 200 python tests/fuzz_codecs.py 60 2>&1 | tail -2
timeout -k 10 300 python bench.py --config c5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['ms_per_step'], d['value'])"
timeout -k 10 200 python tools/bench_others.py hufman delta 2>&1 | tail -2
timeout -k 10 300 python bench.py --cpu-sample 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('headline', d['ms_per_step'], r['frac'], {k:v['us'] for k,v in r['by_class'].items()})"
