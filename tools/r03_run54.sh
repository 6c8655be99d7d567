#!/bin/bash
set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rgbw or cluster or ccol or codec" 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 300 python bench.py --cpu-sample 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['frac'], {k:v['us'] for k,v in r['by_class'].items()})"
done
