#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_rare_branches.py -m gpu -x -q -k "xy or voronoi or pos" > gpurun_out/r41_tests.log 2>&1 || { tail -30 gpurun_out/r41_tests.log; exit 1; }
tail -2 gpurun_out/r41_tests.log
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "voronoi or config3" 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 300 python bench.py --config c3 --steps 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['ms_per_step'])"
done
