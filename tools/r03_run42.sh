#!/bin/bash
# round 3, run 42: voronoi with per-pixel margins
set -e
mkdir -p gpurun_out
CNIIC_XY_MARGINS=4 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_rare_branches.py -m gpu -x -q -k "xy or voronoi or pos" > gpurun_out/r42_tests.log 2>&1 || { tail -30 gpurun_out/r42_tests.log; exit 1; }
tail -2 gpurun_out/r42_tests.log
CNIIC_XY_MARGINS=4 timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "voronoi or config3" 2>&1 | tail -2
for cfg in "CNIIC_XY_MARGINS=0" "CNIIC_XY_MARGINS=4"; do
env $cfg timeout -k 10 300 python bench.py --config c3 --steps 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['config'].get('iterations'), d.get('bytes_per_px'))"
done
