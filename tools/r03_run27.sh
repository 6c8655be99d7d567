#!/bin/bash
# round 3, run 27: full schedule with a pool of cells drawn by whoever runs out
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_rare_branches.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r27_tests.log 2>&1 || { tail -30 gpurun_out/r27_tests.log; exit 1; }
tail -2 gpurun_out/r27_tests.log
for cfg in "CNIIC_KM_POOL=0" "CNIIC_KM_POOL=100" "CNIIC_KM_POOL=250" "CNIIC_KM_POOL=400" "CNIIC_KM_POOL=600" "CNIIC_KM_POOL=250 CNIIC_KM_NO_BLOCK_BUILD=1"; do
env $cfg timeout -k 10 300 python bench.py --cpu-sample 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$cfg', d['ms_per_step'], r['frac'], {k:v['us'] for k,v in r['by_class'].items()})"
done
