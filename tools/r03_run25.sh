#!/bin/bash
set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for cfg in "CNIIC_KM_REBALANCE=0" "CNIIC_KM_REBALANCE=3 CNIIC_KM_REBALANCE_DRY=1" "CNIIC_KM_REBALANCE=3" "CNIIC_KM_REBALANCE=6" "CNIIC_KM_REBALANCE=100"; do
env $cfg timeout -k 10 300 python bench.py --cpu-sample 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$cfg', d['ms_per_step'], r['frac'], {k:v['us'] for k,v in r['by_class'].items()})"
done
