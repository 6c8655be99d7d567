#!/bin/bash
# round 3, run 23: the cost model of the full schedule's ranges, now that a cell's build is cheap
for cc in 512 256 128 64 0; do for sc in 256 512; do
CNIIC_CELL_COST=$cc CNIIC_CELL_SWEEP_COST=$sc timeout -k 10 300 python bench.py --cpu-sample 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cell $cc sweep $sc', d['ms_per_step'], r['frac'], {k:v['us'] for k,v in r['by_class'].items()})"
done; done
