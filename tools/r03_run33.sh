#!/bin/bash
set -e
for cfg in "CNIIC_XY_DYN=0" "CNIIC_XY_DYN=1" "CNIIC_XY_DYN=16" "CNIIC_XY_DYN=48" "CNIIC_XY_DYN=96" "CNIIC_XY_DYN=160" "CNIIC_XY_DYN=256"; do
env $cfg timeout -k 10 300 python bench.py --config c3 --steps 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'])"
done
