#!/bin/bash
set -e
for i in 1 2 3; do
timeout -k 10 300 python bench.py --config c3 --steps 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['ms_per_step'])"
done
