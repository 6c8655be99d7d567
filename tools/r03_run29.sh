#!/bin/bash
set -e
timeout -k 10 300 python bench.py --config c5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('runs', d['ms_per_step'], d['value'])"
CNIIC_HUF_HOST_MERGE=1 timeout -k 10 300 python bench.py --config c5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host merge', d['ms_per_step'], d['value'])"
CNIIC_TRACE_HOST=1 timeout -k 10 300 python tools/bench_others.py delta16k 2>&1 | tail -40
