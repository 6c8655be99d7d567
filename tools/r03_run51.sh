#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_decode_device.py -m gpu -x -q -k "delta or codec or golden or fuzz" > gpurun_out/r51_tests.log 2>&1 || { tail -30 gpurun_out/r51_tests.log; exit 1; }
tail -2 gpurun_out/r51_tests.log
timeout -k 10 200 python tests/fuzz_codecs.py 45 2>&1 | tail -1
CNIIC_KERNEL_TIMERS=1 timeout -k 10 200 python tools/bench_others.py delta delta16k 2>&1 | tail -2
for i in 1 2; do timeout -k 10 300 python bench.py --config c5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['ms_per_step'], d['value'], d.get('stages'))"; done
