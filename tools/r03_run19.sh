#!/bin/bash
# round 3, run 19: hufman / delta encode with the tree built from runs of equal count
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_huffman_optimal.py tests/test_decode_device.py tests/test_boundary.py -m gpu -x -q > gpurun_out/r19_tests.log 2>&1 || { tail -30 gpurun_out/r19_tests.log; exit 1; }
tail -2 gpurun_out/r19_tests.log
timeout -k 10 200 python tests/fuzz_codecs.py 90 > gpurun_out/r19_fuzz.log 2>&1 || { tail -30 gpurun_out/r19_fuzz.log; exit 1; }
tail -3 gpurun_out/r19_fuzz.log
timeout -k 10 200 python tools/bench_others.py hufman delta delta16k > gpurun_out/r19_others.log 2>&1
cat gpurun_out/r19_others.log
CNIIC_HUF_HOST_MERGE=1 timeout -k 10 200 python tools/bench_others.py hufman delta > gpurun_out/r19_others_hostmerge.log 2>&1
cat gpurun_out/r19_others_hostmerge.log
