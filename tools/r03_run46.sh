#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_huffman_optimal.py tests/test_decode_device.py tests/test_boundary.py tests/test_golden.py -m gpu -x -q > gpurun_out/r46_tests.log 2>&1 || { tail -30 gpurun_out/r46_tests.log; exit 1; }
tail -2 gpurun_out/r46_tests.log
timeout -k 10 200 python tests/fuzz_codecs.py 60 2>&1 | tail -2
timeout -k 10 200 python tools/bench_others.py hufman delta delta16k 2>&1 | tail -3
