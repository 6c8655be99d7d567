timeout -k 10 400 python tests/fuzz_codecs.py 150 2>&1 | tail -5
FUZZ_SEED=7 timeout -k 10 400 python tests/fuzz_codecs.py 100 2>&1 | tail -3
