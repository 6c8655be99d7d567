timeout -k 10 300 python tools/batch_probe.py 64 1,8,16 2>&1 | tail -6
CNIIC_KM_MAX_BLOCKS=192 timeout -k 10 300 python tools/batch_probe.py 64 1,8,16 2>&1 | tail -4
CNIIC_KM_MAX_BLOCKS=96 timeout -k 10 300 python tools/batch_probe.py 64 8,16,32 2>&1 | tail -4
CNIIC_KM_MAX_BLOCKS=384 timeout -k 10 300 python tools/batch_probe.py 64 8,16 2>&1 | tail -3
timeout -k 10 300 python tools/decode_dev_probe.py 4096 hufman 2>&1 | grep codec
