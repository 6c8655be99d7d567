CNIIC_HD_STATS=1 timeout -k 10 300 python tools/decode_dev_probe.py 4096 "cluster-colors(256)" delta 2>&1 | grep "^\[hd\]\|codec" | tail -24
CNIIC_HD_STATS=1 timeout -k 10 300 python tools/decode_dev_probe.py 16384 delta 2>&1 | grep "^\[hd\]\|codec" | tail -8
