#!/bin/bash
timeout -k 10 300 python tools/decode_dev_probe.py 4096 "cluster-colors(256)" delta hufman 2>&1 | grep codec
timeout -k 10 300 python tools/decode_dev_probe.py 16384 delta 2>&1 | grep codec
