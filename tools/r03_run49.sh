#!/bin/bash
for mb in 48 96 192 384; do
CNIIC_KM_MAX_BLOCKS=$mb timeout -k 10 300 python tools/batch_probe.py 64 8,16,32 2>&1 | grep streams
done
