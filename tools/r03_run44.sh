#!/bin/bash
for cap in 10 20 40 80 182; do CNIIC_XY_MARGINS=4 timeout -k 10 120 python tools/voronoi_probe.py 4096 2048 $cap 2>&1 | grep "xy margins" | tail -1 | sed "s/^/cap $cap: /"; done
