#!/bin/bash
set -e
for cap in 1 2 3 4 6 10 20 60; do timeout -k 10 120 python tools/voronoi_probe.py 4096 2048 $cap 2>/dev/null | tail -1; done
