#!/bin/bash
set -e
for m in 0 4; do for cap in 4 6 10 20 40 80 182; do CNIIC_XY_MARGINS=$m timeout -k 10 120 python tools/voronoi_probe.py 4096 2048 $cap 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('margins $m cap $cap: iters', d['iters'], 'ms', round(d['sec']*1e3,2), 'cand/px/iter', d['cand_per_px'], 'moved_last', d['moved_last'])"; done; done
