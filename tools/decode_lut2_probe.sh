#!/bin/bash
# decode of configs[4]'s stream over the second table's size (CNIIC_HD_LUT2_BITS, testing build; the third tables resolve what it leaves)
R=$(cd "$(dirname "$0")/.." && pwd)
for b in 13 14 15 16 17 18 19 20; do
  CNIIC_USE_TESTING_LIB=1 CNIIC_HD_LUT2_BITS=$b python3 $R/bench.py --decode --config c5 --cpu-sample 0 --steps 4 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c5 lut2 bits $b:', d['ms_per_step'], 'ms', d['stages'])"
done
