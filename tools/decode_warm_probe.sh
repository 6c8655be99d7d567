#!/bin/bash
# decode of configs[4] / configs[1] streams over the warm-up length of pass 0 (CNIIC_HD_WARM, testing build)
R=$(cd "$(dirname "$0")/.." && pwd)
for cfg in c5 c2; do
for w in 64 128 192 256 320 384 480; do
  CNIIC_USE_TESTING_LIB=1 CNIIC_HD_WARM=$w python3 $R/bench.py --decode --config $cfg --cpu-sample 0 --steps 4 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg warm $w:', d['ms_per_step'], 'ms', d['stages'])"
done
done
