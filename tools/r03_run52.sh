#!/bin/bash
set -e
R=$(pwd); O=$R/gpurun_out/r52; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o s -- python3 $R/tools/bench_others.py delta16k > /dev/null 2>&1
cp $(find $O/s -name '*kernel_stats.csv' | head -1) $O/delta16k_kernel_stats.csv; rm -rf $O/s
python3 - $O/delta16k_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    n = r['Name'].split('(')[0].replace('cniic::', '').replace('void ', '')
    print("%-28s calls %4s  avg %9.1f us  per encode %8.1f us" % (n[:28], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3 / 4))
PY
