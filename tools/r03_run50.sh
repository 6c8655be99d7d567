#!/bin/bash
set -e
R=$(pwd); O=$R/gpurun_out/r50; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/tools/batch_probe.py 64 8 > $O/probe.txt 2>&1
python3 - $(find $O/t -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last batch call: take the last 40% of rows by time as a proxy; compute union busy time and sum of durations over the final 45 ms window
t_end = max(int(r["End_Timestamp"]) for r in rows)
win = [r for r in rows if int(r["Start_Timestamp"]) > t_end - 45_000_000]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
tot = sum(e - s for s, e in iv)
span = iv[-1][1] - iv[0][0]
print("window %.1f ms: %d kernels, sum of durations %.1f ms, union busy %.1f ms, mean duration %.1f us, overlap factor %.2f" % (span / 1e6, len(iv), tot / 1e6, busy / 1e6, tot / len(iv) / 1e3, tot / busy))
PY
rm -rf $O/t; tail -2 $O/probe.txt
