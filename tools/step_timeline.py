#!/usr/bin/env python3
"""Where one cluster-colors encode spends its wall time outside the K-means launches: reads a `rocprofv3 --kernel-trace
--output-format csv` trace of tools/trace_host.py (four encodes) and prints, for the last encode, every kernel with its start
(relative), duration and the gap before it, the K-means launches folded into one line.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o k -- python3 tools/trace_host.py
    python3 tools/step_timeline.py gpurun_out/kt [first-kernel-of-an-encode [which-encode]]
"""
import csv
import glob
import os
import sys

f = [p for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)][0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the encodes start with the first partition kernel (k_sp_count); take the last one
first = sys.argv[2] if len(sys.argv) > 2 else "k_sp_count"   # the kernel an encode starts with
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # which encode of the trace
starts = [i for i, n in enumerate(names) if first in n]
a = starts[which]
seg = rows[a:(starts[which + 1] if which != -1 and which + 1 < len(starts) else len(rows))]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
km = None
total_gap = 0
print("%9s %9s %8s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cniic::", "")
    if "k_rgbw_assign_cells" in n:
        if km is None:
            km = [s, e, 1]
        else:
            km[1] = e; km[2] += 1
        prev_end = e
        continue
    if km is not None:
        print("%9.1f %9.1f %8s  k_rgbw_assign_cells x %d (start of the first to the end of the last)" % ((km[0] - t0) / 1e3, (km[1] - km[0]) / 1e3, "", km[2]))
        km = None
    gap = (s - prev_end) / 1e3
    total_gap += max(0.0, gap)
    print("%9.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n[:70]))
    prev_end = max(prev_end, e)
print("encode: %.1f us from the first kernel's start to the last one's end; gaps outside the K-means loop %.1f us" % ((prev_end - t0) / 1e3, total_gap))
