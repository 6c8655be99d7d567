#!/usr/bin/env python3
"""HBM traffic of every assign launch of ONE real encode: two rocprofv3 passes over tools/launch_trace.py (--pmc FETCH_SIZE and
--pmc WRITE_SIZE, separately: they do not fit one pass on gfx950) joined with the un-profiled per-launch trace (durations,
classes) of the same encode.  Writes
    profiles/rNN_launch_trace.csv   one line per launch: duration, class, centroids / points moved, FETCH / WRITE KiB, HBM bytes
    profiles/traffic.json           per class means; `hbm_bytes_per_launch` = the full-schedule launches (bench.py: roofline.traffic)
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B read requests at 64 B, so the read side is doubled;
WRITE_SIZE is taken as is; both are in KiB.
    usage: make_traffic.py <fetch_dir> <write_dir> <trace.csv> <size> <K> <U> <out_prefix e.g. profiles/r02>"""
import collections
import csv
import glob
import json
import os
import sys


def per_dispatch(d, name, kernel, last):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
            by[int(r["Dispatch_Id"])] = by.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    ids = sorted(by)[-last:]          # the last encode of the process
    return [by[i] for i in ids]


fetch_dir, write_dir, trace, size, K, U, prefix = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
rows = list(csv.DictReader(open(trace)))
n = len(rows)
fe = per_dispatch(fetch_dir, "FETCH_SIZE", "k_rgbw_assign", n)
wr = per_dispatch(write_dir, "WRITE_SIZE", "k_rgbw_assign", n)
assert len(fe) == n and len(wr) == n, (len(fe), len(wr), n)
cls = collections.OrderedDict()
with open(prefix + "_launch_trace.csv", "w") as f:
    f.write("launch,us,class,centroids_moved_before,points_moved,FETCH_SIZE_KiB_raw,WRITE_SIZE_KiB_raw,hbm_bytes\n")
    for r, a, b in zip(rows, fe, wr):
        hbm = int((2 * a + b) * 1024)
        f.write("%s,%s,%s,%s,%s,%.1f,%.1f,%d\n" % (r["launch"], r["us"], r["class"], r["centroids_moved_before"], r["points_moved"], a, b, hbm))
        c = cls.setdefault(r["class"], {"launches": 0, "us": 0.0, "hbm_bytes": 0})
        c["launches"] += 1; c["us"] += float(r["us"]); c["hbm_bytes"] += hbm
for c in cls.values():
    c["us"] = round(c["us"] / c["launches"], 2)
    c["hbm_bytes"] = int(c["hbm_bytes"] / c["launches"])
    c["hbm_GBps"] = round(c["hbm_bytes"] / (c["us"] * 1e-6) / 1e9, 1)
out = {"kernel": "k_rgbw_assign_cells", "size": size, "K": K, "unique_colours": U, "algorithmic_bytes_per_launch": 10 * U,
       "hbm_bytes_per_launch": cls.get("full", {}).get("hbm_bytes"), "by_class": cls,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/launch_trace.py (one real encode, every assign "
                 "dispatch), joined with the un-profiled per-launch trace; read side doubled (gfx950 FETCH_SIZE counts 128-B requests at 64 B); "
                 "hbm_bytes_per_launch = mean over the full-schedule launches; per-launch table: " + os.path.basename(prefix) + "_launch_trace.csv"}
json.dump(out, open(os.path.join(os.path.dirname(prefix) or ".", "traffic.json"), "w"), indent=1)
print(json.dumps(out))
