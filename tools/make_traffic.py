#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of tools/prof_assign.py (FETCH_SIZE, WRITE_SIZE in separate runs) into
profiles/traffic.json, read by bench.py for roofline.traffic.  gfx950 correction (MI355X_MICROARCH.md, HBM):
FETCH_SIZE counts 128-B read requests at 64 B, so the read side is doubled; WRITE_SIZE is taken as is; both
are in KiB.   usage: make_traffic.py <fetch_dir> <write_dir> <size> <K> <U>"""
import collections
import csv
import json
import os
import sys


def mean_counter(d, name, kernel):
    rows = list(csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))))
    v = [float(r["Counter_Value"]) for r in rows if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    v = v[len(v) // 2:]  # steady-state half
    return sum(v) / len(v), len(v)


fetch_dir, write_dir, size, K, U = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
f, nf = mean_counter(fetch_dir, "FETCH_SIZE", "assign_cells")
w, nw = mean_counter(write_dir, "WRITE_SIZE", "assign_cells")
out = {"kernel": "k_rgbw_assign_cells", "size": size, "K": K, "unique_colours": U,
       "FETCH_SIZE_KiB_raw": round(f, 1), "WRITE_SIZE_KiB_raw": round(w, 1), "launches_averaged": nf,
       "hbm_bytes_per_launch": int((2 * f + w) * 1024),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_assign.py; "
                 "read side doubled (gfx950 FETCH_SIZE counts 128-B requests at 64 B); steady-state launches (no point moves)"}
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1)
print(out)
