#!/usr/bin/env python3
"""Prints per-dispatch PMC counters of one kernel from a rocprofv3 --pmc run (counter_collection.csv): dispatches of the LAST encode."""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
want = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
n = int(sys.argv[4]) if len(sys.argv) > 4 else 68
ids = ids[-n:]
names = sorted({k for i in ids for k in by[i]})
print("launch " + " ".join(names))
for j, i in enumerate(ids):
    if want is None or j in want:
        print(j, " ".join("%.4g" % by[i].get(k, float("nan")) for k in names))
