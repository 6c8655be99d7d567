"""Sum rocprofv3 --pmc counter rows for one kernel, per dispatch index (tools only)."""
import csv, sys, collections
path, name = sys.argv[1], sys.argv[2]
want = [int(x) for x in sys.argv[3:]]  # which launches (by order) of that kernel
rows = [r for r in csv.DictReader(open(path)) if name in r["Kernel_Name"]]
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
for n in want:
    did = ids[n]
    acc = collections.OrderedDict()
    for r in rows:
        if int(r["Dispatch_Id"]) == did:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    print("launch", n, " ".join("%s=%.4g" % kv for kv in acc.items()))
