"""Per-launch durations of one kernel from a rocprofv3 --kernel-trace CSV (tools only, not shipped)."""
import csv, sys
path, name = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path)) if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("launches", len(d), "total_ms", round(sum(d) / 1e3, 3))
for i in range(0, len(d), 10):
    print(i, " ".join("%7.1f" % x for x in d[i:i + 10]))
