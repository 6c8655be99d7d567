#!/usr/bin/env python3
"""Per-counter totals and per-dispatch means of one kernel from a rocprofv3 --pmc run (counter_collection.csv).
    python tools/pmc_kernel_mean.py DIR kernel_name_substring
FETCH_SIZE / WRITE_SIZE are printed raw (KiB) and as HBM bytes per dispatch with the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte requests as 64: x2)."""
import collections, csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(float)
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Counter_Name"]].add(r["Dispatch_Id"])
print("kernel", pat)
for k in sorted(tot):
    n = len(disp[k])
    print("%-22s dispatches %6d  total %.6g  per dispatch %.6g" % (k, n, tot[k], tot[k] / max(1, n)))
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    n = len(disp["FETCH_SIZE"])
    hbm = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / max(1, n)
    print("HBM bytes per dispatch (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes): %.4g" % hbm)
if "SQ_WAVE_CYCLES" in tot:
    wc = tot["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in tot:
            print("%s / SQ_WAVE_CYCLES = %.3f" % (k, tot[k] / wc))
    # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles (MI355X_MICROARCH.md); a VALU instruction occupies its SIMD for one
    if "SQ_INSTS_VALU" in tot:
        f = tot["SQ_INSTS_VALU"] / wc
        print("SQ_INSTS_VALU / SQ_WAVE_CYCLES (quad-cycles) = %.3f of a wave's time issuing VALU" % f)
        if "SQ_WAVES" in tot:
            print("  (x resident waves per SIMD = the SIMD's VALU issue utilisation; e.g. 4 waves per SIMD: %.2f)" % (4 * f))
