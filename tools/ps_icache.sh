#!/bin/bash
# instruction-cache counters of the persistent K-means launch on the headline image
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  rm -rf $R/gpurun_out/pmc_tmp
  PS_BLOCKS_TRACE=0 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_tmp -o ps --output-format csv -- python3 $R/tools/ps_trace.py 4096 256 $R/gpurun_out/ps_trace_pmc.csv > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_tmp/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "persist" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print("%-28s %16.0f per launch (%d launches)" % (k, v / n, n))
PY
done
rm -rf $R/gpurun_out/pmc_tmp
