# SQ counters of the persistent K-means launch on the headline image (k_rgbw_persist), at several iteration caps: the differences are the
# instruction counts of the iteration groups.  bash tools/ps_pmc.sh > gpurun_out/ps_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for it in ${PS_PMC_ITERS:-1 13 25 0}; do
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  rm -rf $R/gpurun_out/pmc_tmp
  PS_MAX_ITERS=$it PS_BLOCKS_TRACE=0 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_tmp -o ps --output-format csv -- python3 $R/tools/ps_trace.py 4096 256 $R/gpurun_out/ps_trace_pmc.csv > /dev/null 2>&1
  echo "== iteration cap $it"
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_tmp/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "persist" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print("%-24s %16.0f per launch (%d launches)" % (k, v / n, n))
PY
done
done
rm -rf $R/gpurun_out/pmc_tmp
