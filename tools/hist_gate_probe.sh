#!/bin/bash
# k_hist_* with the aggregated atomics' heterogeneity gate (the shipped build) and with all eight rounds (-DCNIIC_ATOMIC_COUNT_EIGHT, tools/build_variant.sh ac8 ...):
# hufman encodes of a photograph at 512^2 / 1024^2 (dense-table histogram) and delta / hufman of a 4096^2 checkerboard (the pathology the rounds are for)
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp; export TMPDIR=/tmp
for lib in libcniic_hip_testing.so libcniic_hip_ac8.so; do
  echo "== $lib"
  rm -rf /tmp/hg
  CNIIC_LIB_FILE=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hg -o h -- python3 $R/tools/all_probe.py 512,1024 > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
for fn in glob.glob("/tmp/hg/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_hist" in r["Name"] or "k_delta_hist" in r["Name"] or "k_sp_hist" in r["Name"]:
            print("  %-70s calls %5s  avg %9.1f ns  total %10.0f ns" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"])))
PY
  CNIIC_LIB_FILE=$lib python3 $R/tools/adversarial_probe.py 4096 "two colours,flat,stripes,half flat" 2>/dev/null | grep -v amdgpu
done
