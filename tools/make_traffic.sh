#!/bin/bash
# HBM bytes per launch of every bench block's dominant kernel, from the PMC counters (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes with --kernel-trace only; FETCH_SIZE doubled on gfx950) -> profiles/traffic.json keys
#   c2_persist  k_rgbw_persist        the headline encode's one K-means launch
#   c4          k_rgbw_persist        configs[3]'s one-GPU share (128 frames, one palette)
#   c5          k_delta_gather_p2     configs[4]
#   c2_decode   k_hd_write            decode of the headline stream        c5_decode   k_hd_compact decode of configs[4]'s stream (kept symbols copied to their places)
# usage (on the GPU box): bash tools/make_traffic.sh [outdir]   -- then copy outdir/traffic.json over profiles/traffic.json
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-$R/gpurun_out/traffic}
mkdir -p $O
O=$(cd "$O" && pwd)
export TMPDIR=/tmp
cd /tmp
cp $R/profiles/traffic.json $O/traffic.json
run() {   # key kernel-substring command...
  key=$1; pat=$2; shift 2
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$ctr
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o p -- "$@" > /dev/null 2>&1
  done
  python3 - "$O" "$key" "$pat" <<'PY'
import collections, csv, glob, json, sys
O, key, pat = sys.argv[1:4]
tot, disp = collections.defaultdict(float), collections.defaultdict(set)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(O + "/pmc_" + ctr + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                tot[ctr] += float(r["Counter_Value"]); disp[ctr].add(r["Dispatch_Id"])
n = max(1, len(disp["FETCH_SIZE"]))
f, w = tot["FETCH_SIZE"] / n, tot["WRITE_SIZE"] / max(1, len(disp["WRITE_SIZE"]))
j = json.load(open(O + "/traffic.json"))
j[key] = {"kernel": pat, "hbm_bytes_per_launch": int((2 * f + w) * 1024), "fetch_KiB_raw_per_launch": round(f, 1), "write_KiB_per_launch": round(w, 1), "launches": n,
          "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KiB), mean over the kernel's launches of the command; reads doubled (gfx950: 128-B requests tallied at 64 B)"}
json.dump(j, open(O + "/traffic.json", "w"), indent=1)
print(key, pat, j[key])
PY
}
run c2_persist k_rgbw_persist python3 $R/bench.py --no-extras --cpu-sample 0 --steps 3 --warmup 1
run c4 k_rgbw_persist python3 $R/bench.py --config c4 --cpu-sample 0 --steps 1 --warmup 1
run c5 k_delta_gather_p2 python3 $R/bench.py --config c5 --cpu-sample 0 --steps 2 --warmup 1
run c2_decode k_hd_write python3 $R/bench.py --decode --cpu-sample 0 --steps 2 --warmup 1
run c5_decode k_hd_compact python3 $R/bench.py --decode --config c5 --cpu-sample 0 --steps 2 --warmup 1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/traffic.json
