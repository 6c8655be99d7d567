#!/bin/bash
# Everything under profiles/ for one round, on one MI355X:  tools/make_profiles.sh r02   (writes gpurun_out/prof_<tag>/, copy what is wanted)
set -e
set -x
TAG=${1:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py > $O/${TAG}_bench_unprofiled.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-extras --cpu-sample 0 > $O/${TAG}_bench_profiled.json 2>> $O/bench.err
cp $O/stats/*kernel_stats.csv $O/${TAG}_kernel_stats.csv 2>/dev/null || cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats.csv
echo "stats done"
python3 $R/tools/launch_trace.py $O/trace.csv > $O/launch_trace.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o p -- python3 $R/tools/launch_trace.py /tmp/x.csv > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o p -- python3 $R/tools/launch_trace.py /tmp/x.csv > /dev/null 2>&1
U=$(python3 -c "import json;print(json.load(open('$O/${TAG}_bench_unprofiled.json'))['config']['unique_colours'])")
mkdir -p $O/out
python3 $R/tools/make_traffic.py $O/pmc_f $O/pmc_w $O/trace.csv 4096 256 $U $O/out/$TAG > $O/traffic.txt
echo "traffic done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python3 $R/tools/launch_trace.py /tmp/x.csv > /dev/null 2>&1
python3 $R/tools/pmc_rows.py $O/pmc_sq k_rgbw_assign > $O/${TAG}_pmc_sq_assign_per_launch.txt
python3 $R/tools/bench_others.py > $O/${TAG}_others.jsonl 2>> $O/bench.err
python3 $R/bench.py --config c5 2>> $O/bench.err | tail -1 > $O/${TAG}_c5_bench.json
python3 $R/bench.py --config c3 --steps 3 2>> $O/bench.err | tail -1 > $O/${TAG}_c3_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_d16 -o d -- python3 $R/tools/bench_others.py delta16k > /dev/null 2>&1
cp $(find $O/stats_d16 -name '*kernel_stats.csv' | head -1) $O/${TAG}_delta16k_kernel_stats.csv
# voronoi: the FULL run (configs[2], to convergence), per-kernel stats and one PMC pass each for traffic and for what the waves do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_v -o v -- python3 $R/bench.py --config c3 --steps 2 --warmup 1 --cpu-sample 0 > $O/${TAG}_c3_bench_profiled.json 2>> $O/bench.err
cp $(find $O/stats_v -name '*kernel_stats.csv' | head -1) $O/${TAG}_voronoi_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_v1 -o p -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_v3 -o p -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_v2 -o p -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
python3 $R/tools/pmc_kernel_mean.py $O/pmc_v1 k_xy_assign > $O/${TAG}_voronoi_pmc_traffic.txt 2>&1
python3 $R/tools/pmc_kernel_mean.py $O/pmc_v3 k_xy_assign >> $O/${TAG}_voronoi_pmc_traffic.txt 2>&1
python3 $R/tools/pmc_kernel_mean.py $O/pmc_v2 k_xy_assign > $O/${TAG}_voronoi_pmc_sq.txt 2>&1
# ... its mean HBM bytes per launch into traffic.json (key "c3"; bench.py --config c3: roofline.traffic); per-launch durations of the run
python3 - "$O/${TAG}_voronoi_pmc_traffic.txt" "$O/out/traffic.json" <<'PY'
import json, re, sys
t = open(sys.argv[1]).read()
f = float(re.search(r"FETCH_SIZE\s+dispatches\s+\d+\s+total\s+\S+\s+per dispatch\s+(\S+)", t).group(1))
w = float(re.search(r"WRITE_SIZE\s+dispatches\s+\d+\s+total\s+\S+\s+per dispatch\s+(\S+)", t).group(1))
j = json.load(open(sys.argv[2]))
j["c3"] = {"kernel": "k_xy_assign", "size": 4096, "K": 2048, "hbm_bytes_per_launch": int((2 * f + w) * 1024),
           "note": "mean over the launches of two whole runs (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB; reads doubled: gfx950)"}
json.dump(j, open(sys.argv[2], "w"), indent=1)
PY
rocprofv3 --kernel-trace --output-format csv -d $O/trace_v -o t -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
python3 $R/tools/trace_iters.py $(find $O/trace_v -name '*kernel_trace.csv' | head -1) k_xy_assign > $O/${TAG}_voronoi_per_launch_final_us.txt
rm -rf $O/trace_v
# decode (the other half of the trait): bench lines and per-kernel stats
python3 $R/bench.py --decode 2>> $O/bench.err | tail -1 > $O/${TAG}_decode_c2_bench.json
python3 $R/bench.py --decode --config c5 --c5-size 4096 2>> $O/bench.err | tail -1 > $O/${TAG}_decode_delta4096_bench.json
python3 $R/bench.py --decode --config c5 2>> $O/bench.err | tail -1 > $O/${TAG}_decode_c5_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dec -o d -- python3 $R/tools/decode_dev_probe.py 4096 "cluster-colors(256)" delta hufman "hilbert(rle)" > $O/${TAG}_decode_probe.txt 2>&1
cp $(find $O/stats_dec -name '*kernel_stats.csv' | head -1) $O/${TAG}_decode_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dec16 -o d -- python3 $R/tools/decode_dev_probe.py 16384 delta "hilbert(rle)" >> $O/${TAG}_decode_probe.txt 2>&1
cp $(find $O/stats_dec16 -name '*kernel_stats.csv' | head -1) $O/${TAG}_decode_delta16k_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -o c -- python3 $R/bench.py --config c4 --steps 2 --cpu-sample 0 > $O/${TAG}_c4_bench_profiled.json 2>> $O/bench.err
cp $(find $O/stats_c4 -name '*kernel_stats.csv' | head -1) $O/${TAG}_c4_kernel_stats.csv
rm -rf $O/stats $O/stats_d16 $O/stats_v $O/stats_c4 $O/pmc_f $O/pmc_w $O/pmc_sq $O/pmc_v1 $O/pmc_v2 $O/pmc_v3 $O/stats_dec $O/stats_dec16
ls -la $O $O/out
