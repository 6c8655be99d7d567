#!/bin/bash
# The round-5 set under profiles/ on one MI355X:  bash tools/make_profiles_r05.sh   (writes gpurun_out/prof_r05/; copy what is wanted)
#   r05_bench_default.json        the driver's command, unprofiled
#   r05_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the same command without the extra blocks: k_rgbw_persist's average = roofline.launch_ms
#   traffic.json                  tools/make_traffic.sh (separate --pmc FETCH_SIZE / WRITE_SIZE passes per block)
#   r05_persist_block_timeline.txt, r05_persist_pmc_sq.txt     the persistent launch per iteration and block; its SQ counters by iteration cap
#   r05_{voronoi,delta16k,decode_delta16k,c4}_kernel_stats.csv per-kernel stats of the other blocks' commands
set -x
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_r05
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py > $O/r05_bench_default.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-extras --cpu-sample 0 > $O/r05_bench_profiled.json 2>> $O/bench.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/r05_kernel_stats.csv
echo "stats done"
bash $R/tools/make_traffic.sh $O/traffic > $O/make_traffic.log 2>&1
cp $O/traffic/traffic.json $O/traffic.json
echo "traffic done"
CNIIC_USE_TESTING_LIB=1 python3 $R/tools/ps_trace.py 4096 256 $O/ps_trace.csv > $O/r05_persist_block_timeline.txt 2>&1
bash $R/tools/ps_pmc.sh > $O/r05_persist_pmc_sq.txt 2>&1
echo "persist done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_v -o v -- python3 $R/bench.py --config c3 --steps 2 --warmup 1 --cpu-sample 0 > $O/r05_c3_bench_profiled.json 2>> $O/bench.err
cp $(find $O/stats_v -name '*kernel_stats.csv' | head -1) $O/r05_voronoi_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_d -o d -- python3 $R/bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 > $O/r05_c5_bench_profiled.json 2>> $O/bench.err
cp $(find $O/stats_d -name '*kernel_stats.csv' | head -1) $O/r05_delta16k_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dd -o d -- python3 $R/bench.py --decode --config c5 --steps 3 --warmup 1 --cpu-sample 0 > $O/r05_c5_decode_bench_profiled.json 2>> $O/bench.err
cp $(find $O/stats_dd -name '*kernel_stats.csv' | head -1) $O/r05_decode_delta16k_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -o c -- python3 $R/bench.py --config c4 --steps 2 --cpu-sample 0 > $O/r05_c4_bench_profiled.json 2>> $O/bench.err
cp $(find $O/stats_c4 -name '*kernel_stats.csv' | head -1) $O/r05_c4_kernel_stats.csv
python3 $R/bench.py --decode --config c5 --cpu-sample 0 2>> $O/bench.err | tail -1 > $O/r05_decode_c5_bench.json
python3 $R/bench.py --decode --cpu-sample 0 2>> $O/bench.err | tail -1 > $O/r05_decode_c2_bench.json
rm -rf $O/stats $O/stats_v $O/stats_d $O/stats_dd $O/stats_c4 $O/traffic/pmc_*
ls -la $O
