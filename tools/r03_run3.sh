set -x
mkdir -p gpurun_out/r03c
python bench.py --decode --cpu-sample 0 > gpurun_out/r03c/dec_c2.json 2> gpurun_out/r03c/dec_c2.err; echo "rc $?"; tail -3 gpurun_out/r03c/dec_c2.err
python bench.py --decode --config c5 --c5-size 4096 --cpu-sample 0 > gpurun_out/r03c/dec_d4k.json 2> gpurun_out/r03c/dec_d4k.err; echo "rc $?"; tail -3 gpurun_out/r03c/dec_d4k.err
python bench.py --decode --config c5 --cpu-sample 0 > gpurun_out/r03c/dec_c5.json 2> gpurun_out/r03c/dec_c5.err; echo "rc $?"; tail -3 gpurun_out/r03c/dec_c5.err
cat gpurun_out/r03c/dec_c2.json gpurun_out/r03c/dec_d4k.json gpurun_out/r03c/dec_c5.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03c/prof_decode -o dec -- python3 $GRAFT_REPO_ROOT/tools/decode_probe.py 4096 "cluster-colors(256)" delta hufman > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_stats.py gpurun_out/r03c/prof_decode | head -40
