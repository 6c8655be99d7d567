set -x
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03a/pytest.log
tail -5 gpurun_out/r03a/pytest.log
python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc $?"
CNIIC_TRACE_HOST=1 python tools/decode_probe.py 4096 > gpurun_out/r03a/decode_probe.txt 2>&1
python tools/decode_probe.py 16384 delta >> gpurun_out/r03a/decode_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03a/prof_decode -o dec -- python3 $GRAFT_REPO_ROOT/tools/decode_probe.py 4096 "cluster-colors(256)" delta hufman > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/r03a | head -30
cat gpurun_out/r03a/decode_probe.txt | tail -60
