# full GPU check on the box: tests, then the default bench (as the driver runs it)
cd $GRAFT_REPO_ROOT
if [ "$1" != "benchonly" ]; then
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gputests.txt 2>&1; rc=$?; tail -15 gpurun_out/gputests.txt
[ $rc -eq 0 ] || exit $rc
fi
SECONDS=0; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; rc=$?
echo "bench wall: ${SECONDS}s"; tail -3 gpurun_out/bench_default.err
python scripts/show_bench.py gpurun_out/bench_default.json 2>/dev/null | head -60 || cat gpurun_out/bench_default.json | cut -c1-3000
exit $rc
