# Profiles of the default bench.py command (run on the GPU box): kernel-trace stats + the two PMC passes.
#   usage: bash scripts/profile_bench.sh <tag> [extra bench.py flags]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-also --steps 100 --warmup 5 $*"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -- python3 $B > gpurun_out/prof_${tag}_stats.json 2> gpurun_out/prof_${tag}_stats.err || exit 1
echo "stats pass done"
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_${tag}_fetch -- python3 $B > gpurun_out/prof_${tag}_fetch.json 2> gpurun_out/prof_${tag}_fetch.err || exit 1
echo "fetch pass done"
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_${tag}_write -- python3 $B > gpurun_out/prof_${tag}_write.json 2> gpurun_out/prof_${tag}_write.err || exit 1
echo "write pass done"
python scripts/make_pmc_summary.py gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write $tag gpurun_out/pmc_summary_${tag}.json
cp $(ls gpurun_out/prof_${tag}_stats/*/*kernel_stats.csv | head -1) gpurun_out/kernel_stats_${tag}.csv
python scripts/show_bench.py gpurun_out/prof_${tag}_stats.json
