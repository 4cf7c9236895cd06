set -e
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite_final.txt 2>&1 || { tail -30 gpurun_out/gpu_suite_final.txt; exit 1; }
tail -3 gpurun_out/gpu_suite_final.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_n1_steps20.json 2> gpurun_out/bench_n1_steps20.err || { tail -5 gpurun_out/bench_n1_steps20.err; exit 1; }
python scripts/show_bench.py gpurun_out/bench_n1_steps20.json
