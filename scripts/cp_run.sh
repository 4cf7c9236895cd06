set -e
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite.txt 2>&1 || { tail -30 gpurun_out/gpu_suite.txt; exit 1; }
tail -3 gpurun_out/gpu_suite.txt
for st in csr auto; do
  python bench.py --workload complex --stream $st --no-cpu-baseline --no-also --steps 100 --warmup 10 > gpurun_out/cp_bench_$st.json 2> gpurun_out/cp_bench_$st.err || { tail -5 gpurun_out/cp_bench_$st.err; exit 1; }
  python - $st <<'PY'
import json, sys
j = json.loads(open("gpurun_out/cp_bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "%.0f it/s" % j["value"], "spmv %.2f us" % r["avg_launch_us"], "frac %.3f" % r["frac"], "create %.1f ms" % j["create_ms"], r["kernel"][:50])
PY
done
