set -e
for w in "complex csr" "complex auto" "banded auto" "poisson2d auto"; do
  set -- $w
  python bench.py --workload $1 --stream $2 --no-cpu-baseline --no-also --steps 500 --warmup 50 > gpurun_out/small_$1_$2.json 2> gpurun_out/small_$1_$2.err || { tail -5 gpurun_out/small_$1_$2.err; exit 1; }
  python - $1 $2 <<'PY'
import json, sys
j = json.loads(open("gpurun_out/small_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], sys.argv[2], "%.0f it/s  %.2f us/iteration" % (j["value"], j["ms_per_step"] * 1e3), "spmv %.2f us" % r["avg_launch_us"], "frac %.3f" % r["frac"], j["timing"].get("profiled_pass", {}).get("ms_per_step_with_events"))
PY
done
