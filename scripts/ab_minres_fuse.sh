# same box: cfg 3 / cfg 4 with M3 inside M1 (default) and without (spmv_fuse=0)
for w in "complex" "banded"; do
  for f in 0 -1 1 0 -1 1; do
    python bench.py --workload $w --no-cpu-baseline --no-also --steps 500 --warmup 50 --set spmv_fuse=$f > gpurun_out/mf_${w}_$f.json 2> gpurun_out/mf_${w}_$f.err || { tail -5 gpurun_out/mf_${w}_$f.err; exit 1; }
    python - $w $f <<'PY'
import json, sys
j = json.loads(open("gpurun_out/mf_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "spmv_fuse", sys.argv[2], "%.0f it/s  %.2f us/iteration" % (j["value"], j["ms_per_step"] * 1e3), "| SpMV launch %.2f us" % r["avg_launch_us"], "frac %.3f" % r["frac"])
PY
  done
done
