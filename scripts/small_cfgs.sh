# the three cache-resident BASELINE configs, default knobs, 2 repetitions each
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for wl in poisson2d banded complex; do
  timeout -k 10 100 python bench.py --workload $wl --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/small_$wl.json 2> gpurun_out/small_$wl.err || { tail -3 gpurun_out/small_$wl.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/small_$wl.json"))
print("%-10s %9.0f it/s  %.2f us/it  spmv %.1f us" % ("$wl", d["value"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_us"]))
PY
done; done
