cd $GRAFT_REPO_ROOT
for knobs in "spmv_period=0" "spmv_period=1" "spmv_period=0 --set spmv_grid=1280" "spmv_period=1 --set spmv_grid=1280" "spmv_period=0" "spmv_period=1"; do
  timeout -k 10 100 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --set $knobs > gpurun_out/pr.json 2> gpurun_out/pr.err || { tail -3 gpurun_out/pr.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/pr.json"))
print("%-40s %7.1f it/s  %.4f ms/it spmv %.1f us" % ("$knobs", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
done
