# cache-resident configs (cfg 2/3/4): element-wise grid and poll interval sweep
cd $GRAFT_REPO_ROOT
for wl in poisson2d banded complex; do
  for knobs in "grid=512" "grid=1024" "grid=2048" "grid=4096" "grid=2048 --set poll=64" "grid=2048 --set spmv_grid=2048" "grid=2048 --set spmv_grid=512"; do
    tag=$(echo "${wl}_$knobs" | tr -d ' =-')
    timeout -k 10 100 python bench.py --workload $wl --steps 1000 --warmup 100 --no-cpu-baseline --set $knobs > gpurun_out/sw_$tag.json 2> gpurun_out/sw_$tag.err || { tail -3 gpurun_out/sw_$tag.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/sw_$tag.json"))
print("%-10s %-40s %9.0f it/s  %.2f us/it  spmv %.1f us" % ("$wl", "$knobs", d["value"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_us"]))
PY
  done
done
