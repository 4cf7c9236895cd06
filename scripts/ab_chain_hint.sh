# A/B of the chain kernel's cache hints inside the solve (GPU box): fused and unfused, per hint mask
cd $GRAFT_REPO_ROOT
for fuse in 1 0; do for h in 0 1 2 4 6 3 7; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --set spmv_fuse=$fuse --set chain_hint=$h > gpurun_out/ab_hint.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_hint.json"))
print("fuse $fuse hint $h: %.1f it/s  spmv avg %.1f us" % (d["value"], d["roofline"]["avg_launch_us"]))
PY
done; done
