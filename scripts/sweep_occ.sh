# How does the pair-code SpMV scale with resident wavefronts?  (spmv_grid = workgroups of 4 wavefronts; 1024 = 4 per SIMD)
cd $GRAFT_REPO_ROOT
for knobs in ${SWEEP:-"spmv_grid=512" "spmv_grid=768" "spmv_grid=1024"}; do
  k2=$(echo $knobs | sed 's/,/ --set /g')
  timeout -k 10 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --set $k2 > gpurun_out/pr.json 2> gpurun_out/pr.err || { tail -3 gpurun_out/pr.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/pr.json"))
print("%-50s %7.1f it/s  %.4f ms/it spmv %.1f us" % ("$knobs", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
done
