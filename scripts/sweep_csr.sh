cd $GRAFT_REPO_ROOT
for knobs in ${SWEEP:-"spmv_nt=-1" "spmv_nt=1" "spmv_nt=-1" "spmv_nt=1"}; do
  k2=$(echo $knobs | sed 's/,/ --set /g')
  timeout -k 10 150 python bench.py --stream csr --steps 20 --warmup 5 --no-cpu-baseline --no-also --set $k2 > gpurun_out/csr.json 2> gpurun_out/csr.err || { tail -3 gpurun_out/csr.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/csr.json"))
print("%-30s csr %7.1f it/s  %.4f ms/it  spmv %.1f us (%.3f)" % ("$knobs", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
