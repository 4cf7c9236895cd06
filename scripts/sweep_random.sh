# cfg-5 pattern with random values (offset-code stream): walk order knobs
cd $GRAFT_REPO_ROOT
for knobs in ${SWEEP:-"spmv_period=0" "spmv_period=1" "spmv_period=0" "spmv_period=1"}; do
  k2=$(echo $knobs | sed 's/,/ --set /g')
  timeout -k 10 100 python bench.py --values random --steps 20 --warmup 5 --no-cpu-baseline --no-also --set $k2 > gpurun_out/rnd.json 2> gpurun_out/rnd.err || { tail -3 gpurun_out/rnd.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/rnd.json"))
print("%-44s %7.1f it/s  spmv %.1f us frac %.3f" % ("$knobs", d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
