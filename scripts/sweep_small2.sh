cd $GRAFT_REPO_ROOT
for wl in poisson2d banded complex; do
for knobs in ${SWEEP:-"ew_chunk=0" "ew_chunk=1" "ew_chunk=0" "ew_chunk=1"}; do
  k2=$(echo $knobs | sed 's/,/ --set /g')
  timeout -k 10 150 python bench.py --workload $wl --steps 500 --warmup 50 --no-cpu-baseline --no-also --set $k2 > gpurun_out/s2.json 2> gpurun_out/s2.err || { tail -3 gpurun_out/s2.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/s2.json"))
print("%-10s %-34s %9.1f it/s  %.4f ms/it" % ("$wl", "$knobs", d["value"], d["ms_per_step"]))
PY
done
done
