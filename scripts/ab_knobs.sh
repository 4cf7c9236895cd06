# alternating A/B of (library build, knob set) pairs on the headline: bash scripts/ab_knobs.sh "lib:k=v,k=v" "lib:" ...
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for spec in "$@"; do
  lib=${spec%%:*}; knobs=${spec#*:}
  cp sprsolve_amd/ab_$lib.so sprsolve_amd/libsprsolve_hip.so
  k2=""; [ -n "$knobs" ] && k2="--set $(echo $knobs | sed 's/,/ --set /g')"
  timeout -k 10 150 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also $k2 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab.json"))
print("%-36s %7.1f it/s  %.4f ms/it  spmv %.1f us" % ("$spec", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
done
done
