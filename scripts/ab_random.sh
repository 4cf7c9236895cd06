# alternating A/B of library builds on the random-values cfg-5 leg: bash scripts/ab_random.sh libA libB ...
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for lib in "$@"; do
  cp sprsolve_amd/ab_$lib.so sprsolve_amd/libsprsolve_hip.so
  timeout -k 10 150 python bench.py --values random --steps 20 --warmup 5 --no-cpu-baseline --no-also > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab.json"))
print("%-8s random %7.1f it/s  %.4f ms/it  spmv %.1f us (%.3f)" % ("$lib", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
done
