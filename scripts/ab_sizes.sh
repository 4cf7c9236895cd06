# alternating A/B of two library builds over problem sizes (per-rank sizes of N = 8 / 4 / 2 / 1 and the small configs)
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  cp sprsolve_amd/ab_$lib.so sprsolve_amd/libsprsolve_hip.so
  for g in 500x500x25 500x500x50 500x500x100 500x500x200; do
    timeout -k 10 150 python bench.py --grid $g --steps 40 --warmup 5 --no-cpu-baseline --no-also > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/ab.json"))
print("%-6s %-12s %9.1f it/s  %.4f ms/it  spmv %.1f us" % ("$lib", "$g", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
  done
  for wl in poisson2d banded complex; do
    timeout -k 10 150 python bench.py --workload $wl --steps 500 --warmup 50 --no-cpu-baseline --no-also > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/ab.json"))
print("%-6s %-12s %9.1f it/s  %.4f ms/it" % ("$lib", "$wl", d["value"], d["ms_per_step"]))
PY
  done
done
