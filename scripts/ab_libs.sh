# A/B of two builds of the library (sprsolve_amd/ab_<name>.so, git-ignored) on the small configs and the headline.
#   usage: bash scripts/ab_libs.sh nameA nameB ...
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for lib in "$@"; do
  cp sprsolve_amd/ab_$lib.so sprsolve_amd/libsprsolve_hip.so
  for wl in ${WORKLOADS:-poisson2d banded complex poisson3d}; do
    extra="--steps 500 --warmup 50"; [ $wl = poisson3d ] && extra="--steps 30 --warmup 5"
    timeout -k 10 150 python bench.py --workload $wl $extra --no-cpu-baseline --no-also > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/ab.json"))
print("%-8s %-10s %9.1f it/s  %.4f ms/it" % ("$lib", "$wl", d["value"], d["ms_per_step"]))
PY
  done
done
done
