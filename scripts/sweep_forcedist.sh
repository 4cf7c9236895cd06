# one rank, real RCCL: the distributed code path (halo plan, split launches, all-reduced scalars) at per-rank sizes
cd $GRAFT_REPO_ROOT
for nz in 200 100 50 25; do
  for mode in "" "--force-dist"; do
    timeout -k 10 200 python bench.py --grid 500x500x$nz --steps 40 --warmup 5 --no-cpu-baseline --no-also $mode > gpurun_out/fd.json 2> gpurun_out/fd.err || { tail -3 gpurun_out/fd.err; exit 1; }
    [ $(wc -l < gpurun_out/fd.json) -eq 1 ] || { echo "stdout is not ONE line:"; cat gpurun_out/fd.json | cut -c1-100; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/fd.json"))
print("nz=%-4s %-12s %9.1f it/s  %.4f ms/it" % ("$nz", "$mode" or "plain", d["value"], d["ms_per_step"]))
PY
  done
done
