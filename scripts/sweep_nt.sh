# cross-over of non-temporal operand loads in the fused kernels by vector size (grid 500x500xNZ: NZ*2 MB per vector)
cd $GRAFT_REPO_ROOT
for nz in ${NZS:-25 32 40 50 100}; do
  for nt in 0 1 0 1; do
    timeout -k 10 150 python bench.py --grid 500x500x$nz --steps 60 --warmup 5 --no-cpu-baseline --no-also --set stream_nt=$nt > gpurun_out/nt.json 2> gpurun_out/nt.err || { tail -3 gpurun_out/nt.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/nt.json"))
print("nz=%-4s (%4d MB/vector) stream_nt=$nt %9.1f it/s  %.4f ms/it" % ("$nz", $nz*2, d["value"], d["ms_per_step"]))
PY
  done
done
