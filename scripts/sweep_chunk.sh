# cross-over of the XCD-chunked walk of the fused kernels by vector size (grid 500x500xNZ: NZ*2 MB per vector)
cd $GRAFT_REPO_ROOT
for nz in ${NZS:-2 4 8 16 25 50 200}; do
  for ch in 0 1 0 1; do
    timeout -k 10 150 python bench.py --grid 500x500x$nz --steps 100 --warmup 10 --no-cpu-baseline --no-also --set ew_chunk=$ch > gpurun_out/ch.json 2> gpurun_out/ch.err || { tail -3 gpurun_out/ch.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/ch.json"))
print("nz=%-4s (%4d MB/vector) ew_chunk=$ch %9.1f it/s  %.4f ms/it" % ("$nz", $nz*2, d["value"], d["ms_per_step"]))
PY
  done
done
