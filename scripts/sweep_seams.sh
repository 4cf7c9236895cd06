# How much do the non-uniform (seam) blocks cost the pair-code kernel?  Same n = 50 M, different x-line lengths.
cd $GRAFT_REPO_ROOT
for g in 500x500x200 2000x125x200 5000x100x100 250x500x400; do
  timeout -k 10 150 python bench.py --grid $g --steps 30 --warmup 5 --no-cpu-baseline --no-also > gpurun_out/seam.json 2> gpurun_out/seam.err || { tail -3 gpurun_out/seam.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/seam.json"))
s=d["config"]["spmv_stream"]
print("%-14s %7.1f it/s  %.4f ms/it  spmv %.1f us  uniform %d of %d blocks" % ("$g", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], s["uniform_blocks"], s["row_blocks"]))
PY
done
