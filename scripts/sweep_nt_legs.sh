cd $GRAFT_REPO_ROOT
for leg in "--stream csr" "--values random"; do
  for nt in 0 1 0 1; do
    timeout -k 10 150 python bench.py $leg --steps 20 --warmup 5 --no-cpu-baseline --no-also --set stream_nt=$nt > gpurun_out/nt.json 2> gpurun_out/nt.err || { tail -3 gpurun_out/nt.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/nt.json"))
print("%-16s stream_nt=$nt %7.1f it/s  %.4f ms/it  spmv %.1f us (%.3f)" % ("$leg", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
  done
done
