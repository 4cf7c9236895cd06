cd $GRAFT_REPO_ROOT
for knobs in "xcd_chunk=0" "xcd_chunk=1" "xcd_chunk=1 --set spmv_nt=1" "xcd_chunk=0 --set spmv_grid=1536" "xcd_chunk=1 --set spmv_grid=1536" "xcd_chunk=0 --set spmv_grid=768" "xcd_chunk=1 --set spmv_grid=768" "xcd_chunk=0"; do
  tag=$(echo "rnd_$knobs" | tr -d ' =-')
  timeout -k 10 100 python bench.py --values random --steps 20 --warmup 5 --no-cpu-baseline --no-also --set $knobs > gpurun_out/$tag.json 2> gpurun_out/$tag.err || { tail -3 gpurun_out/$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/$tag.json"))
print("%-44s %7.1f it/s  spmv %.1f us frac %.3f" % ("$knobs", d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
