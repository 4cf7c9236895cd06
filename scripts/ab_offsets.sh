# A/B of the offset-code SpMV (cfg-5 pattern, random values) inside the solve: value-load width x grid
#   usage: bash scripts/ab_offsets.sh  -> gpurun_out/ab_offsets.txt
rm -f gpurun_out/ab_offsets.txt
for knobs in ${SWEEP:-"spmv_wideload=0" "spmv_wideload=1" "spmv_wideload=0,spmv_grid=768" "spmv_wideload=1,spmv_grid=768" "spmv_wideload=0" "spmv_wideload=1" "spmv_wideload=1,spmv_grid=768"}; do
  sets=""; for kv in ${knobs//,/ }; do sets="$sets --set $kv"; done
  timeout -k 10 200 python bench.py --values random --no-also --no-cpu-baseline --steps 30 --warmup 5 $sets > gpurun_out/abo.json 2> gpurun_out/abo.err || { tail -3 gpurun_out/abo.err; exit 1; }
  python - <<PY >> gpurun_out/ab_offsets.txt
import json
d = json.load(open("gpurun_out/abo.json"))
r = d["roofline"]
print("%-36s %7.1f it/s  spmv %7.1f us  frac %.3f  (%s, %d launches)" % ("$knobs", d["value"], r["avg_launch_us"], r["frac"], r["stream"], r["launches"]))
PY
done
cat gpurun_out/ab_offsets.txt
