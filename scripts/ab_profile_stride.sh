# same box: the default bench line with HIP events around every SpMV launch / one pair in four / one pair in sixteen
for st in 1 4 1 4 16; do
  python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline --profile-stride $st > gpurun_out/ps_$st.json 2> gpurun_out/ps_$st.err || { tail -3 gpurun_out/ps_$st.err; exit 1; }
  python - $st <<'PY'
import json, sys
j = json.loads(open("gpurun_out/ps_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print("stride", sys.argv[1], "%.1f it/s  %.4f ms/step" % (j["value"], j["ms_per_step"]), "| SpMV launches timed %d of %d, avg %.1f us, frac %.3f" % (r["launches"], r.get("launches_in_region", -1), r["avg_launch_us"], r["frac"]))
PY
done
