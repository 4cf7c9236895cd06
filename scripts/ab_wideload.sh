# A/B of the 16-byte stream loads (knob spmv_wideload) on every leg they touch, alternating, same box
#   usage: bash scripts/ab_wideload.sh -> gpurun_out/ab_wideload.txt
rm -f gpurun_out/ab_wideload.txt
run() {   # label, bench flags...
  label=$1; shift
  timeout -k 10 200 python bench.py --no-also --no-cpu-baseline "$@" > gpurun_out/abw.json 2> gpurun_out/abw.err || { echo "$label FAILED"; tail -3 gpurun_out/abw.err; return; }
  python - "$label" <<'PY' >> gpurun_out/ab_wideload.txt
import json, sys
d = json.load(open("gpurun_out/abw.json"))
r = d["roofline"]
print("%-52s %9.1f it/s  spmv %7.1f us  frac %.3f  (%s)" % (sys.argv[1], d["value"], r["avg_launch_us"], r["frac"], r["stream"]))
PY
}
for rep in 1 2; do
  for wl in 0 1; do
    run "cfg5 csr wideload=$wl" --stream csr --steps 40 --warmup 5 --set spmv_wideload=$wl
    run "cfg5 csr wideload=$wl grid=1024" --stream csr --steps 40 --warmup 5 --set spmv_wideload=$wl --set spmv_grid=1024
    run "cfg2 csr wideload=$wl" --workload poisson2d --stream csr --steps 500 --warmup 50 --set spmv_wideload=$wl
    run "cfg3 banded (offset codes) wideload=$wl" --workload banded --steps 500 --warmup 50 --set spmv_wideload=$wl
    run "cfg3 banded csr wideload=$wl" --workload banded --stream csr --steps 500 --warmup 50 --set spmv_wideload=$wl
    run "slab 500x500x25 csr wideload=$wl" --grid 500x500x25 --stream csr --steps 200 --warmup 20 --set spmv_wideload=$wl
    run "slab 500x500x25 random wideload=$wl" --grid 500x500x25 --values random --steps 40 --warmup 5 --set spmv_wideload=$wl
  done
done
cat gpurun_out/ab_wideload.txt
