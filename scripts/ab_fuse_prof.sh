# per-kernel times of the cfg-5 iteration with and without the fused SpMV input, same box (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for fuse in 0 1 0 1; do
  rm -rf gpurun_out/prof_ab_fuse
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_fuse -- python3 bench.py --no-cpu-baseline --no-also --steps 100 --warmup 5 --set spmv_fuse=$fuse $* > gpurun_out/prof_ab_fuse.json 2> gpurun_out/prof_ab_fuse.err || exit 1
  python3 - <<PY
import csv, glob, json
d = json.load(open("gpurun_out/prof_ab_fuse.json"))
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_ab_fuse/*/*kernel_stats.csv")[0])))
out = []
for r in rows[:6]:
    n = r["Name"]
    for key in ("BicgK5", "BicgK1", "BicgK3", "spmv_chain_kernel<0", "spmv_chain_kernel<2", "spmv_chain_kernel<3", "spmv_tile_kernel", "AxpyF"):
        if key in n:
            tag = key
            if key == "spmv_chain_kernel<0": tag = "chain<DOT%s>" % n[n.index("NoPro, ") + 7]
            out.append("%s %.1f" % (tag, float(r["AverageNs"]) / 1e3))
print("fuse $fuse: %.1f it/s | " % d["value"] + " | ".join(out))
PY
done
