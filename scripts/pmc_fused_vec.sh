# TA / TCP / SQ counters of the vector kernels of the cfg-5 BiCGStab iteration (fused_kernel<BicgK1 / K3 / K5>, five-launch
# iteration: spmv_fuse=0) and of the chain SpMV kernels, plain and with the fused input (round 4).
# One rocprofv3 --pmc pass per small counter set; the program directly after `--`.
#   usage: bash scripts/pmc_fused_vec.sh        -> gpurun_out/pmc_fused_vec.json (+ .txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/pmc_fused_vec.txt
for fuse in 0 1; do
  i=0
  for set in "TA_BUSY_avr TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
             "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    d=gpurun_out/pmcvec_${fuse}_$i
    timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 scripts/pmc_vec.py 12 spmv_fuse=$fuse > $d.log 2>&1 \
      || { echo "fuse$fuse pass $i ($set) FAILED: $(grep -m1 -i 'error code\|exceeds' $d.log | cut -c1-160)" | tee -a gpurun_out/pmc_fused_vec.txt; rm -rf $d; continue; }
    python3 - "$fuse" "$d" <<'PY' >> gpurun_out/pmc_fused_vec.txt
import csv, glob, collections, statistics, sys
fuse, d = sys.argv[1], sys.argv[2]
def tag(n):
    for k in ("BicgK1", "BicgK3", "BicgK5"):
        if "fused_kernel" in n and k in n: return k
    if "spmv_chain_kernel" in n:
        a = n[n.index("spmv_chain_kernel<") + 18:]
        return "chain_" + ("plain_dot" + a.split(",")[2].strip() if a.startswith("0") else ("K3inK4" if a.startswith("2") else "K1inK2"))
    return None
vals = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t = tag(r["Kernel_Name"])
        if t: vals[(t, r["Counter_Name"])].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t = tag(r["Kernel_Name"])
        if t: dur[t].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (t, c) in sorted(vals):
    print("fuse%s|%s|%s|%.6g|%d|%.1f" % (fuse, t, c, statistics.median(vals[(t, c)]), len(vals[(t, c)]), statistics.median(dur[t]) if dur[t] else -1))
PY
    rm -rf $d
  done
done
python3 - <<'PY'
import json
out = {}
for ln in open("gpurun_out/pmc_fused_vec.txt"):
    p = ln.strip().split("|")
    if len(p) != 6:
        out.setdefault("failed_passes", []).append(ln.strip()); continue
    fuse, kname, ctr, val, n, us = p
    e = out.setdefault(fuse + ":" + kname, dict(counters={}, launches=int(n)))
    e["counters"][ctr] = dict(median_per_launch=float(val), kernel_us_under_this_pass=float(us))
for k, e in out.items():
    if k == "failed_passes": continue
    c = {a: v["median_per_launch"] for a, v in e["counters"].items()}
    d = {}
    if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0: d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c: d["fabric_GB"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e9
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
        for a in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
            if a in c: d[a + "_over_WAVE_CYCLES"] = c[a] / c["SQ_WAVE_CYCLES"]
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc > 0:
        d["launch_cycles"] = cyc
        for name, key in (("ta_busy_fraction", "TA_TA_BUSY_sum"), ("ta_addr_stalled_by_tcp_fraction", "TA_ADDR_STALLED_BY_TC_CYCLES_sum"),
                          ("ta_data_stalled_by_tcp_fraction", "TA_DATA_STALLED_BY_TC_CYCLES_sum"), ("tcp_pending_stall_fraction", "TCP_PENDING_STALL_CYCLES_sum"),
                          ("tcp_ta_data_stall_fraction", "TCP_TCP_TA_DATA_STALL_CYCLES_sum")):
            if key in c: d[name] = c[key] / (cyc * 256)
    us = [v["kernel_us_under_this_pass"] for v in e["counters"].values()]
    d["kernel_us_median_over_passes"] = sorted(us)[len(us) // 2]
    e["derived"] = d
json.dump(out, open("gpurun_out/pmc_fused_vec.json", "w"), indent=1)
print(json.dumps({k: v.get("derived", v) for k, v in out.items()}, indent=1))
PY
