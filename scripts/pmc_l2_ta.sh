# L2 / TA / TCP counters of the three f64 SpMV kernels at cfg 5 (round 3).  One rocprofv3 --pmc pass per SMALL counter
# set: round 2 asked for six derived counters of one block per pass and rocprofv3 refused with
# "rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to collect"
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots": TCC has 4 slots) — a configuration error, not an abort of the pool.
#   usage: bash scripts/pmc_l2_ta.sh            -> gpurun_out/pmc_l2_ta.json (+ .txt)
# The program goes directly after `--` (no env / bash -c hop: the profiler's preloaded library has initialised the GPU).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/pmc_l2_ta.txt
for which in ${WHICH:-pair offsets csr}; do
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "TCC_REQ_sum TCC_TAG_STALL_sum" \
             "TA_BUSY_avr TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    d=gpurun_out/pmcl2_${which}_$i
    timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 scripts/pmc_spmv.py $which 10 > $d.log 2>&1 \
      || { echo "$which pass $i ($set) FAILED: $(grep -m1 -i 'error code\|exceeds' $d.log | cut -c1-160)" | tee -a gpurun_out/pmc_l2_ta.txt; rm -rf $d; continue; }
    python3 - "$which" "$d" <<'PY' >> gpurun_out/pmc_l2_ta.txt
import csv, glob, collections, statistics, sys
which, d = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(list); dur = []; kname = None
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmv" in r["Kernel_Name"] and "sprs" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"])); kname = r["Kernel_Name"].split("(")[0]
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmv" in r["Kernel_Name"] and "sprs" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(vals):
    print("%s|%s|%s|%.6g|%d|%.1f" % (which, kname, k, statistics.median(vals[k]), len(vals[k]), statistics.median(dur) if dur else -1))
PY
    rm -rf $d
  done
done
python3 - <<'PY'
import json
out = {}
for ln in open("gpurun_out/pmc_l2_ta.txt"):
    p = ln.strip().split("|")
    if len(p) != 6:
        out.setdefault("failed_passes", []).append(ln.strip()); continue
    which, kname, ctr, val, n, us = p
    e = out.setdefault(which, dict(kernel=kname, counters={}, launches=int(n)))
    e["counters"][ctr] = dict(median_per_launch=float(val), kernel_us_under_this_pass=float(us))
for which, e in out.items():
    if which == "failed_passes":
        continue
    c = {k: v["median_per_launch"] for k, v in e["counters"].items()}
    d = {}
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "TCC_EA0_RDREQ_sum" in c:
        d["fabric_read_bytes_128B_requests"] = c["TCC_EA0_RDREQ_sum"] * 128
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
            if k in c:
                d[k + "_over_WAVE_CYCLES"] = c[k] / c["SQ_WAVE_CYCLES"]
    # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (17.9 M "cycles" for a 1048 us launch at ~2.1 GHz): one XCD's
    # count is the launch's duration in cycles; TA_* / TCP_* "_sum" counters add up the 256 CUs' units
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc > 0:
        d["launch_cycles"] = cyc
        for name, key in (("ta_busy_fraction", "TA_TA_BUSY_sum"), ("ta_addr_stalled_by_tcp_fraction", "TA_ADDR_STALLED_BY_TC_CYCLES_sum"),
                          ("ta_data_stalled_by_tcp_fraction", "TA_DATA_STALLED_BY_TC_CYCLES_sum"),
                          ("tcp_pending_stall_fraction", "TCP_PENDING_STALL_CYCLES_sum"),
                          ("tcp_ta_data_stall_fraction", "TCP_TCP_TA_DATA_STALL_CYCLES_sum")):
            if key in c:
                d[name] = c[key] / (cyc * 256)           # mean over the 256 per-CU units
    if "TCP_TCC_READ_REQ_sum" in c:
        d["l1_miss_bytes_128B_lines"] = c["TCP_TCC_READ_REQ_sum"] * 128
    e["derived"] = d
json.dump(out, open("gpurun_out/pmc_l2_ta.json", "w"), indent=1)
print(json.dumps({k: v.get("derived", v) for k, v in out.items()}, indent=1))
PY
