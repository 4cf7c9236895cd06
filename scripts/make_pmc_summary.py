"""Build the per-kernel HBM-side traffic summary from two rocprofv3 --pmc passes of bench.py
(FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) plus a kernel-trace.

  usage: python scripts/make_pmc_summary.py <fetch_dir> <write_dir> <key> [summary.json]

Adds / replaces the entry <key> of profiles/r03_pmc_summary.json with, per kernel: launches, median
FETCH_SIZE / WRITE_SIZE (KiB per dispatch), average duration, and for the dominant SpMV kernel the
corrected traffic: 2 * FETCH_SIZE (gfx950 counts 128-B fabric requests at 64 B; calibrated 0.510 / 0.509,
see "calibration" in the same file) + WRITE_SIZE."""
import csv
import glob
import json
import os
import statistics
import sys


def read_counter(d, name):
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == name:
                    per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return per


def read_durations(d):
    per = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                per.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per


def short(name):
    """Kernel name without its argument list (keeps template arguments and `(anonymous namespace)`)."""
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not name.startswith("(anonymous namespace)", i):
            return name[:i].strip()
    return name.strip()


def main():
    fetch_dir, write_dir, key = sys.argv[1:4]
    out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                     "profiles", "r03_pmc_summary.json")
    fe, wr, du = read_counter(fetch_dir, "FETCH_SIZE"), read_counter(write_dir, "WRITE_SIZE"), read_durations(fetch_dir)
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        if "sprs" not in k:
            continue
        kernels[short(k)] = dict(launches=len(fe.get(k, [])),
                                 FETCH_SIZE_KiB_median=statistics.median(fe[k]) if k in fe else None,
                                 WRITE_SIZE_KiB_median=statistics.median(wr[k]) if k in wr else None,
                                 avg_us_under_pmc=sum(du[k]) / len(du[k]) if k in du else None)
    spmv = {k: v for k, v in kernels.items() if "spmv" in k and v["launches"] > 20}
    entry = dict(kernels=kernels)
    if spmv:
        tot = sum(v["launches"] for v in spmv.values())
        f = sum(v["FETCH_SIZE_KiB_median"] * v["launches"] for v in spmv.values()) / tot * 1024
        w = sum(v["WRITE_SIZE_KiB_median"] * v["launches"] for v in spmv.values()) / tot * 1024
        entry["spmv_in_solve"] = dict(kernels=sorted(spmv), launches=tot, fabric_read_bytes=2 * f, write_bytes=w,
                                      traffic_bytes=2 * f + w,
                                      note="launch-weighted mean over the in-solve SpMV launches; 2*FETCH_SIZE + WRITE_SIZE; "
                                           "L2->fabric bytes, Infinity-Cache hits included (upper bound on HBM bytes)")
    summary = json.load(open(out_path)) if os.path.exists(out_path) else {}
    summary.setdefault("units", "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (rocprofv3 --pmc, one counter per pass); traffic_bytes = "
                       "2 * FETCH_SIZE + WRITE_SIZE in bytes (gfx950 tallies 128-B fabric read requests at 64 B: MI355X_MICROARCH.md "
                       "§HBM; calibration in r01_pmc_summary.json)")
    import subprocess
    try:
        entry["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(out_path),
                                                  stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:       # the GPU box has no .git: the commit is filled in when the summary is copied into profiles/
        entry["commit"] = None
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_digest          # what bench.py compares with to flag a stale `roofline.traffic`
    entry["csrc_digest"] = csrc_digest()
    summary[key] = entry
    json.dump(summary, open(out_path, "w"), indent=1)
    print(json.dumps(entry.get("spmv_in_solve", {}), indent=1))


if __name__ == "__main__":
    main()
