"""Pretty-print the one-line JSON of bench.py."""
import json, sys
for f in sys.argv[1:]:
    b = json.loads([l for l in open(f) if l.startswith("{")][-1])
    r = b["roofline"]
    print("%s: %.1f %s  %.3f ms/step  | SpMV %.0f GB/s (%.1f%% of %.0f) %.0f us/launch | eff %.0f GB/s" % (
        f, b["value"], b["unit"], b["ms_per_step"], r["achieved"], 100 * r["frac"], r["peak"], r.get("avg_launch_us", 0),
        b.get("effective_GBs_reference_oplist", 0)))
    for k, v in b.get("also", {}).items():
        print("   also %s: %.0f it/s  %.1f us/step  SpMV in-solve %.1f us (%.0f GB/s), back-to-back %.1f us (%.0f GB/s)" % (
            k, v["value"], v["ms_per_step"] * 1e3, v["spmv_us_in_solve"], v["spmv_GBs_in_solve"], v["spmv_us_back_to_back"], v["spmv_GBs_back_to_back"]))
    if "cpu_baseline" in b:
        print("   cpu_baseline: %.3f it/s on %d cores" % (b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"]))
