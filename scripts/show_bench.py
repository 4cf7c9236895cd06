"""Pretty-print the one-line JSON of bench.py."""
import json, sys
for f in sys.argv[1:]:
    b = json.loads([l for l in open(f) if l.startswith("{")][-1])
    r = b["roofline"]
    print("%s: %.1f %s  %.3f ms/step  | SpMV %.0f GB/s (%.1f%% of %.0f) %.0f us/launch | eff %.0f GB/s" % (
        f, b["value"], b["unit"], b["ms_per_step"], r["achieved"], 100 * r["frac"], r["peak"], r.get("avg_launch_us", 0),
        b.get("effective_GBs_reference_oplist", 0)))
    if "stream" in r:
        print("   stream %s: %.2f GB/launch in the format -> %.0f GB/s (%.1f%% of peak); PMC traffic %s" % (
            r["stream"], r["format_bytes_per_launch"] / 1e9, r["format_GBs"], 100 * r["format_frac_of_hbm_peak"],
            ("%.2f GB" % (r["traffic"] / 1e9)) if r.get("traffic") else "n/a"))
    for k, v in b.get("also", {}).items():
        if "spmv_us_back_to_back" not in v:
            print("   also %s: %.1f it/s  %.3f ms/step  SpMV %.0f us (%.0f GB/s = %.1f%% of peak)" % (
                k, v["value"], v["ms_per_step"], v["spmv_us_in_solve"], v["spmv_GBs"], 100 * v["spmv_frac_of_hbm_peak"]))
            continue
        print("   also %s: %.0f it/s  %.1f us/step  SpMV in-solve %.1f us (%.0f GB/s), back-to-back %.1f us (%.0f GB/s)" % (
            k, v["value"], v["ms_per_step"] * 1e3, v["spmv_us_in_solve"], v["spmv_GBs_in_solve"], v["spmv_us_back_to_back"], v["spmv_GBs_back_to_back"]))
    if "cpu_baseline" in b:
        print("   cpu_baseline: %.3f it/s on %d cores" % (b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"]))
