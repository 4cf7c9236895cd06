"""Pretty-print the one-line JSON of bench.py."""
import json, sys


def roof(r, pre="   "):
    print("%s%s: %.2f GB/launch in %.0f us -> %.0f GB/s = %.1f%% of %.0f%s%s" % (
        pre, r["stream"], r["format_bytes_per_launch"] / 1e9, r["avg_launch_us"], r["achieved"], 100 * r["frac"], r["peak"],
        ("; CSR-equivalent %.0f GB/s" % r["csr_equivalent_GBs"]) if "csr_equivalent_GBs" in r else "",
        ("; PMC traffic %.2f GB" % (r["traffic"] / 1e9)) if r.get("traffic") else ""))


for f in sys.argv[1:]:
    b = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print("%s: %.1f %s  %.4f ms/step (n_gpus %d)" % (f, b["value"], b["unit"], b["ms_per_step"], b["n_gpus"]))
    if b.get("roofline"):
        roof(b["roofline"], "   roofline ")
    if b.get("roofline_plain_csr"):
        roof(b["roofline_plain_csr"], "   roofline_plain_csr ")
    for k, v in b.get("also", {}).items():
        if not isinstance(v, dict) or "value" not in v:
            continue
        print("   also %s: %.1f it/s  %.4f ms/step" % (k, v["value"], v.get("ms_per_step", 1e3 / v["value"] if v["value"] else 0.0)) + (
            "  SpMV in-solve %.1f us, back-to-back %.1f us" % (v["spmv_us_in_solve"], v["spmv_us_back_to_back"]) if "spmv_us_back_to_back" in v else ""))
        if "roofline" in v:
            roof(v["roofline"], "        ")
    if "cpu_baseline" in b:
        c = b["cpu_baseline"]
        print("   cpu_baseline: " + ", ".join("%s %.3f it/s" % (k, c[k]["value"]) for k in c if isinstance(c[k], dict)) + " (%s)" % c.get("cpu_model"))
    for k in ("rccl_ranks", "halo_bytes"):
        if k in b:
            print("   %s: %s" % (k, b[k]))
