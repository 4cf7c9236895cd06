"""Summary of a bench.py line (round-3 fields).  usage: python scripts/show_r3.py file.json [...]"""
import json
import sys


def roof(r):
    return "%7.1f us  %5.0f GB/s  frac %.3f%s" % (r["avg_launch_us"], r["achieved"], r["frac"],
                                                ("  (sv8d %.3f)" % r["frac_survey_8d"]) if "frac_survey_8d" in r else "")


for f in sys.argv[1:]:
    d = json.load(open(f))
    print("%s: %.1f it/s  %.1f us/it  create %.1f ms  [%s]" % (f, d["value"], d["ms_per_step"] * 1e3, d.get("create_ms", -1), d["timing"]["ms_per_step_from"][:40]))
    print("   roofline %-13s %s" % (d["roofline"]["stream"], roof(d["roofline"])))
    if "roofline_plain_csr" in d:
        print("   plain csr leg          %s   %.1f it/s" % (roof(d["roofline_plain_csr"]), d["also"]["cfg5_plain_csr_stream"]["value"]))
    a = d.get("also", {})
    if "cfg5_random_values" in a:
        r = a["cfg5_random_values"]
        print("   random values leg      %s   %.1f it/s  create %.1f ms" % (roof(r["roofline"]), r["value"], r["create_ms"]))
    if "cfg2_poisson2d_1M_bicgstab_jacobi" in a:
        c = a["cfg2_poisson2d_1M_bicgstab_jacobi"]
        print("   cfg2                   %.0f it/s  spmv %.1f us  create %.1f ms" % (c["value"], c["spmv_us_in_solve"], c["create_ms"]))
    if "cpu_baseline" in d:
        print("   cpu baseline %.2f it/s at %d threads" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
    if "allreduce_us" in d:
        print("   dist: rccl_ranks %s allreduce %.2f us halo %s" % (d.get("rccl_ranks"), d["allreduce_us"], d.get("halo_bytes")))
