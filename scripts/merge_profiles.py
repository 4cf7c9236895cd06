"""Copy what scripts/profile_all.sh and scripts/pmc_l2_ta.sh left under gpurun_out/ into profiles/ (round 4 names) and merge the
per-workload PMC summaries into profiles/r04_pmc_summary.json — refusing summaries taken with other SpMV sources than the tree's.
  usage: python scripts/merge_profiles.py"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dg = bench.csrc_digest()
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
P = os.path.join(ROOT, "profiles", "r04_pmc_summary.json")
d = json.load(open(P)) if os.path.exists(P) else {}
for tag in ("cfg5_pair", "cfg5_csr", "cfg5_random", "cfg3_banded", "cfg4_complex", "cfg2_poisson2d"):
    e = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_summary_%s.json" % tag)))[tag]
    assert e.get("csrc_digest") == dg, (tag, e.get("csrc_digest"), dg)
    e["commit"] = commit
    d[tag] = e
    print(tag, "traffic %.3f GB per in-solve SpMV launch" % (e["spmv_in_solve"]["traffic_bytes"] / 1e9))
    shutil.copy(os.path.join(ROOT, "gpurun_out", "kernel_stats_%s.csv" % tag), os.path.join(ROOT, "profiles", "r04_kernel_stats_%s.csv" % tag))
    shutil.copy(os.path.join(ROOT, "gpurun_out", "prof_%s_stats.json" % tag), os.path.join(ROOT, "profiles", "r04_bench_%s_under_rocprof.json" % tag))
json.dump(d, open(P, "w"), indent=1)
for a, b in (("pmc_fused_vec.json", "r04_pmc_fused_vec.json"), ("pmc_fused_vec.txt", "r04_pmc_fused_vec_raw.txt")):
    if os.path.exists(os.path.join(ROOT, "gpurun_out", a)):
        shutil.copy(os.path.join(ROOT, "gpurun_out", a), os.path.join(ROOT, "profiles", b))
