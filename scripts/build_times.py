"""Compile every translation unit of libsprsolve_hip.so in parallel (like `make -jN`) and report the per-file and total wall
time — profiles/r04_tuning.md "build cost".   usage: python scripts/build_times.py [jobs]"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sprsolve_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-function".split()


def one(src, out):
    t0 = time.time()
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", out])
    return os.path.basename(src), time.time() - t0


if __name__ == "__main__":
    jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    tmp = tempfile.mkdtemp()
    t0 = time.time()
    with cf.ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(lambda s: one(s, os.path.join(tmp, os.path.basename(s) + ".o")), srcs))
    objs = [os.path.join(tmp, os.path.basename(s) + ".o") for s in srcs]
    t1 = time.time()
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(tmp, "lib.so")] + objs + ["-ldl"])
    for name, dt in sorted(res, key=lambda r: -r[1]):
        print("%-22s %6.1f s" % (name, dt))
    print("compile wall (-j%d) %.1f s, link %.1f s, total %.1f s" % (jobs, t1 - t0, time.time() - t1, time.time() - t0))
