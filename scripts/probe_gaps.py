"""cfg 5 iteration time with the SpMV profile events on and off.  python3 scripts/probe_gaps.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import sprsolve_amd as sa
from sprsolve_amd import gen_torch
import bench

ctx = sa.default_ctx(0)
dev = torch.device("cuda", 0)
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev)
n = 500 * 500 * 200
A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctx)
s = sa.BiCGStab.new(A, n)
x = torch.zeros(n, dtype=torch.float64, device=dev)
bench.run_fixed_iterations(s, None, rhs, x, 10)
for rep in range(3):
    for prof in (False, True):
        s.set_profile(prof)
        x.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_fixed_iterations(s, None, rhs, x, 150)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("cfg5 profile %-5s %8.2f us/iteration (set-up included)  %.1f it/s" % (prof, dt / 150 * 1e6, 150 / dt), flush=True)
