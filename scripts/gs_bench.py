"""Gauss-Seidel sweep rate: level-scheduled HIP sweeps (graph replay on/off) vs the oracle's serial sweep.
usage: python scripts/gs_bench.py [nx ny nz] [sweeps]      (run on the GPU box)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sprsolve_amd as sa  # noqa: E402
from sprsolve_amd import gen  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    nx, ny, nz = (a + [200, 200, 200])[:3] if len(a) >= 3 else (200, 200, 200)
    sweeps = a[3] if len(a) > 3 else 20
    indptr, indices, data, rhs = gen.poisson3d(nx, ny, nz)
    n = indptr.size - 1
    ctx = sa.default_ctx(0)
    A = sa.HipCsr.new((n, n), indptr, indices, data)
    gs = sa.GaussSeidel.new(A)
    d_rhs = sa.DevVec.from_numpy(rhs)
    out = dict(grid=[nx, ny, nz], rows=n, nnz=int(indptr[-1]), levels=gs.levels, sweeps=sweeps)
    for graph in (1, 0):
        ctx.set("gs_graph", graph)
        for rep in range(2):        # first pass captures the graphs
            d_x = sa.DevVec.from_numpy(np.zeros(n))
            ctx.sync(); t0 = time.perf_counter()
            try:
                gs.solve(d_rhs, d_x, sweeps, 0.0)
            except sa.error.InsufficientIterNum:
                pass
            ctx.sync(); dt = time.perf_counter() - t0
        out["hip_ms_per_sweep_graph%d" % graph] = 1e3 * dt / sweeps
        x_gpu = d_x.to_numpy()
    if os.environ.get("GS_CPU", "1") == "1":
        from oracle import oracle as orc
        orc.build()
        k = min(sweeps, 3)
        t0 = time.perf_counter()
        ref = orc.gauss_seidel(indptr, indices, data, rhs, np.zeros(n), k, 0.0)
        out["cpu_ms_per_sweep_incl_residual"] = 1e3 * (time.perf_counter() - t0) / k
        d_x = sa.DevVec.from_numpy(np.zeros(n))
        try:
            gs.solve(d_rhs, d_x, k, 0.0)
        except sa.error.InsufficientIterNum:
            pass
        out["bit_identical_after_%d_sweeps" % k] = bool(np.array_equal(d_x.to_numpy(), ref.x))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
