"""complex SpMV alone, HIP-event timed: python3 scripts/cp_probe.py"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import sprsolve_amd as sa
from sprsolve_amd import gen

def run(nx, ny, knobs):
    ctx = sa.default_ctx(0)
    ip, ix, dv, rhs, _ = gen.complex_symmetric_grid(nx, ny)
    n = nx * ny
    for k, v in knobs.items(): ctx.set(k, v)
    A = sa.HipCsr.new((n, n), ip, ix, dv, ctx=ctx)
    x = torch.from_numpy(rhs).cuda(); y = torch.empty_like(x)
    for _ in range(20): A.mul_vec_unchecked(x, y)
    ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(100): A.mul_vec_unchecked(x, y)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 100 * 1e6
    print("%dx%d" % (nx, ny), knobs, "format", A.stream_format(), "blocks", A.wide_blocks(), "%.2f us/launch" % dt, flush=True)
    for k, v in knobs.items(): ctx.set(k, -1)

for nx, ny in ((500, 1000), (13500, 37)):
    run(nx, ny, {"spmv_dict": 0})
    run(nx, ny, {"spmv_dict": 1})
    run(nx, ny, {"spmv_dict": 2})
    run(nx, ny, {"spmv_dict": 2, "spmv_uniform": 0})
