"""GPU tuning probe (not part of the product): SpMV time vs grid / xcd_chunk, and the
streaming-kernel ceiling measured with the library's own axpy."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sprsolve_amd as sa
from sprsolve_amd import gen_torch, _lib
import ctypes as C

dev = torch.device("cuda", 0)
ctx = sa.default_ctx(0)
which = sys.argv[1] if len(sys.argv) > 1 else "3d"
if which == "3d":
    nx, ny, nz = 500, 500, int(sys.argv[2]) if len(sys.argv) > 2 else 200
    ip, ix, dv, rhs = gen_torch.poisson3d(nx, ny, nz, device=dev)
    n = nx * ny * nz
else:
    ip, ix, dv, rhs, diag = gen_torch.grid_laplacian_dirichlet(1000, 1000, device=dev)
    n = 10**6
nnz = int(ip[-1].item())
x = torch.rand(n, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
yref = None
B = nnz * 12 + (n + 1) * 4 + 2 * n * 8
mats = {}
for strip in (0, 1):
    ctx.set("spmv_strip", strip)
    mats[strip] = sa.HipCsr.from_device((n, n), nnz, ip, ix, dv, adopt=True, ctx=ctx)
    mats[strip].mul_vec_unchecked(x, y)
    if yref is None:
        yref = y.clone()
    assert torch.equal(y, yref), "schedule changed the result"
for rnd in range(2):
    for strip, A in mats.items():
        for grid in (1024,):
            for chunk in (-1,):
                for nt in (0, 1):
                    ctx.set("spmv_grid", grid); ctx.set("xcd_chunk", chunk); ctx.set("spmv_nt", nt)
                    A.time_mul_vec(x, y, reps=3)
                    ms = A.time_mul_vec(x, y, reps=20)
                    print("strip %5d grid %5d chunk %d nt %d : %9.1f us  %7.1f GB/s" % (strip, grid, chunk, nt, ms * 1e3, B / ms / 1e6), flush=True)
# streaming ceiling with the library's axpy (3*n*8 bytes)
L = _lib.lib()
for grid in (1024, 2048, 4096):
    ctx.set("grid", grid)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            L.sprs_axpy_d(ctx.h, n, 0.5, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()))
        ctx.sync(); dt = (time.perf_counter() - t0) / 20
    print("axpy grid %d: %.1f us  %.1f GB/s" % (grid, dt * 1e6, 3 * n * 8 / dt / 1e9), flush=True)
