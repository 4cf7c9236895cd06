"""PMC probe: 3 SpMV launches (cfg 5, default settings) + 3 stand-alone axpy launches for comparison."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sprsolve_amd as sa
from sprsolve_amd import gen_torch, _lib
dev = torch.device("cuda", 0); ctx = sa.default_ctx(0)
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev); n = 50_000_000
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctx)
for _ in range(3):
    A.mul_vec_unchecked(x, y)
for _ in range(3):
    _lib.lib().sprs_axpy_d(ctx.h, n, 0.5, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()))
ctx.sync()
