"""Probe: can a streaming kernel (axpy, HBM-bound) hide under the compressed-stream SpMV (latency / vector-memory
bound) when both run on their own streams?  Prints sequential vs concurrent wall time per (SpMV + axpy) pair."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sprsolve_amd as sa  # noqa: E402
from sprsolve_amd import _lib, gen_torch  # noqa: E402

dev = torch.device("cuda", 0)
ctxA = sa.default_ctx(0)
ctxB = sa.Context(0)
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev); n = 50_000_000
A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctxA)
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
a = torch.rand(n, dtype=torch.float64, device=dev); b = torch.rand(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
L = _lib.lib()
pa, pb = C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr())
sx = getattr(L, "sprs_mul_vec_dev_d")


def spmv():
    sx(A.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()))


def axpy(ctx):
    L.sprs_axpy_d(ctx.h, n, 1e-3, pa, pb)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ctxA.sync(); ctxB.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctxA.sync(); ctxB.sync()
    return (time.perf_counter() - t0) / reps * 1e6


out = dict(spmv_us=timed(spmv), axpy_us=timed(lambda: axpy(ctxA)),
           sequential_us=timed(lambda: (spmv(), axpy(ctxA))),
           concurrent_us=timed(lambda: (spmv(), axpy(ctxB))))
print(json.dumps(out))
