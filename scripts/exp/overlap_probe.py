"""Does a streaming pass on a second stream overlap with the pair-code SpMV?  (experiment, not part of the product)
The SpMV is bound by the CUs' vector-memory pipe and leaves ~half of the HBM bandwidth idle; BiCGStab's x update
(x += alpha p + omega s) is off the recurrence's critical path.  Times: SpMV alone, two axpys alone, both at once."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
import sprsolve_amd as sa                    # noqa: E402
from sprsolve_amd import _lib, gen_torch     # noqa: E402
from sprsolve_amd.device import dev_ptr      # noqa: E402

dev = torch.device("cuda", 0)
ctx = sa.default_ctx(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ctx.set(k, int(v))
nx, ny, nz = 500, 500, 200
ip, ix, dv, rhs = gen_torch.poisson3d(nx, ny, nz, device=dev)
n, nnz = nx * ny * nz, int(ip[-1].item())
A = sa.HipCsr.from_device((n, n), nnz, ip, ix, dv, adopt=True, ctx=ctx)
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
xx = torch.zeros_like(x); p = torch.rand_like(x); s = torch.rand_like(x)
torch.cuda.synchronize()
L = _lib.lib()
side = torch.cuda.Stream(device=dev)
N = 30


def spmv():
    st = L.sprs_mul_vec_dev_d(A.h, dev_ptr(x), dev_ptr(y))
    assert st == 0


def axpys():
    with torch.cuda.stream(side):
        xx.add_(p, alpha=0.5)
        xx.add_(s, alpha=0.25)


def timed(fn):
    for _ in range(3):
        fn()
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    ctx.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


t_spmv = timed(spmv)
t_ax = timed(axpys)
t_both = timed(lambda: (spmv(), axpys()))
print("SpMV alone %.1f us, two axpys alone %.1f us, serial sum %.1f us, both at once %.1f us per round" % (t_spmv, t_ax, t_spmv + t_ax, t_both))
