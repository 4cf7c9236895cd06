"""A short cfg-5 BiCGStab solve for a rocprofv3 --pmc pass over the fused vector kernels and the chain SpMV kernels.
  usage: python3 scripts/pmc_vec.py [iterations] [KEY=VALUE knobs ...]      (spmv_fuse=0: the five-launch iteration with K1 / K3)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sprsolve_amd as sa  # noqa: E402
from sprsolve_amd import gen_torch  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0); ctx = sa.default_ctx(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set(k, int(v))
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev); n = 50_000_000
A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctx)
s = sa.BiCGStab.new(A, n)
x = torch.zeros(n, dtype=torch.float64, device=dev)
try:
    s.solve(rhs, x, iters, 0.0)
except sa.error.InsufficientIterNum:
    pass
ctx.sync()
