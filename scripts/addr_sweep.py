"""Does the back-to-back time of the offset-code tile SpMV depend on WHERE its arrays sit?  The same matrix with val / x / y at
different addresses (padding allocations in between), one process.   usage: python3 scripts/addr_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sprsolve_amd as sa  # noqa: E402
from sprsolve_amd import gen_torch  # noqa: E402

dev = torch.device("cuda", 0); ctx = sa.default_ctx(0)
n = 50_000_000
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev, values="random")
nnz = int(ip[-1].item())
keep = []
for trial, pad in enumerate([0, 4096, 65536 + 256, 1 << 20, (2 << 20) + 4096, (64 << 20) + 8192, (1 << 30) + 4096, 0, 12345 * 16]):
    if pad:
        keep.append(torch.empty(pad, dtype=torch.uint8, device=dev))
    v2 = dv.clone(); x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
    A = sa.HipCsr.from_device((n, n), nnz, ip, ix, v2, adopt=True, ctx=ctx)
    us = [A.time_mul_vec(x, y, 20) * 1e3 for _ in range(3)]
    print("pad %11d  val @ %#x (mod 2 MiB %#8x)  x @ %#x  y @ %#x   %s us" % (pad, v2.data_ptr(), v2.data_ptr() % (2 << 20), x.data_ptr(), y.data_ptr(),
                                                                                " ".join("%.1f" % u for u in us)), flush=True)
    del A, v2, x, y
    torch.cuda.empty_cache() if trial % 2 else None
