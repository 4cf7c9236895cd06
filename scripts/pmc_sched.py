"""PMC probe: L2->fabric read traffic of the SpMV under the row-block schedules (run under rocprofv3 --pmc FETCH_SIZE).
Launch order: strip 0 (natural) x3, strip 1 (XCD-period) x3, strip 16384 x3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sprsolve_amd as sa
from sprsolve_amd import gen_torch
dev = torch.device("cuda", 0); ctx = sa.default_ctx(0)
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev); n = 50_000_000
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
for strip in (0, 1, 16384):
    ctx.set("spmv_strip", strip)
    A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctx)
    for _ in range(3):
        A.mul_vec_unchecked(x, y)
    A.close()
print("done")
