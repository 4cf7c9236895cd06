"""PMC probe for the dictionary-compressed SpMV (run under rocprofv3 --pmc <counter> --kernel-trace).
Variants launched 3x each, in this order; parse with scripts/pmc_parse.py.
  usage: python3 scripts/pmc_dict.py [grid]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sprsolve_amd as sa  # noqa: E402
from sprsolve_amd import gen_torch  # noqa: E402

dev = torch.device("cuda", 0); ctx = sa.default_ctx(0)
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev); n = 50_000_000
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
VARIANTS = [("csr", dict(spmv_dict=0, spmv_grid=1024)),
            ("offsets", dict(spmv_dict=1, spmv_grid=1024)),
            ("dict", dict(spmv_dict=2, spmv_grid=grid)),
            ("dict_chunk", dict(spmv_dict=2, spmv_grid=grid, xcd_chunk=1)),
            ("dict_period", dict(spmv_dict=2, spmv_grid=grid, spmv_strip=1)),
            ("dict_strip8k", dict(spmv_dict=2, spmv_grid=grid, spmv_strip=8192))]
only = os.environ.get("PMC_VARIANTS")
if only:
    VARIANTS = [v for v in VARIANTS if v[0] in only.split(",")]
for name, knobs in VARIANTS:
    for k, v in dict(spmv_dict=-1, spmv_grid=-1, xcd_chunk=-1, spmv_strip=0).items():
        ctx.set(k, v)
    for k, v in knobs.items():
        ctx.set(k, v)
    A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctx)
    for _ in range(3):
        A.mul_vec_unchecked(x, y)
    ctx.sync()
    A.close()
print("variants:", ",".join(v[0] for v in VARIANTS))
