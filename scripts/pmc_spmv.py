"""Stand-alone launches of ONE of the three f64 SpMV kernels on cfg 5 for a rocprofv3 --pmc pass.
  usage: python3 scripts/pmc_spmv.py <pair|offsets|csr> [reps] [KEY=VALUE knobs ...]
pair: constant-coefficient cfg 5, pair codes (spmv_pair2_kernel); offsets: the same pattern with random values, offset
codes (spmv_dict_kernel<double, PAIR=false>); csr: the plain stream (spmv_kernel<double>)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sprsolve_amd as sa  # noqa: E402
from sprsolve_amd import gen_torch  # noqa: E402

which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0); ctx = sa.default_ctx(0)
ctx.set("spmv_dict", {"pair": -1, "offsets": -1, "csr": 0}[which])
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ctx.set(k, int(v))
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev, values="random" if which == "offsets" else "poisson"); n = 50_000_000
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctx)
assert A.stream_format()[0] == {"pair": 2, "offsets": 1, "csr": 0}[which], A.stream_format()
for _ in range(reps):
    A.mul_vec_unchecked(x, y)
ctx.sync()
