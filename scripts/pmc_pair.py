"""Stand-alone launches of the headline SpMV (f64 pair codes, two rows per lane) on cfg 5 for a rocprofv3 --pmc pass.
  usage: python3 scripts/pmc_pair.py [reps] [KEY=VALUE knobs ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sprsolve_amd as sa  # noqa: E402
from sprsolve_amd import gen_torch  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0); ctx = sa.default_ctx(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set(k, int(v))
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev); n = 50_000_000
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctx)
assert A.stream_format()[0] == 2
for _ in range(reps):
    A.mul_vec_unchecked(x, y)
ctx.sync()
