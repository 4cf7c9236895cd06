"""Back-to-back time of the pair-code tile SpMV (cfg 5) as a function of the distance between x and y: both carved from one
allocation, y = x + n*8 + gap.   usage: python3 scripts/addr_sweep_xy.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sprsolve_amd as sa  # noqa: E402
from sprsolve_amd import gen_torch  # noqa: E402

dev = torch.device("cuda", 0); ctx = sa.default_ctx(0)
n = 50_000_000
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev)
A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctx)
print("tile plan", A.tile_plan(), flush=True)
big = torch.empty(2 * n + (8 << 20), dtype=torch.float64, device=dev)
big[:n] = torch.rand(n, dtype=torch.float64, device=dev)
x = big[:n]
for gap in [0, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 32768, 131072, 524288, 2097152 // 8 * 1, 262144 + 16, 4000000]:
    y = big[n + gap:2 * n + gap]          # gap in doubles
    us = [A.time_mul_vec(x, y, 20) * 1e3 for _ in range(3)]
    print("gap %9d B   (y - x) mod 2 MiB = %#8x   %s us" % (gap * 8, (y.data_ptr() - x.data_ptr()) % (2 << 20), " ".join("%.1f" % u for u in us)), flush=True)
