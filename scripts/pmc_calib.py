"""PMC calibration probe (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): SpMV launches on
matrices whose HBM traffic is known by construction, to calibrate FETCH_SIZE for the access
widths the SpMV kernel uses (MI355X_MICROARCH.md §HBM: only 16-B/lane streams are calibrated).
  A: 7 nnz/row, every col_idx = 0        -> stream only (x traffic ~ 0)
  B: 7 nnz/row, every col_idx = row      -> stream + x read exactly once
  C: the cfg-5 7-point matrix            -> the kernel being measured
Launch order is A, B, C, each REPS times, after one warm-up each (so dispatch order identifies them)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sprsolve_amd as sa
from sprsolve_amd import gen_torch
dev = torch.device("cuda", 0)
ctx = sa.default_ctx(0)
nx, ny, nz = 500, 500, 200
n = nx * ny * nz
REPS = 3
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
ipA = (torch.arange(n + 1, dtype=torch.int64, device=dev) * 7).to(torch.int32)
val = torch.rand(7 * n, dtype=torch.float64, device=dev)
for name in ("A", "B"):
    if name == "A":
        ci = torch.zeros(7 * n, dtype=torch.int32, device=dev)
    else:
        ci = torch.arange(n, dtype=torch.int32, device=dev).repeat_interleave(7)
    M = sa.HipCsr.from_device((n, n), 7 * n, ipA, ci, val, adopt=True, ctx=ctx)
    for _ in range(REPS + 1):
        M.mul_vec_unchecked(x, y)
    ctx.sync(); M.close(); del ci
ip, ix, dv, rhs = gen_torch.poisson3d(nx, ny, nz, device=dev)
C = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctx)
for _ in range(REPS + 1):
    C.mul_vec_unchecked(x, y)
ctx.sync()
print("n", n, "nnzA", 7 * n, "nnzC", int(ip[-1].item()))
