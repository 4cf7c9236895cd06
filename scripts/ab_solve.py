"""A/B of context knobs on the full BiCGStab iteration (cfg 5 or cfg 2), interleaved rounds in one process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sprsolve_amd as sa
from sprsolve_amd import gen_torch
dev = torch.device("cuda", 0); ctx = sa.default_ctx(0)
which = sys.argv[1] if len(sys.argv) > 1 else "3d"
variants = [dict(v.split("=") for v in arg.split(",")) for arg in sys.argv[2:]] or [{}]
if which == "3d":
    ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev); n = 50_000_000; P = None; steps = 100
else:
    ip, ix, dv, rhs, diag = gen_torch.grid_laplacian_dirichlet(1000, 1000, device=dev); n = 10**6; steps = 500
    P = sa.DiagPrecond.new(diag.cpu().numpy(), ctx=ctx)
A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=True, ctx=ctx)
s = sa.BiCGStab.new(A, n)
x = torch.zeros(n, dtype=torch.float64, device=dev)
def run():
    x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        s.precond_solve(P, rhs, x, steps, 0.0) if P is not None else s.solve(rhs, x, steps, 0.0)
    except sa.error.InsufficientIterNum:
        pass
    return (time.perf_counter() - t0) / steps
for rnd in range(3):
    for v in variants:
        for k, val in v.items():
            ctx.set(k, int(val))
        run()
        t = min(run() for _ in range(2))
        print("round %d %-40s %9.1f us/iter  %8.1f it/s" % (rnd, v, t * 1e6, 1 / t), flush=True)
