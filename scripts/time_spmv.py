"""Back-to-back launch time of ONE f64 SpMV stream on cfg 5 (HIP events on the library's stream around `reps` launches)
under ctx knobs given as KEY=VALUE.   usage: python3 scripts/time_spmv.py <pair|offsets|csr> [reps] [KEY=VALUE ...]
Knobs read at creation are set before the handle is made; several VALUEs separated by commas are run one after another."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sprsolve_amd as sa  # noqa: E402
from sprsolve_amd import gen_torch  # noqa: E402

which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sweeps = [(kv.split("=")[0], [int(v) for v in kv.split("=")[1].split(",")]) for kv in sys.argv[3:]]
dev = torch.device("cuda", 0); ctx = sa.default_ctx(0)
ctx.set("spmv_dict", {"pair": -1, "offsets": -1, "csr": 0}[which])
n = 50_000_000
ip, ix, dv, rhs = gen_torch.poisson3d(500, 500, 200, device=dev, values="random" if which == "offsets" else "poisson")
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x); u = torch.rand(n, dtype=torch.float64, device=dev)
w = torch.zeros(n, dtype=torch.float64, device=dev)
for combo in itertools.product(*[vs for _, vs in sweeps]) if sweeps else [()]:
    for (k, _), v in zip(sweeps, combo):
        ctx.set(k, v)
    A = sa.HipCsr.from_device((n, n), int(ip[-1].item()), ip, ix, dv, adopt=False, ctx=ctx)
    us = A.time_mul_vec(x, y, reps) * 1e3
    print(dict(zip([k for k, _ in sweeps], combo)), "tile_plan", A.tile_plan(), "spmv back to back %.1f us" % us, flush=True)
    del A
