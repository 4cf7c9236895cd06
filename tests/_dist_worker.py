"""Worker of tests/test_partition_gloo.py: one rank of a world_size-N gloo job on the CPU.

Emulates the multi-GPU data path with numpy + the CPU oracle kernels: the partition / halo plan
of sprsolve_amd.partition drives a distributed SpMV (pack, exchange, local SpMV on the extended
vector) and a distributed BiCGStab (reference op list, dots all-reduced) whose results are
compared with the single-process oracle by the parent test."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_global(kind):
    from sprsolve_amd import gen
    if kind == "poisson3d":
        ip, ix, d, rhs = gen.poisson3d(7, 6, 9)
        return ip, ix, d, rhs, 42          # plane
    if kind == "banded":
        ip, ix, d, rhs = gen.symmetric_banded(503, hbw=4)
        return ip, ix, d, rhs, None
    # ragged: random pattern with far-away columns => every rank talks to every other
    rng = np.random.default_rng(3)
    n = 400
    cnt = rng.integers(1, 8, n)
    ip = np.zeros(n + 1, np.int64); np.cumsum(cnt, out=ip[1:])
    ix = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in cnt])
    d = rng.uniform(-1, 1, ip[-1])
    # make it diagonally dominant so BiCGStab converges
    rows = np.repeat(np.arange(n), cnt)
    d[ix == rows] = 0.0
    dg = np.zeros(n); np.add.at(dg, rows, np.abs(d))
    import scipy.sparse as sp
    M = sp.csr_matrix((d, ix, ip), shape=(n, n)) + sp.diags(dg + 1.0)
    M = M.tocsr(); M.sort_indices()
    return M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data, rng.uniform(-1, 1, n), None


class DistEmu:
    """numpy twin of csrc/dist.hip + the distributed branches of csrc/krylov.hip."""

    def __init__(self, dist, rank, world, plan, ip_loc, data_loc, orc):
        self.dist, self.rank, self.world, self.plan, self.orc = dist, rank, world, plan, orc
        self.ip, self.data = ip_loc, data_loc
        self.col = plan["col_ext"]
        self.n = plan["n_local"]

    def halo(self, x_ext):
        import torch
        pl = self.plan
        reqs = []
        recv_bufs = []
        for k, p in enumerate(pl["peers"]):
            ns = pl["send_off"][k + 1] - pl["send_off"][k]
            nr = pl["recv_off"][k + 1] - pl["recv_off"][k]
            if ns:
                buf = torch.from_numpy(np.ascontiguousarray(x_ext[pl["send_idx"][pl["send_off"][k]:pl["send_off"][k + 1]]]))
                reqs.append(self.dist.isend(buf, p))
            if nr:
                rb = torch.empty(int(nr), dtype=torch.float64)
                recv_bufs.append((k, rb))
                reqs.append(self.dist.irecv(rb, p))
        for r in reqs:
            r.wait()
        for k, rb in recv_bufs:
            x_ext[self.n + pl["recv_off"][k]: self.n + pl["recv_off"][k + 1]] = rb.numpy()

    def spmv(self, x_local):
        x_ext = np.zeros(self.plan["n_ext"]); x_ext[: self.n] = x_local
        self.halo(x_ext)
        return self.orc.spmv(self.ip, self.col, self.data, x_ext)

    def allsum(self, v):
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t)
        return float(t.item())

    def dot(self, a, b):
        return self.allsum(self.orc.conj_dot(a, b))

    def norm2(self, a):
        return np.sqrt(self.allsum(self.orc.norm2(a) ** 2))

    def bicgstab(self, rhs, max_iter, tol):
        """src/bicg_stab.rs:35-200 with distributed SpMV / dots (no restart/breakdown on these inputs)."""
        orc = self.orc
        x = np.zeros(self.n)
        rhs_norm = self.norm2(rhs); tol2 = tol * rhs_norm
        r = self.spmv(x); orc.axpy(-1.0, rhs, r)
        r0 = r.copy(); r0n = self.norm2(r0)
        rho = r0n * r0n
        y = r.copy(); v = self.spmv(y)
        alpha = rho / self.dot(r0, v)
        orc.axpy(-alpha, v, r); t = self.spmv(r)
        tt = self.dot(t, t); w = self.dot(t, r) / tt if tt > 0 else 0.0
        orc.axpy(-alpha, y, x); orc.axpy(-w, r, x); orc.axpy(-w, t, r)
        trace = [(0, r0n, rho, alpha, w)]
        for its in range(1, max_iter):
            rn = self.norm2(r)
            if rn <= tol2:
                return x, its, rn / rhs_norm, trace
            rho_old = rho; rho = self.dot(r0, r)
            beta = (rho / rho_old) * (alpha / w)
            orc.axpby(-beta * w, v, beta, y); orc.axpy(1.0, r, y)
            v = self.spmv(y)
            alpha = rho / self.dot(r0, v)
            orc.axpy(-alpha, v, r); t = self.spmv(r)
            tt = self.dot(t, t); w = self.dot(t, r) / tt if tt > 0 else 0.0
            orc.axpy(-alpha, y, x); orc.axpy(-w, r, x); orc.axpy(-w, t, r)
            trace.append((its, rn, rho, alpha, w))
        return x, max_iter, None, trace


def main(rank, world, rdzv, kind, outdir):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + rdzv, rank=rank, world_size=world)   # no TCP port to pick
    from oracle import oracle as orc
    from sprsolve_amd import partition
    ip, ix, d, rhs, plane = build_global(kind)
    n = rhs.size
    starts = partition.slab_starts(n // plane, plane, world) if plane else partition.row_starts(n, world)
    r0, r1 = int(starts[rank]), int(starts[rank + 1])
    ip_loc = (ip[r0:r1 + 1] - ip[r0]).astype(np.int32)
    ix_loc = ix[ip[r0]:ip[r1]]
    d_loc = d[ip[r0]:ip[r1]]

    def gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    plan = partition.build_plan(ix_loc, starts, rank, gather)
    emu = DistEmu(dist, rank, world, plan, ip_loc, d_loc, orc)
    xg = np.linspace(-1.0, 1.0, n) ** 3
    y_loc = emu.spmv(xg[r0:r1])
    x_sol, its, res, trace = emu.bicgstab(rhs[r0:r1], 400, 1e-10)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), y=y_loc, x=x_sol, its=its, trace=np.array(trace),
             n_ext=plan["n_ext"], peers=np.array(plan["peers"]), r0=r0, r1=r1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5])
