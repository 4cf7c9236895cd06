"""One rank of tests/test_gpu_dist_multirank.py: several processes share GPU 0; the library's RCCL calls go
to tests/mock_rccl (SPRS_RCCL_LIB).  Bootstrap over gloo; data path = the real C++ recurrence + real kernels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(rank, world, rdzv, kind, outdir, exchange="halo"):
    import torch
    import torch.distributed as tdist
    # rendezvous through a file in the caller's private temp directory: no TCP port to pick, none to lose
    tdist.init_process_group("gloo", init_method="file://" + rdzv, rank=rank, world_size=world)
    import sprsolve_amd as sa
    from sprsolve_amd import dist as sdist, gen, partition
    dev = torch.device("cuda", 0)
    ctx = sa.default_ctx(0)
    if kind.startswith("poisson3d"):
        # "_long": x-lines of 272 rows hold full uniform 128-row blocks (scalar pattern, column triples) on every rank
        nx, ny, nz = (272, 6, 8) if kind == "poisson3d_long" else ((160, 128, 24) if kind.startswith("poisson3d_tiles") else (24, 20, 18))
        if kind.startswith("poisson3d_tiles"):
            ctx.set("spmv_tile", 1)      # LDS-window tiles on every rank's slab (12 planes of 20480 rows): interior tiles + boundary blocks
        plane = nx * ny
        starts = partition.slab_starts(nz, plane, world)
        ip, ix, d, rhs = gen.poisson3d(nx, ny, nz, int(starts[rank] // plane), int(starts[rank + 1] // plane),
                                       values="random" if kind.endswith("_rand") else "poisson")
        n = nx * ny * nz
        solver_cls, pdiag = sa.BiCGStab, np.full(rhs.size, 6.0)
    else:   # symmetric banded, MINRES, rows split unevenly
        n = 30011
        gip, gix, gd, grhs = gen.symmetric_banded(n, hbw=4)
        starts = partition.row_starts(n, world)
        r0, r1 = int(starts[rank]), int(starts[rank + 1])
        ip = (gip[r0:r1 + 1] - gip[r0]).astype(np.int32); ix = gix[gip[r0]:gip[r1]]; d = gd[gip[r0]:gip[r1]]; rhs = grhs[r0:r1]
        solver_cls, pdiag = sa.MinRes, None

    def gather(obj):
        out = [None] * world
        tdist.all_gather_object(out, obj)
        return out
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    comm = sdist.Comm(ctx, rank, world, tdist)
    if exchange in ("halo_bad", "allgather_bad"):
        # ONE rank holds an out-of-range column: the plan builder is collective, so EVERY rank must come back with
        # SPRS_INVALID_ARGUMENT (none may be left waiting in an exchange) and no rank's adopted column array may have
        # been renumbered.  A well-formed creation on the same communicator must still work afterwards.
        ixb = np.array(ix, copy=True)
        if rank == world - 1:
            ixb[ixb.size // 2] = n + 7
        col_dev = t(ixb)
        try:
            sdist.DistCsr.from_global(comm, starts, int(ip[-1]), t(ip), col_dev, t(d), exchange=exchange[:-4], adopt=True)
            status = 0
        except ValueError:
            status = 7
        untouched = bool(np.array_equal(col_dev.cpu().numpy(), ixb))
        col2 = t(ix)
        A2 = sdist.DistCsr.from_global(comm, starts, int(ip[-1]), t(ip), col2, t(d), exchange=exchange[:-4], adopt=True)
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), status=status, untouched=untouched, n_ext=A2.plan["n_ext"])
        tdist.barrier()
        comm.close()
        tdist.destroy_process_group()
        return
    if exchange in ("halo_c", "allgather_c"):
        # the plan builder behind the C ABI (csrc/dist.hip) against partition.py's plan of the same row block
        ref = (partition.build_plan(np.asarray(ix), starts, rank, gather) if exchange == "halo_c"
               else partition.allgather_plan(np.asarray(ix), starts, rank))
        col_dev = t(ix)
        A = sdist.DistCsr.from_global(comm, starts, int(ip[-1]), t(ip), col_dev, t(d), exchange=exchange[:-2], adopt=True)
        plan = A.plan
        assert np.array_equal(col_dev.cpu().numpy(), np.asarray(ref["col_ext"])), "renumbered columns differ from partition.py"
        if exchange == "halo_c":
            assert plan["n_local"] == ref["n_local"] and plan["n_ext"] == ref["n_ext"], (plan["n_ext"], ref["n_ext"])
            assert plan["peers"] == [int(q) for q in ref["peers"]]
            assert np.array_equal(plan["send_off"], ref["send_off"]) and np.array_equal(plan["recv_off"], ref["recv_off"])
            assert np.array_equal(plan["send_idx"], ref["send_idx"])
        else:
            assert plan["send_entries"] == ref["slice"]
            plan = dict(plan, n_ext=ref["slice"])
    elif exchange == "allgather":
        plan = partition.allgather_plan(t(ix), starts, rank)
        plan["n_ext"] = plan["slice"]            # the SpMV input only needs this rank's (padded) slice
        A = sdist.DistCsr.from_allgather_plan(comm, plan, int(ip[-1]), t(ip), t(d), adopt=True)
    else:
        plan = partition.build_plan(t(ix), starts, rank, gather)
        A = sdist.DistCsr.from_plan(comm, plan, int(ip[-1]), t(ip), t(d), adopt=True, to_device=t)
    n_loc = plan["n_local"]
    # distributed SpMV of a global test vector
    xg = np.linspace(-1.0, 1.0, n) ** 3
    x_ext = torch.zeros(plan["n_ext"], dtype=torch.float64, device=dev)
    x_ext[:n_loc] = t(xg[int(starts[rank]):int(starts[rank + 1])])
    y = torch.empty(n_loc, dtype=torch.float64, device=dev)
    A.mul_vec_ext(x_ext, y)
    # distributed solves: plain and Jacobi (BiCGStab) / plain (MINRES), fused and literal
    res = {}
    p2p = bool(comm.p2p())          # the ranks mapped each other's mailboxes (real hipIpc handles between these processes)
    for mode in ("fused", "literal", "fused_rccl"):
        # "fused": the scalar hand-offs go through the peer-to-peer mailboxes (no stream operation; csrc/device.hpp mbox_post /
        # mbox_sum2); "fused_rccl": the same solve with ncclAllReduce hand-offs (here: the mock's rank-order sum)
        ctx.set("p2p_allreduce", 0 if mode == "fused_rccl" else -1)
        s = solver_cls.new(A, n_loc); s.set_mode("fused" if mode == "fused_rccl" else mode); s.set_trace(6)
        xs = torch.zeros(n_loc, dtype=torch.float64, device=dev)
        its, rr = s.solve(t(rhs), xs, 3000, 1e-10)
        res[mode] = (its, rr, xs.cpu().numpy(), s.trace())
    ctx.set("p2p_allreduce", -1)
    extra = None
    if pdiag is not None:
        P = sa.DiagPrecond.new(pdiag)
        s = solver_cls.new(A, n_loc)
        xs = torch.zeros(n_loc, dtype=torch.float64, device=dev)
        its, rr = s.precond_solve(P, t(rhs), xs, 3000, 1e-10)
        extra = (its, rr, xs.cpu().numpy())
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), y=y.cpu().numpy(),
             x_fused=res["fused"][2], its_fused=res["fused"][0], res_fused=res["fused"][1], trace_fused=res["fused"][3],
             x_lit=res["literal"][2], its_lit=res["literal"][0], trace_lit=res["literal"][3],
             x_rccl=res["fused_rccl"][2], its_rccl=res["fused_rccl"][0], res_rccl=res["fused_rccl"][1], trace_rccl=res["fused_rccl"][3], p2p=int(p2p),
             x_pc=extra[2] if extra else np.zeros(0), its_pc=extra[0] if extra else -1,
             n_ext=plan["n_ext"], n_loc=n_loc, overlap=int(A.h is not None), tiles=np.array(A.tile_plan() if hasattr(A, "tile_plan") else (0, 0, 0)))
    tdist.barrier()
    comm.close()
    tdist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6] if len(sys.argv) > 6 else "halo")
