"""Multi-GPU front end: one process per GPU, bootstrapped through torch.distributed, collectives
on the data path issued by the C++ recurrence through RCCL (sprsolve_amd/csrc/dist.hip)."""
import ctypes as C

import numpy as np

from . import _lib, partition
from .device import dev_ptr, dev_sfx, pre_sync
from .error import check
from .mat import HipCsr


class Comm:
    """RCCL communicator of this rank.  The 128-byte unique id is created on rank 0 and
    broadcast with torch.distributed (any backend)."""

    def __init__(self, ctx, rank, world, tdist=None):
        L = _lib.lib()
        buf = (C.c_char * 128)()
        if rank == 0:
            check(L.sprs_comm_unique_id(buf), ctx.h)
        if world > 1:
            obj = [bytes(buf)] if rank == 0 else [None]
            tdist.broadcast_object_list(obj, src=0)
            buf = (C.c_char * 128).from_buffer_copy(obj[0])
        h = C.c_void_p()
        check(L.sprs_comm_create(ctx.h, int(world), int(rank), buf, C.byref(h)), ctx.h)
        self.h, self.ctx, self.rank, self.world = h, ctx, rank, world

    def allreduce_sum(self, dev, count):
        pre_sync(dev)
        check(_lib.lib().sprs_comm_allreduce_sum_f64(self.h, dev_ptr(dev), int(count)), self.ctx.h)

    def allreduce_us(self, dev, count=4, reps=200):
        """Mean microseconds of one in-place all-reduce of `count` doubles on the library's stream (collective)."""
        pre_sync(dev)
        us = C.c_double(0.0)
        check(_lib.lib().sprs_comm_allreduce_timed_f64(self.h, dev_ptr(dev), int(count), int(reps), C.byref(us)), self.ctx.h)
        return float(us.value)

    def count(self):
        """ncclCommCount: how many ranks RCCL itself sees on this communicator."""
        n = C.c_int(0)
        check(_lib.lib().sprs_comm_count(self.h, C.byref(n)), self.ctx.h)
        return int(n.value)

    def p2p(self):
        """True when the ranks mapped each other's mailboxes at creation: the solvers' scalar hand-offs then need no stream
        operation (csrc/dist.hip p2p_setup, ctx knob "p2p_allreduce")."""
        n = C.c_int(0)
        check(_lib.lib().sprs_comm_p2p(self.h, C.byref(n)), self.ctx.h)
        return bool(n.value)

    def close(self):
        if self.h:
            _lib.lib().sprs_comm_destroy(self.h)
            self.h = None


class DistCsr(HipCsr):
    """Row block of a matrix partitioned over the ranks; `impl MatVecMul` like HipCsr, usable
    with every solver (`BiCGStab.new(A, A.rows())`, rhs / x = this rank's slices)."""

    @classmethod
    def from_plan(cls, comm, plan, nnz, indptr_dev, data_dev, adopt=True, to_device=None):
        """plan: partition.build_plan(...) with col_ext already a device array; to_device(np_array)
        uploads the small index lists."""
        s = dev_sfx(data_dev)
        peers = np.asarray(plan["peers"], dtype=np.int32)
        send_off = np.ascontiguousarray(plan["send_off"], dtype=np.int64)
        recv_off = np.ascontiguousarray(plan["recv_off"], dtype=np.int64)
        send_idx_dev = to_device(np.ascontiguousarray(plan["send_idx"], dtype=np.int32)) if plan["send_idx"].size else None
        h = C.c_void_p()
        st = getattr(_lib.lib(), "sprs_dist_csr_create_dev_" + s)(
            comm.h, plan["n_local"], plan["n_ext"], int(nnz), dev_ptr(indptr_dev), dev_ptr(plan["col_ext"]),
            dev_ptr(data_dev), 1 if adopt else 0, int(peers.size), peers.ctypes.data_as(C.c_void_p),
            send_off.ctypes.data_as(C.c_void_p), dev_ptr(send_idx_dev) if send_idx_dev is not None else None,
            recv_off.ctypes.data_as(C.c_void_p), C.byref(h))
        check(st, comm.ctx.h)
        from .device import NP_OF
        A = cls(h, comm.ctx, NP_OF[s], (plan["n_local"], plan["n_ext"]), keepalive=(indptr_dev, plan["col_ext"], data_dev, send_idx_dev))
        A.comm, A.plan = comm, plan
        return A

    @classmethod
    def from_allgather_plan(cls, comm, plan, nnz, indptr_dev, data_dev, adopt=True):
        """plan: partition.allgather_plan(...) — full all-gather of x before every SpMV (north_star's literal design)."""
        s = dev_sfx(data_dev)
        h = C.c_void_p()
        st = getattr(_lib.lib(), "sprs_dist_csr_create_allgather_dev_" + s)(
            comm.h, plan["n_local"], plan["slice"], int(nnz), dev_ptr(indptr_dev), dev_ptr(plan["col_ext"]),
            dev_ptr(data_dev), 1 if adopt else 0, C.byref(h))
        check(st, comm.ctx.h)
        from .device import NP_OF
        A = cls(h, comm.ctx, NP_OF[s], (plan["n_local"], plan["slice"] * comm.world), keepalive=(indptr_dev, plan["col_ext"], data_dev))
        A.comm, A.plan = comm, plan
        return A

    @classmethod
    def from_global(cls, comm, starts, nnz, indptr_dev, col_global_dev, data_dev, exchange="halo", adopt=True):
        """The whole setup behind the C ABI (sprs_dist_csr_create_global_dev_*): `col_global_dev` holds GLOBAL column
        numbers (device, int32) and — with adopt — is renumbered in place; the exchange plan is derived on the device
        and over RCCL (csrc/dist.hip), not in Python.  Collective."""
        s = dev_sfx(data_dev)
        st_ = np.ascontiguousarray(starts, dtype=np.int64)
        assert st_.size == comm.world + 1
        pre_sync(indptr_dev, col_global_dev, data_dev)
        h = C.c_void_p()
        st = getattr(_lib.lib(), "sprs_dist_csr_create_global_dev_" + s)(
            comm.h, st_.ctypes.data_as(C.c_void_p), int(nnz), dev_ptr(indptr_dev), dev_ptr(col_global_dev), dev_ptr(data_dev),
            1 if adopt else 0, {"halo": 0, "allgather": 1}[exchange], C.byref(h))
        check(st, comm.ctx.h)
        from .device import NP_OF
        n_local = int(st_[comm.rank + 1] - st_[comm.rank])
        A = cls(h, comm.ctx, NP_OF[s], (n_local, n_local), keepalive=(indptr_dev, col_global_dev, data_dev))
        A.comm = comm
        A.plan = A.plan_info()
        A.plan["mode"] = exchange
        if exchange == "allgather":
            A.plan["slice"] = A.plan["send_entries"]        # every rank contributes its (padded) slice
        A.shape = (n_local, A.plan["n_ext"])
        return A

    def plan_info(self):
        """The exchange plan of this operator read back through the C ABI (sizes, peers, offsets, packed indices)."""
        L = _lib.lib()
        nl, ne, npeer, ns, nr = C.c_int64(), C.c_int64(), C.c_int(), C.c_int64(), C.c_int64()
        check(L.sprs_dist_csr_info(self.h, C.byref(nl), C.byref(ne), C.byref(npeer), C.byref(ns), C.byref(nr)), self.ctx.h)
        k = npeer.value
        peers = np.zeros(max(k, 1), np.int32); so = np.zeros(k + 1, np.int64); ro = np.zeros(k + 1, np.int64)
        check(L.sprs_dist_csr_peers(self.h, max(k, 1), peers.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p),
                                    ro.ctypes.data_as(C.c_void_p)), self.ctx.h)
        sidx = np.zeros(max(int(so[-1]), 1), np.int32)
        check(L.sprs_dist_csr_send_idx(self.h, sidx.size, sidx.ctypes.data_as(C.c_void_p)), self.ctx.h)
        return dict(n_local=nl.value, n_ext=ne.value, peers=[int(p) for p in peers[:k]], send_off=so, recv_off=ro,
                    send_idx=sidx[: int(so[-1])], send_entries=ns.value, recv_entries=nr.value)

    def cols(self):
        # the solvers are created with the number of OWNED entries
        return self.shape[0]

    def mul_vec_ext(self, x_ext, y_local):
        """y_local = A_local * x after the halo exchange; x_ext has n_ext entries, owned slice first."""
        pre_sync(x_ext, y_local)
        st = getattr(_lib.lib(), "sprs_dist_mul_vec_dev_" + self._s())(
            self.h, dev_ptr(x_ext), dev_ptr(y_local))
        check(st, self.ctx.h)
        self.ctx.sync()


def bench_poisson3d(torch, tdist, ctx, rank, world, nx, ny, nz, steps, warmup, time_marginal, exchange="halo"):
    """N > 1 leg of bench.py: cfg 5 row-partitioned in z-slabs, strong scaling."""
    import sprsolve_amd as sa
    from . import gen_torch
    dev = torch.device("cuda", torch.cuda.current_device())
    plane = nx * ny
    starts = partition.slab_starts(nz, plane, world)
    z0, z1 = int(starts[rank] // plane), int(starts[rank + 1] // plane)
    ip, ix, dv, rhs = gen_torch.poisson3d(nx, ny, nz, z0, z1, device=dev)
    nnz_loc = int(ip[-1].item())

    def gather(obj):
        out = [None] * world
        tdist.all_gather_object(out, obj)
        return out
    comm = Comm(ctx, rank, world, tdist)
    # the plan is built by the library (device passes + RCCL), partition.py only names the row ranges
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    A = DistCsr.from_global(comm, starts, nnz_loc, ip, ix, dv, exchange=exchange, adopt=True)
    torch.cuda.synchronize()
    create_ms = (time.perf_counter() - t0) * 1e3        # plan (device passes + RCCL exchanges) + handle (row blocks, dictionaries)
    plan = A.plan
    n_loc = plan["n_local"]
    s = sa.BiCGStab.new(A, n_loc)
    x = torch.zeros(n_loc, dtype=torch.float64, device=dev)
    # the distributed SpMV is an exchange plus two launches: its profile brackets them with two event records each, which
    # cost dispatch slots of their own — the timed regions run without them, a separate pass takes the per-SpMV time
    ms_step, dt, _, trec = time_marginal(torch, tdist, s, None, rhs, x, steps, warmup, world, profile=False)
    _, _, prof, _ = time_marginal(torch, tdist, s, None, rhs, x, max(steps, 100), 0, world, profile=True)
    t_spmv = prof["spmv_ms_total"] / max(prof["spmv_launches"], 1) * 1e-3
    bs = nnz_loc * 12 + (n_loc + 1) * 4 + 2 * n_loc * 8
    x.zero_()
    its, res = s.solve(rhs, x, 5000, 1e-8)
    def reduce_scalar(v, op):
        t_ = torch.tensor([float(v)], dtype=torch.float64, device=dev if tdist.get_backend() == "nccl" else "cpu")
        tdist.all_reduce(t_, op=op)
        return float(t_.item())
    err = reduce_scalar(float((x - 1.0).abs().max().item()), tdist.ReduceOp.MAX)
    tot = reduce_scalar(nnz_loc, tdist.ReduceOp.SUM)
    check_ = dict(tol=1e-8, iters=its, rel_res=res, max_abs_err_vs_exact=err,
                  exchange=exchange, halo_entries_per_rank=int(plan["recv_entries"]), peers=[int(p) for p in plan["peers"]])
    mode, n_off, n_pair = A.stream_format()      # what this rank's SpMV streams (csrc/spmv_dict.hip)
    per_nnz = {0: 12, 1: 9, 2: 1}[mode]
    sinfo = dict(stream={0: "csr", 1: "offset-codes", 2: "pair-codes"}[mode], mode=mode, distinct_offsets=n_off,
                 distinct_pairs=n_pair, bytes_per_nnz=per_nnz,
                 format_bytes_per_launch=nnz_loc * per_nnz + (n_loc + 1) * 4 + 2 * n_loc * 8, rows=n_loc, nnz=nnz_loc)
    nb, nu = A.wide_blocks()
    uf = nu / nb if nb else 0.0
    code_b = 0 if mode == 0 else 1
    sinfo.update(row_blocks=nb, descriptor_only_blocks=nu,
                 bytes_moved_per_launch=int(nnz_loc * (per_nnz - code_b) + (1.0 - uf) * (nnz_loc * code_b + (n_loc + 1) * 4) + 2 * n_loc * 8))
    tiles = A.tile_plan()        # the interior launch's LDS-window tiles (csrc/dist.hip), or the whole slab's at one rank
    if tiles[0] > 0:
        sinfo.update(kernel_id=3 if mode == 2 else 4, lds_window_tiles=tiles[0], blocks_in_tiles=tiles[1], blocks_walked_singly=tiles[2])
    # evidence that the collectives really span `world` ranks, and what each rank moves per SpMV
    halo_b = int(plan["recv_entries"]) * 8
    send_b = int(plan["send_entries"]) * 8
    per_rank = gather(dict(rank=rank, rccl_ranks=comm.count(), halo_recv_bytes=halo_b, halo_send_bytes=send_b,
                           peers=[int(p) for p in plan["peers"]], rows=n_loc, nnz=nnz_loc))
    # what one dot-product hand-off costs on this communicator: an all-reduce of 4 doubles on the solver's stream
    scratch = torch.zeros(4, dtype=torch.float64, device=dev)
    ar_us = comm.allreduce_us(scratch, 4, 200)
    p2p_on = bool(comm.p2p()) and ctx.get("p2p_allreduce") != 0
    dist_info = dict(rccl_ranks=min(p["rccl_ranks"] for p in per_rank), exchange=exchange,
                     halo_bytes=[p["halo_recv_bytes"] for p in per_rank], allreduce_us=ar_us,
                     allreduce_note="mean of 200 back-to-back ncclAllReduce(4 x f64) on the solver's stream; a BiCGStab iteration has 3 scalar hand-offs"
                                    + (" — which this run did NOT pay: scalar_handoff = peer-to-peer mailboxes" if p2p_on else ""),
                     scalar_handoff=("peer-to-peer mailboxes: the producing kernel's last workgroup posts its reduced values into every rank's "
                                     "mailbox (hipIpc-mapped, uncached), the consumer kernels sum the entries in rank order; no stream operation "
                                     "(csrc/device.hpp mbox_post / mbox_sum2)" if p2p_on else "ncclAllReduce on the solver's stream, 3 per iteration"),
                     per_rank=per_rank, ms_per_step=ms_step, timing=trec, create_ms=create_ms)
    return dt, prof, t_spmv, bs, check_, nx * ny * nz, int(tot), sinfo, dist_info
