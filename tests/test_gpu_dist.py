"""The RCCL data path on ONE GPU (the only multi-rank-capable box for N > 1 is the driver's):
a world_size-1 communicator exercises every RCCL call the distributed solvers make — the
unique-id bootstrap, ncclAllReduce of the dot scalars, and grouped ncclSend/ncclRecv of the halo
(to self: a set of owned columns is declared 'remote' and travels through pack -> send -> recv
into the halo tail).  Results must equal the plain single-GPU path bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import sprsolve_amd as sa
    from sprsolve_amd import dist as sdist
    ctx = sa.default_ctx(0)
    comm = sdist.Comm(ctx, 0, 1)
    yield dict(torch=torch, sa=sa, sdist=sdist, ctx=ctx, comm=comm, dev=torch.device("cuda", 0))
    ctx.set("spmv_wide", -1)
    comm.close()


def _self_halo_plan(torch, dev, n, indices, remote_mask_fn, segments=1):
    """Declare the owned columns selected by remote_mask_fn as 'remote, owned by rank 0 (self)'."""
    cols = np.unique(indices[remote_mask_fn(indices)])
    pos = np.searchsorted(cols, indices)
    is_remote = remote_mask_fn(indices)
    col_ext = np.where(is_remote, n + np.minimum(pos, max(cols.size - 1, 0)), indices).astype(np.int32)
    k = cols.size
    if k and segments == 2:
        # two exchange segments, both with rank 0 (self): exercises the per-peer offset arithmetic
        peers, off = [0, 0], np.array([0, k // 3, k], dtype=np.int64)
    else:
        peers = [0] if k else []
        off = np.array([0, k] if k else [0], dtype=np.int64)
    return dict(col_ext=torch.from_numpy(col_ext).to(dev), n_local=n, n_ext=n + k, peers=peers,
                send_off=off, send_idx=cols.astype(np.int32), recv_off=off)


@pytest.mark.parametrize("with_halo", ["scattered", "tail-overlapped", "tail-no-overlap"])
def test_dist_operator_wide_kernel(env, oracle, with_halo):
    """The same distributed SpMV with the two-rows-per-lane kernel on the interior / boundary subsets (split made on
    pairs of 64-row blocks): y bit-identical to the reference fold; the solve agrees with the oracle."""
    torch, sa, sdist, dev = env["torch"], env["sa"], env["sdist"], env["dev"]
    from sprsolve_amd import gen
    R = 96
    indptr, indices, data = gen.grid_laplacian_dirichlet(R, R)
    rhs = gen.dirichlet_rhs(R, R)
    n = R * R
    mask = {"scattered": lambda c: (c % 7 == 3) | (c > n - 2 * R)}.get(with_halo, lambda c: c > n - 3 * R)
    plan = _self_halo_plan(torch, dev, n, indices, mask)
    env["ctx"].set("halo_overlap", 0 if with_halo == "tail-no-overlap" else 1)
    env["ctx"].set("spmv_wide", 1)
    ip_d = torch.from_numpy(indptr).to(dev); dv_d = torch.from_numpy(data).to(dev)
    A = sdist.DistCsr.from_plan(env["comm"], plan, int(indptr[-1]), ip_d, dv_d, adopt=True,
                                to_device=lambda a: torch.from_numpy(a).to(dev))
    if with_halo != "scattered":
        assert A.stream_format()[0] == 2          # few (offset, value) pairs survive the halo renumbering
    x = np.linspace(-1, 1, n) ** 3
    x_ext = torch.zeros(plan["n_ext"], dtype=torch.float64, device=dev)
    x_ext[:n] = torch.from_numpy(x).to(dev)
    x_ext[n:] = float("nan")
    y = torch.empty(n, dtype=torch.float64, device=dev)
    A.mul_vec_ext(x_ext, y)
    ref = oracle.spmv(indptr, indices, data, x)
    assert np.array_equal(y.cpu().numpy().view(np.uint64), ref.view(np.uint64))
    xs = torch.zeros(n, dtype=torch.float64, device=dev)
    its, res = sa.BiCGStab.new(A, n).solve(torch.from_numpy(rhs).to(dev), xs, 5000, 1e-10)
    i, j = np.meshgrid(np.arange(R), np.arange(R), indexing="ij")
    assert np.max(np.abs(xs.cpu().numpy() - (i + j).ravel())) < 1e-6
    env["ctx"].set("spmv_wide", -1)


@pytest.mark.parametrize("with_halo", ["none", "scattered", "tail-overlapped", "tail-2-segments", "tail-no-overlap"])
def test_dist_operator_equals_plain(env, oracle, with_halo):
    torch, sa, sdist, dev = env["torch"], env["sa"], env["sdist"], env["dev"]
    from sprsolve_amd import gen
    R = 96
    indptr, indices, data = gen.grid_laplacian_dirichlet(R, R)
    rhs = gen.dirichlet_rhs(R, R)
    n = R * R
    mask = {"none": lambda c: np.zeros(c.shape, bool),
            "scattered": lambda c: (c % 7 == 3) | (c > n - 2 * R),      # every row block touches the halo: no split
            }.get(with_halo, lambda c: c > n - 3 * R)                   # only the last rows do: interior/boundary overlap
    plan = _self_halo_plan(torch, dev, n, indices, mask, segments=2 if with_halo == "tail-2-segments" else 1)
    env["ctx"].set("halo_overlap", 0 if with_halo == "tail-no-overlap" else 1)
    # bit-for-bit equality of the SCALARS needs the same grouping of the fused dot partials: the plain handle must
    # use the 64-row kernels like the distributed operator's subset launches (y itself is bit-identical either way)
    env["ctx"].set("spmv_wide", 0)
    if with_halo != "none":
        assert plan["n_ext"] > n
    ip_d = torch.from_numpy(indptr).to(dev); dv_d = torch.from_numpy(data).to(dev)
    A = sdist.DistCsr.from_plan(env["comm"], plan, int(indptr[-1]), ip_d, dv_d, adopt=True,
                                to_device=lambda a: torch.from_numpy(a).to(dev))
    # SpMV through the halo exchange == reference fold, bit for bit
    x = np.linspace(-1, 1, n) ** 3
    x_ext = torch.zeros(plan["n_ext"], dtype=torch.float64, device=dev)
    x_ext[:n] = torch.from_numpy(x).to(dev)
    x_ext[n:] = float("nan")                       # the halo tail must be overwritten by the exchange
    y = torch.empty(n, dtype=torch.float64, device=dev)
    A.mul_vec_ext(x_ext, y)
    ref = oracle.spmv(indptr, indices, data, x)
    assert np.array_equal(y.cpu().numpy().view(np.uint64), ref.view(np.uint64))

    # the solvers on the distributed operator follow the plain path bit for bit (world = 1:
    # same partials, same local reduction order, all-reduce of one rank is the identity)
    P = sa.DiagPrecond.new(np.where(np.diff(indptr) == 1, 1.0, -4.0))
    plain = sa.HipCsr.new((n, n), indptr, indices, data)
    K = 12
    for cls, pc in ((sa.BiCGStab, None), (sa.BiCGStab, P), (sa.MinRes, None)):
        outs = []
        for op in (plain, A):
            s = cls.new(op, n); s.set_trace(K)
            xs = torch.zeros(n, dtype=torch.float64, device=dev)
            b = torch.from_numpy(rhs).to(dev)
            try:
                if pc is not None:
                    s.precond_solve(pc, b, xs, K, 0.0)
                else:
                    s.solve(b, xs, K, 0.0)
            except sa.error.InsufficientIterNum:
                pass
            outs.append((s.trace(), xs.cpu().numpy()))
        assert outs[0][0].shape[0] == K
        if with_halo in ("tail-overlapped", "tail-2-segments"):
            # interior and boundary rows are multiplied by two launches: the dot partials are grouped
            # differently, so scalars agree to rounding, not bit for bit (and on this Dirichlet grid the
            # trajectories part after the unrolled iteration — see test_solver_trace_lockstep)
            assert np.allclose(outs[0][0][0], outs[1][0][0], rtol=1e-12, atol=0), cls.__name__
        else:
            assert np.array_equal(outs[0][0], outs[1][0], equal_nan=True), cls.__name__
            assert np.array_equal(outs[0][1].view(np.uint64), outs[1][1].view(np.uint64))
    # and converge to the known solution through the distributed operator
    s = sa.BiCGStab.new(A, n)
    xs = torch.zeros(n, dtype=torch.float64, device=dev)
    its, res = s.precond_solve(P, torch.from_numpy(rhs).to(dev), xs, 5000, 1e-10)
    i, j = np.meshgrid(np.arange(R), np.arange(R), indexing="ij")
    assert np.max(np.abs(xs.cpu().numpy() - (i + j).ravel())) < 1e-5


@pytest.mark.parametrize("values", ["poisson", "random"])
def test_in_launch_finalize_under_load(env, values):
    """The distributed hand-offs' final reduction is made by the last-arriving workgroup of the producing launch
    (csrc/device.hpp, finalize_last_block: sc1 stores of the partials, drained, an agent-scope arrival counter, sc1 loads
    by the workgroup whose add came last).  A stale or missing partial would change a scalar: 300 BiCGStab iterations
    on a 2 M-row system with full grids (1024 SpMV workgroups, ~500 fused-kernel workgroups, all eight XCDs) through the
    distributed path at world 1 must reproduce the plain path BIT FOR BIT — every traced scalar and x — three times."""
    torch, sa, sdist, dev = env["torch"], env["sa"], env["sdist"], env["dev"]
    from sprsolve_amd import gen_torch
    nx, ny, nz = 200, 200, 50
    n = nx * ny * nz
    ip, ix, dv, rhs = gen_torch.poisson3d(nx, ny, nz, device=dev, values=values)
    nnz = int(ip[-1].item())
    plain = sa.HipCsr.from_device((n, n), nnz, ip, ix, dv, adopt=True)
    ixg = ix.clone()                                          # from_global renumbers in place (world 1: the identity)
    A = sdist.DistCsr.from_global(env["comm"], np.array([0, n], dtype=np.int64), nnz, ip, ixg, dv, exchange="halo", adopt=True)
    assert A.plan["peers"] == [] and A.stream_format()[0] == plain.stream_format()[0]
    K = 300

    def run(op):
        s = sa.BiCGStab.new(op, n); s.set_trace(K)
        xs = torch.zeros(n, dtype=torch.float64, device=dev)
        try:
            s.solve(rhs, xs, K, 0.0)
        except sa.error.InsufficientIterNum:
            pass
        return s.trace(), xs
    t0, x0 = run(plain)
    assert np.isfinite(t0[:50]).all()
    for rep in range(3):
        t1, x1 = run(A)
        assert np.array_equal(t0, t1, equal_nan=True), "scalars differ in repetition %d" % rep
        assert torch.equal(x0.view(torch.int64), x1.view(torch.int64))


def test_comm_allreduce_world1(env):
    torch = env["torch"]
    t = torch.arange(8, dtype=torch.float64, device=env["dev"])
    env["comm"].allreduce_sum(t, 8)
    assert t.cpu().tolist() == list(range(8))


def test_device_built_matrix_is_validated(env):
    """A CSR assembled in HBM never passes through host validation: out-of-range columns and a
    broken row_ptr must be refused at creation instead of faulting the GPU in the SpMV."""
    torch, sa, dev = env["torch"], env["sa"], env["dev"]
    from sprsolve_amd import gen
    indptr, indices, data = gen.grid_laplacian_dirichlet(16, 16)
    n, nnz = 256, int(indptr[-1])
    ip = torch.from_numpy(indptr).to(dev); dv = torch.from_numpy(data).to(dev)
    bad = indices.copy(); bad[77] = n + 5
    with pytest.raises(ValueError):
        sa.HipCsr.from_device((n, n), nnz, ip, torch.from_numpy(bad).to(dev), dv)
    bad = indices.copy(); bad[3] = -1
    with pytest.raises(ValueError):
        sa.HipCsr.from_device((n, n), nnz, ip, torch.from_numpy(bad).to(dev), dv)
    ipb = indptr.copy(); ipb[10] = ipb[11] + 1
    with pytest.raises(ValueError):
        sa.HipCsr.from_device((n, n), nnz, torch.from_numpy(ipb).to(dev), torch.from_numpy(indices).to(dev), dv)
    A = sa.HipCsr.from_device((n, n), nnz, ip, torch.from_numpy(indices).to(dev), dv)
    assert A.nnz() == nnz


def test_invalid_plan_is_rejected(env):
    torch, sa, sdist, dev = env["torch"], env["sa"], env["sdist"], env["dev"]
    from sprsolve_amd import gen
    indptr, indices, data = gen.grid_laplacian_dirichlet(8, 8)
    n = 64
    plan = dict(col_ext=torch.from_numpy(indices).to(dev), n_local=n, n_ext=n + 3, peers=[0],
                send_off=np.array([0, 3], np.int64), send_idx=np.array([1, 2, 99], np.int32),   # 99 is out of range
                recv_off=np.array([0, 3], np.int64))
    with pytest.raises(ValueError):
        sdist.DistCsr.from_plan(env["comm"], plan, int(indptr[-1]), torch.from_numpy(indptr).to(dev),
                                torch.from_numpy(data).to(dev), to_device=lambda a: torch.from_numpy(a).to(dev))


@pytest.mark.parametrize("exchange", ["halo", "allgather"])
def test_plan_builder_with_the_real_rccl(env, oracle, exchange):
    """sprs_dist_csr_create_global_dev_* on a world-size-1 communicator of the REAL librccl: the device passes
    (mark / scan / renumber), ncclAllGather of the per-peer counts and the (empty) send/recv group all run; with one
    rank every column is owned, so the operator must equal the plain one bit for bit, and ncclCommCount says 1.
    Multi-rank plans are checked against partition.py in tests/test_gpu_dist_multirank.py (mock transport)."""
    torch, sa, sdist, dev = env["torch"], env["sa"], env["sdist"], env["dev"]
    from sprsolve_amd import gen
    ip, ix, d, rhs = gen.poisson3d(20, 18, 16)
    n = rhs.size
    assert env["comm"].count() == 1
    col = torch.from_numpy(ix).to(dev)
    A = sdist.DistCsr.from_global(env["comm"], np.array([0, n], np.int64), int(ip[-1]), torch.from_numpy(ip).to(dev), col,
                                  torch.from_numpy(d).to(dev), exchange=exchange)
    assert np.array_equal(col.cpu().numpy(), ix)                       # nothing is remote: numbering unchanged
    assert A.plan["peers"] == [] and A.plan["n_local"] == n
    x = np.linspace(-1, 1, n) ** 3
    x_ext = torch.zeros(A.plan["n_ext"] if exchange == "halo" else A.plan["slice"], dtype=torch.float64, device=dev)
    x_ext[:n] = torch.from_numpy(x).to(dev)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    A.mul_vec_ext(x_ext, y)
    assert np.array_equal(y.cpu().numpy().view(np.uint64), oracle.spmv(ip, ix, d, x).view(np.uint64))
    xs = torch.zeros(n, dtype=torch.float64, device=dev)
    sa.BiCGStab.new(A, n).solve(torch.from_numpy(rhs).to(dev), xs, 3000, 1e-10)
    assert np.max(np.abs(xs.cpu().numpy() - 1.0)) < 1e-7
    # a column outside the global range is refused, not renumbered
    bad = ix.copy(); bad[11] = n + 3
    with pytest.raises(ValueError):
        sdist.DistCsr.from_global(env["comm"], np.array([0, n], np.int64), int(ip[-1]), torch.from_numpy(ip).to(dev),
                                  torch.from_numpy(bad).to(dev), torch.from_numpy(d).to(dev), exchange=exchange)


def test_bench_stdout_is_one_json_line_with_the_real_rccl():
    """bench.py's contract is ONE JSON line on stdout.  RCCL prints a version banner on file descriptor 1 when its first
    communicator is created; bench.py points descriptor 1 at stderr for the run and writes its line to a duplicate of
    the original.  `--force-dist` takes the RCCL path with one rank (the real library, as the driver's N > 1 runs do)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SPRS_RCCL_LIB"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--grid", "64x48x32", "--steps", "6",
                        "--warmup", "2", "--no-cpu-baseline", "--no-also"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1, p.stdout[:2000]
    d = json.loads(lines[0])
    assert d["rccl_ranks"] == 1 and d["converge_check"]["max_abs_err_vs_exact"] < 1e-5
