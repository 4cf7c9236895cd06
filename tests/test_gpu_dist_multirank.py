"""The distributed solvers with REAL separate ranks on one GPU.

RCCL refuses two ranks on one device, so the library is pointed (SPRS_RCCL_LIB) at tests/mock_rccl — a
shared-memory stand-in for the nine RCCL calls it makes.  Everything else is the product: the exchange plan
(partition.py), pack kernel, grouped send/recv per neighbour, interior/boundary overlap, SpMV, the fused C++
recurrences with all-reduced scalars.  Results are compared with the single-process oracle."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def mock_lib():
    out = os.path.join(tempfile.mkdtemp(prefix="sprs_mock_"), "libmock_rccl.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(HERE, "mock_rccl", "mock_rccl.cpp"), "-o", out, "-L/opt/rocm/lib", "-lamdhip64",
                           "-lrt", "-lpthread"])
    return out


def _run(world, kind, mock_lib, exchange="halo"):
    out = tempfile.mkdtemp(prefix="sprs_distgpu_")
    rdzv = os.path.join(out, "rendezvous")       # file:// store in a private directory: no port is picked, none can be lost
    env = dict(os.environ, SPRS_RCCL_LIB=mock_lib, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_gpu_worker.py"), str(r), str(world), rdzv, kind, out, exchange], env=env)
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    return [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,kind,exchange", [(2, "poisson3d", "halo"), (3, "poisson3d", "halo"), (2, "banded", "halo"),
                                                 (3, "poisson3d", "allgather"), (2, "banded", "allgather"),
                                                 (3, "poisson3d", "halo_c"), (2, "banded", "halo_c"), (3, "banded", "halo_c"),
                                                 (2, "poisson3d", "allgather_c"), (2, "poisson3d_long", "halo"),
                                                 (3, "poisson3d_long", "halo_c"), (2, "poisson3d_tiles", "halo"),
                                                 (2, "poisson3d_tiles_rand", "halo_c")])
def test_real_ranks_match_single_process(oracle, mock_lib, world, kind, exchange):
    from sprsolve_amd import gen
    if kind.startswith("poisson3d"):
        ip, ix, d, rhs = gen.poisson3d(*((272, 6, 8) if kind == "poisson3d_long" else ((160, 128, 24) if kind.startswith("poisson3d_tiles") else (24, 20, 18))),
                                       values="random" if kind.endswith("_rand") else "poisson")
        ref = oracle.bicgstab(ip, ix, d, rhs, np.zeros(rhs.size), 3000, 1e-10, trace_cap=6)
        refpc = oracle.bicgstab(ip, ix, d, rhs, np.zeros(rhs.size), 3000, 1e-10, precond_diag=np.full(rhs.size, 6.0))
    else:
        ip, ix, d, rhs = gen.symmetric_banded(30011, hbw=4)
        ref = oracle.minres(ip, ix, d, rhs, np.zeros(rhs.size), 3000, 1e-10, trace_cap=6)
        refpc = None
    assert ref.status == oracle.OK
    n = rhs.size
    res = _run(world, kind, mock_lib, exchange)
    if kind.startswith("poisson3d_tiles"):
        # every rank's interior launch ran through LDS-window tiles (csrc/dist.hip cuts the interior plan from the handle's)
        for r in res:
            assert int(r["tiles"][0]) >= 8, r["tiles"]
    # SpMV through the real halo exchange: bit-identical to the reference fold on the global matrix
    xg = np.linspace(-1.0, 1.0, n) ** 3
    y = np.concatenate([r["y"] for r in res])
    yref = oracle.spmv(ip, ix, d, xg)
    assert np.array_equal(y.view(np.uint64), yref.view(np.uint64))
    for key in ("fused", "lit"):
        its = [int(r["its_" + key]) for r in res]
        assert len(set(its)) == 1, "ranks disagree on the iteration count: %s" % its
        # (BiCGStab's count at 1e-10 is reduction-order noise: SURVEY §6; it grows with the problem — half a million rows here)
        assert abs(its[0] - ref.its) <= max(3, ref.its // (4 if kind.startswith("poisson3d_tiles") else 10)), (its, ref.its)
        x = np.concatenate([r["x_" + key] for r in res])
        assert np.max(np.abs(x - ref.x)) <= 1e-7 * max(1.0, np.max(np.abs(ref.x)))
        tr = res[0]["trace_" + key]
        for r in res[1:]:
            assert np.array_equal(r["trace_" + key], tr), "ranks must hold bit-identical scalars"
        assert np.allclose(tr[:4], ref.trace[:4], rtol=1e-9, atol=1e-12)
    # the peer-to-peer hand-off (default wherever the ranks could map each other's mailboxes — here: real hipIpc mappings between
    # the 2 / 3 processes) against the ncclAllReduce one: both sum the ranks' values in rank order, so the solves must agree BIT
    # FOR BIT — iteration count, residual, every traced scalar, x — on every rank
    assert all(int(r["p2p"]) == 1 for r in res), "the ranks could not map each other's mailboxes"
    for r in res:
        assert int(r["its_fused"]) == int(r["its_rccl"]) and float(r["res_fused"]) == float(r["res_rccl"])
        assert np.array_equal(r["trace_fused"].view(np.uint64), r["trace_rccl"].view(np.uint64)), "a scalar differs between the mailbox and the RCCL hand-off"
        assert np.array_equal(r["x_fused"].view(np.uint64), r["x_rccl"].view(np.uint64))
    if refpc is not None:
        x = np.concatenate([r["x_pc"] for r in res])
        assert np.max(np.abs(x - refpc.x)) <= 1e-7
        assert len({int(r["its_pc"]) for r in res}) == 1


@pytest.mark.parametrize("world,exchange", [(2, "halo_bad"), (3, "halo_bad"), (3, "allgather_bad")])
def test_plan_builder_errors_are_collective(mock_lib, world, exchange):
    """sprs_dist_csr_create_global_dev_* with a malformed row block on ONE rank (csrc/dist.hip, "collective error
    contract"): every rank returns SPRS_INVALID_ARGUMENT together — the workers would hang in the counts all-gather or the
    index send/recv otherwise (the 300 s limit of _run) — the adopted column arrays are left as they were, and the
    communicator is still usable for a well-formed creation."""
    res = _run(world, "poisson3d", mock_lib, exchange)
    assert [int(r["status"]) for r in res] == [7] * world
    assert all(bool(r["untouched"]) for r in res)
    assert all(int(r["n_ext"]) > 0 for r in res)


def test_bench_launches_its_own_ranks(mock_lib):
    """`python bench.py --gpus 2` with NO launcher around it (how the driver starts the scaling bench): the parent
    must start the two ranks itself, relay rank 0's one JSON line and return 0.  Two ranks share this box's single GPU
    (SPRS_BENCH_DEVICE) through the mock RCCL; gloo bootstraps."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, SPRS_RCCL_LIB=mock_lib, SPRS_BENCH_DEVICE="0", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--grid", "64x48x32",
                        "--steps", "6", "--warmup", "2", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "strong"
    assert d["rccl_ranks"] == 2 and len(d["halo_bytes"]) == 2 and all(b == 64 * 48 * 8 for b in d["halo_bytes"])
    assert d["roofline"]["frac"] <= 1.0 and d["converge_check"]["max_abs_err_vs_exact"] < 1e-5
    # a failing rank must fail the parent
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--grid", "64x48x32",
                        "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--set", "no_such_knob=1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode != 0 and not p.stdout.strip()
