"""The N > 1 path on the CPU: world_size-2 (and 3) gloo jobs exercise sprsolve_amd.partition
(row ranges, column localisation, halo plan exchange) and a numpy twin of the distributed
recurrence; results are compared with the single-process oracle."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(world, kind):
    out = tempfile.mkdtemp(prefix="sprs_dist_")
    rdzv = os.path.join(out, "rendezvous")       # file:// store: no port to pick
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), str(world), rdzv, kind, out],
                              env=dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1"))
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    return [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,kind", [(2, "poisson3d"), (3, "poisson3d"), (2, "banded"), (3, "ragged")])
def test_distributed_path_matches_single_process(oracle, world, kind):
    sys.path.insert(0, HERE)
    import _dist_worker as W
    ip, ix, d, rhs, plane = W.build_global(kind)
    n = rhs.size
    res = _run(world, kind)
    # 1) distributed SpMV == global SpMV, bit for bit (same per-row fold, columns only renumbered)
    xg = np.linspace(-1.0, 1.0, n) ** 3
    y_ref = oracle.spmv(ip, ix, d, xg)
    y = np.concatenate([r["y"] for r in res])
    assert np.array_equal(y.view(np.uint64), y_ref.view(np.uint64))
    # 2) halo sizes: a z-slab partition of the 7-point stencil needs exactly one plane per neighbour
    if kind == "poisson3d":
        for r, rr in enumerate(res):
            nb = (r > 0) + (r < world - 1)
            assert int(rr["n_ext"]) - int(rr["r1"] - rr["r0"]) == nb * plane
            assert sorted(rr["peers"].tolist()) == [p for p in (r - 1, r + 1) if 0 <= p < world]
    if kind == "ragged":
        assert all(len(rr["peers"]) == world - 1 for rr in res)
    # 3) distributed BiCGStab == single-process oracle up to the summation order of the dots
    ref = oracle.bicgstab(ip, ix, d, rhs, np.zeros(n), 400, 1e-10, trace_cap=4)
    assert ref.status == oracle.OK
    x = np.concatenate([r["x"] for r in res])
    assert all(int(r["its"]) == int(res[0]["its"]) for r in res), "ranks must agree on the iteration count"
    assert abs(int(res[0]["its"]) - ref.its) <= max(2, ref.its // 10)
    assert np.max(np.abs(x - ref.x)) <= 1e-7 * max(1.0, np.max(np.abs(ref.x)))
    tr = res[0]["trace"][:4]
    assert np.allclose(tr[:, 1:], ref.trace[:, [1, 2, 4, 6]][: tr.shape[0]], rtol=1e-9, atol=1e-12)
    for r in res[1:]:
        assert np.array_equal(r["trace"], res[0]["trace"]), "all ranks must compute bit-identical scalars"


def test_partition_helpers():
    from sprsolve_amd import partition as P
    assert P.row_starts(10, 3).tolist() == [0, 3, 6, 10]
    assert P.slab_starts(200, 250000, 8).tolist() == [25 * 250000 * r for r in range(9)]
    # torch twin of localize gives the same renumbering
    import torch
    from sprsolve_amd import gen
    ip, ix, d, rhs = gen.poisson3d(5, 4, 6, 2, 4)
    starts = P.slab_starts(6, 20, 3)
    a = P.localize(ix, starts, 1)
    b = P.localize(torch.from_numpy(ix), starts, 1)
    assert np.array_equal(a[0], b[0].numpy()) and a[3] == b[3] and np.array_equal(a[2], b[2])
    assert all(np.array_equal(a[1][p], b[1][p]) for p in a[1])
    assert a[0].max() == 40 + 2 * 20 - 1


def test_allgather_plan_renumbering():
    """north_star's literal exchange: columns renumbered into [rank 0 slice | rank 1 slice | ...] with padded slices."""
    import torch
    from sprsolve_amd import gen, partition as P
    ip, ix, d, rhs = gen.symmetric_banded(1003, hbw=4)
    starts = P.row_starts(1003, 3)                      # 334 / 334 / 335 rows -> slice 336 (even)
    for rank in range(3):
        r0, r1 = int(starts[rank]), int(starts[rank + 1])
        cols = ix[ip[r0]:ip[r1]]
        a = P.allgather_plan(cols, starts, rank)
        b = P.allgather_plan(torch.from_numpy(cols), starts, rank)
        assert a["slice"] == 336 and a["n_local"] == r1 - r0
        assert np.array_equal(a["col_ext"], b["col_ext"].numpy())
        owner = np.searchsorted(starts, cols, side="right") - 1
        assert np.array_equal(a["col_ext"], owner * 336 + (cols - starts[owner]))
        # a gathered vector built that way reproduces the global SpMV rows of this rank
        xg = np.linspace(-1, 1, 1003)
        gathered = np.zeros(3 * 336)
        for q in range(3):
            gathered[q * 336: q * 336 + int(starts[q + 1] - starts[q])] = xg[int(starts[q]):int(starts[q + 1])]
        import scipy.sparse as sp
        loc = sp.csr_matrix((d[ip[r0]:ip[r1]], a["col_ext"], ip[r0:r1 + 1] - ip[r0]), shape=(r1 - r0, 3 * 336))
        glob = sp.csr_matrix((d, ix, ip), shape=(1003, 1003))
        assert np.array_equal(loc @ gathered, (glob @ xg)[r0:r1])
