"""GPU parity of GaussSeidel (csrc/gs.hip, through the C ABI) against the oracle.  The level-scheduled
sweep keeps the reference's per-row fold, so the iterates are BIT-IDENTICAL; only the residual norm (a
reduction) is summed in another order, which can move the stopping sweep when eps sits at rounding level."""
import numpy as np
import pytest

import _golden as G

pytestmark = pytest.mark.gpu
CASES = G.load("gs_kat.json")["cases"]


@pytest.fixture(scope="module")
def sa():
    import sprsolve_amd
    from sprsolve_amd import _lib
    _lib.lib()
    sprsolve_amd.default_ctx(0)
    return sprsolve_amd


def _csr(sa, p):
    n = p["rhs"].size
    return sa.HipCsr.new((n, n), p["indptr"], p["indices"], p["data"])


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_gs_golden(sa, oracle, case):
    p = G.gs_problem(case)
    A = _csr(sa, p)
    gs = sa.GaussSeidel.new(A)
    x = np.zeros_like(p["rhs"])
    ref = oracle.gauss_seidel(p["indptr"], p["indices"], p["data"], p["rhs"], np.zeros_like(p["rhs"]), case["max_iter"], case["eps"])
    if case.get("expect") == "InsufficientIterNum":
        with pytest.raises(sa.error.InsufficientIterNum) as e:
            gs.solve(p["rhs"], x, case["max_iter"], case["eps"])
        assert e.value.iters == case["max_iter"]
        assert np.array_equal(x, ref.x)                  # the sweeps themselves are bit-identical
        return
    its, res = gs.solve(p["rhs"], x, case["max_iter"], case["eps"])
    if "oracle_its" in case:
        # tests/test_solvers.rs:2-31: Ok with eps = 0  =>  exactly zero residual; same sweep count as the oracle
        assert (its, res) == (case["oracle_its"], case["oracle_res"])
        assert np.array_equal(x, p["exact"]) and np.array_equal(x, ref.x)
    else:
        assert abs(its - ref.its) <= 1
        if its == ref.its:
            assert np.array_equal(x, ref.x)
            assert np.isclose(res, ref.res, rtol=1e-6 if case["dtype"] == "f64" else 1e-2)
    ax = oracle.spmv(p["indptr"], p["indices"], p["data"], x)
    assert np.linalg.norm(ax - p["rhs"]) <= 1.01 * case["eps"] * np.linalg.norm(p["rhs"]) + 1e-30


@pytest.mark.parametrize("graph", [0, 1], ids=["launches", "hipgraph"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("sweeps", [1, 2, 9])
def test_gs_sweeps_bit_exact(sa, oracle, dtype, sweeps, graph):
    """k sweeps from a random start on a 3-D 7-point matrix: x identical to the serial sweep, bit for bit."""
    from sprsolve_amd import gen
    indptr, indices, data, _ = gen.poisson3d(9, 8, 7)
    data = data.astype(dtype)
    n = indptr.size - 1
    rng = np.random.default_rng(11)
    rhs = rng.uniform(-1, 1, n).astype(dtype); x0 = rng.uniform(-1, 1, n).astype(dtype)
    A = sa.HipCsr.new((n, n), indptr, indices, data)
    sa.default_ctx(0).set("gs_graph", graph)
    gs = sa.GaussSeidel.new(A)
    assert 1 < gs.levels <= 9 + 8 + 7 - 2              # hyperplanes i+j+k of the 7-point stencil
    x = x0.copy()
    with pytest.raises(sa.error.InsufficientIterNum):
        gs.solve(rhs, x, sweeps, 0.0)
    ref = oracle.gauss_seidel(indptr, indices, data, rhs, x0, sweeps, 0.0)
    sa.default_ctx(0).set("gs_graph", 0)
    assert ref.status == oracle.INSUFFICIENT_ITER
    assert np.array_equal(x, ref.x)


def test_gs_random_pattern_bit_exact(sa, oracle):
    """Irregular pattern with unsorted dependency depth, duplicate-free random columns, strong diagonal."""
    import scipy.sparse as sp
    n = 3000
    rng = np.random.default_rng(5)
    M = sp.random(n, n, density=0.002, random_state=rng, format="csr", dtype=np.float64)
    M = (M + sp.diags(np.full(n, 4.0) + np.asarray(abs(M).sum(axis=1)).ravel())).tocsr()
    M.sort_indices()
    rhs = rng.uniform(-1, 1, n); x0 = rng.uniform(-1, 1, n)
    A = sa.HipCsr.new((n, n), M.indptr, M.indices, M.data)
    gs = sa.GaussSeidel.new(A)
    x = x0.copy()
    its, res = gs.solve(rhs, x, 200, 1e-12)
    ref = oracle.gauss_seidel(M.indptr, M.indices, M.data, rhs, x0, 200, 1e-12)
    assert ref.status == oracle.OK and abs(its - ref.its) <= 1
    if its == ref.its:
        assert np.array_equal(x, ref.x)
    assert np.linalg.norm(M @ x - rhs) <= 1.01e-12 * np.linalg.norm(rhs)


def test_gs_device_vectors(sa, oracle):
    from sprsolve_amd import gen
    indptr, indices, data = gen.grid_laplacian_dirichlet(16, 16)
    rhs = gen.dirichlet_rhs(16, 16)
    A = sa.HipCsr.new((256, 256), indptr, indices, data)
    gs = sa.GaussSeidel.new(A)
    d_rhs = sa.DevVec.from_numpy(rhs); d_x = sa.DevVec.from_numpy(np.zeros(256))
    its, res = gs.solve(d_rhs, d_x, 3000, 1e-9)
    ref = oracle.gauss_seidel(indptr, indices, data, rhs, np.zeros(256), 3000, 1e-9)
    assert abs(its - ref.its) <= 1
    if its == ref.its:
        assert np.array_equal(d_x.to_numpy(), ref.x)


def test_gs_errors(sa, oracle):
    from sprsolve_amd import gen
    indptr, indices, data = gen.grid_laplacian_dirichlet(5, 5)
    rhs = gen.dirichlet_rhs(5, 5)
    A = sa.HipCsr.new((25, 25), indptr, indices, data)
    gs = sa.GaussSeidel.new(A)
    x = np.zeros(25)
    with pytest.raises(sa.error.InsufficientIterNum) as e:       # gauss_seidel.rs:52-54
        gs.solve(rhs, x, 0, 1e-8)
    assert e.value.iters == 0 and not x.any()
    with pytest.raises(sa.error.IncompatibleMatrixFormat, match="doesn't match the matrix size"):
        gs.solve(rhs[:24], np.zeros(24), 10, 1e-8)
    with pytest.raises(sa.error.IncompatibleMatrixFormat, match="do not match"):
        gs.solve(rhs, np.zeros(24), 10, 1e-8)
    # new(): not square / not CSR (gauss_seidel.rs:16-26)
    R = sa.HipCsr.new((2, 3), np.array([0, 1, 2]), np.array([0, 1]), np.array([1.0, 1.0]))
    with pytest.raises(sa.error.IncompatibleMatrixFormat, match="Not a square matrix"):
        sa.GaussSeidel.new(R)
    Ccsc = sa.HipCsr.new((25, 25), indptr, indices, data, storage="CSC")
    with pytest.raises(sa.error.IncompatibleMatrixFormat, match="Not in CSR format"):
        sa.GaussSeidel.new(Ccsc)
    Z = sa.HipCsr.new((25, 25), indptr, indices, data.astype(np.complex128))
    with pytest.raises(TypeError):
        sa.GaussSeidel.new(Z)
    # zero diagonal: rows before the offending one have been updated, the rest untouched (:72-78)
    bad = data.copy(); row = 12
    for k in range(indptr[row], indptr[row + 1]):
        if indices[k] == row:
            bad[k] = 1e-9
    B = sa.HipCsr.new((25, 25), indptr, indices, bad)
    x0 = np.full(25, 0.25); x = x0.copy()
    with pytest.raises(sa.error.ZeorDiagonalElem) as e:
        sa.GaussSeidel.new(B).solve(rhs, x, 10, 1e-8)
    ref = oracle.gauss_seidel(indptr, indices, bad, rhs, x0, 10, 1e-8)
    assert e.value.row == row == ref.its and ref.status == oracle.ZERO_DIAG
    assert np.array_equal(x, ref.x)
    # the reference generator's own quirk: a non-square grid has missing diagonals (row 7 of a 6x7 grid)
    ip, ix, d = gen.grid_laplacian_dirichlet(6, 7)
    Q = sa.HipCsr.new((42, 42), ip, ix, d)
    with pytest.raises(sa.error.ZeorDiagonalElem) as e:
        sa.GaussSeidel.new(Q).solve(gen.dirichlet_rhs(6, 7), np.zeros(42), 10, 0.0)
    assert e.value.row == 7
