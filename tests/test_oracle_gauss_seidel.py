"""The oracle's Gauss-Seidel (oracle/krylov_tmpl.h: orc_gauss_seidel_*) pinned on the reference's own
test (tests/test_solvers.rs:2-31) and on an independent pure-Python restatement of gauss_seidel.rs."""
import numpy as np
import pytest

import _golden as G

CASES = G.load("gs_kat.json")["cases"]


def py_gauss_seidel(indptr, indices, data, rhs, x, max_iter, eps):
    """Pure-Python loops following gauss_seidel.rs:33-140 line by line (small cases only)."""
    dt = data.dtype.type
    n = len(indptr) - 1
    x = x.copy()
    if max_iter == 0:
        return "insufficient", max_iter, None, x
    diag = np.zeros(n, dtype=data.dtype)
    b2 = dt(0)
    for row in range(n):
        sigma, dg = dt(0), dt(0)
        for k in range(indptr[row], indptr[row + 1]):
            c = indices[k]
            if c != row:
                sigma = dt(sigma + dt(data[k] * x[c]))
            else:
                dg = data[k]
        if dt(dg * dg) < np.finfo(dt).eps:
            return "zero_diag", row, None, x
        diag[row] = dg
        b2 = dt(b2 + dt(rhs[row] * rhs[row]))
        x[row] = dt(dt(rhs[row] - sigma) / dg)
    tol2 = dt(dt(eps) * np.sqrt(b2))

    def resid():
        s = dt(0)
        for row in range(n):
            acc = dt(0)
            for k in range(indptr[row], indptr[row + 1]):
                acc = dt(acc + dt(data[k] * x[indices[k]]))
            r = dt(acc - rhs[row])
            s = dt(s + dt(r * r))
        return np.sqrt(s)
    res = resid()
    if res <= tol2:
        return "ok", 1, res, x
    for it in range(1, max_iter):
        for row in range(n):
            sigma = dt(0)
            for k in range(indptr[row], indptr[row + 1]):
                c = indices[k]
                if c != row:
                    sigma = dt(sigma + dt(data[k] * x[c]))
            x[row] = dt(dt(rhs[row] - sigma) / diag[row])
        res = resid()
        if res <= tol2:
            return "ok", it, res, x
    return "insufficient", max_iter, None, x


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_gs_kat(oracle, case):
    p = G.gs_problem(case)
    r = oracle.gauss_seidel(p["indptr"], p["indices"], p["data"], p["rhs"], np.zeros_like(p["rhs"]), case["max_iter"], case["eps"])
    if case.get("expect") == "InsufficientIterNum":
        assert r.status == oracle.INSUFFICIENT_ITER and r.its == case["max_iter"]
        return
    assert r.status == oracle.OK, r
    if "oracle_its" in case:
        # the reference's test unwraps Ok with eps = 0: the residual must become exactly zero
        assert r.its == case["oracle_its"] and r.res == case["oracle_res"]
        assert np.array_equal(r.x, p["exact"])
    tol = 1e-6 if case["dtype"] == "f64" else 2e-3
    assert np.max(np.abs(r.x - p["exact"])) / np.max(p["exact"]) < tol
    ax = oracle.spmv(p["indptr"], p["indices"], p["data"], r.x)
    assert np.linalg.norm(ax - p["rhs"]) <= 1.01 * case["eps"] * np.linalg.norm(p["rhs"]) + 1e-30
    assert np.isclose(r.res, np.linalg.norm(ax - p["rhs"]), rtol=1e-3, atol=1e-30)      # ABSOLUTE residual (:107,136)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_gs_matches_python_restatement(oracle, dtype):
    from sprsolve_amd import gen
    indptr, indices, data = gen.grid_laplacian_dirichlet(6, 6)
    rhs = gen.dirichlet_rhs(6, 6).astype(dtype); data = data.astype(dtype)
    rng = np.random.default_rng(3)
    x0 = rng.uniform(-1, 1, rhs.size).astype(dtype)
    for max_iter, eps in [(0, 0.0), (1, 0.0), (7, 0.0), (400, 1e-4)]:
        kind, its, res, x = py_gauss_seidel(indptr, indices, data, rhs, x0, max_iter, eps)
        r = oracle.gauss_seidel(indptr, indices, data, rhs, x0, max_iter, eps)
        assert {"ok": oracle.OK, "insufficient": oracle.INSUFFICIENT_ITER}[kind] == r.status
        assert r.its == its
        assert np.array_equal(r.x, x)                   # bit-identical iterates
        if kind == "ok":
            assert r.res == float(res)


def test_gs_zero_diagonal(oracle):
    """ZeorDiagonalElem(row) after rows < row were already updated (gauss_seidel.rs:72-78)."""
    from sprsolve_amd import gen
    indptr, indices, data = gen.grid_laplacian_dirichlet(5, 5)
    rhs = gen.dirichlet_rhs(5, 5)
    data = data.copy()
    row = 12                                            # an interior row: zero its diagonal
    for k in range(indptr[row], indptr[row + 1]):
        if indices[k] == row:
            data[k] = 1e-9                              # |d|^2 < eps
    x0 = np.full(rhs.size, 0.25)
    r = oracle.gauss_seidel(indptr, indices, data, rhs, x0, 10, 1e-8)
    assert r.status == oracle.ZERO_DIAG and r.its == row
    kind, prow, _, x = py_gauss_seidel(indptr, indices, data, rhs, x0, 10, 1e-8)
    assert kind == "zero_diag" and prow == row
    assert np.array_equal(r.x, x)
    assert np.array_equal(r.x[row:], x0[row:]) and not np.array_equal(r.x[:row], x0[:row])
