"""Pins the CPU oracle (oracle/sprs_oracle.c) against every known-answer test the reference
holds for the hot path.  CPU only."""
import numpy as np
import pytest

import _golden as G


@pytest.mark.parametrize("case", G.load("spmv_kat.json")["cases"], ids=lambda c: c["name"])
def test_spmv_kat(oracle, case):
    indptr, indices, data, x, exp = G.spmv_case(case)
    if case["storage"] == "CSC":
        y = oracle.spmv_csc(case["shape"][0], indptr, indices, data, x)
    else:
        y = oracle.spmv(indptr, indices, data, x)
        yp = oracle.spmv(indptr, indices, data, x, parallel=True)
        assert np.array_equal(y.view(np.float64), yp.view(np.float64)), "row-parallel bits differ from serial"
    assert np.all(np.abs(y - exp) < case["eps"] * (np.sqrt(2) if case.get("complex") else 1))
    if case.get("complex"):
        assert np.all(np.abs(y.real - exp.real) < case["eps"]) and np.all(np.abs(y.imag - exp.imag) < case["eps"])


def test_mul_vec_dot_kat(oracle):
    """src/mkl_mat.rs:432-463: mul_vec_dot(x, y) == conj_dot(x, A x)."""
    spec = G.load("spmv_kat.json")
    case = [c for c in spec["cases"] if c["name"] == spec["mul_vec_dot"]["case"]][0]
    indptr, indices, data, x, exp = G.spmv_case(case)
    y, d = oracle.spmv_dot(indptr, indices, data, x)
    assert np.all(np.abs(y - exp) < 2e-8)
    e = oracle.conj_dot(x, y)
    assert d == e
    assert abs(d - np.vdot(x, y)) < 1e-14


def _run_vecalg(orc, case):
    dt = case["dtype"]
    op = case["op"]
    x = G.vec(case["x"], dt)
    if op == "norm2":
        return orc.norm2(x)
    if op in ("dot", "conj_dot"):
        return getattr(orc, op)(x, G.vec(case["y"], dt))
    if op == "scale":
        return orc.scale(G.scalar(case["a"], dt), x)
    if op == "rscale":
        return orc.rscale(case["a"], x)
    if op == "conj":
        return orc.conj(x)
    if op == "axpy":
        y = G.vec(case["y"], dt)
        a = float(case["a_real"]) if "a_real" in case else G.scalar(case["a"], dt)
        return orc.axpy(a, x, y)
    if op == "axpy_repeat":
        y = G.vec(case["y"], dt)
        for _ in range(case["repeat"]):
            orc.axpy(G.scalar(case["a"], dt), x, y)
        return y
    raise KeyError(op)


@pytest.mark.parametrize("case", G.load("vecalg_kat.json")["cases"], ids=lambda c: c["name"])
def test_vecalg_kat(oracle, case):
    dt = case["dtype"]
    eps = case.get("eps", 1e-13)
    if case["op"] == "axpby_sequence":
        x = G.vec(case["x"], dt); y = G.vec(case["y"], dt)
        for st in case["steps"]:
            for _ in range(st["repeat"]):
                oracle.axpby(G.scalar(st["a"], dt), x, G.scalar(st["b"], dt), y)
            assert np.all(np.abs(y - st["expected_fill"]) <= eps)
        return
    got = _run_vecalg(oracle, case)
    if "expected" in case:
        assert abs(got - G.scalar(case["expected"], dt)) <= eps
    elif "expected_fill" in case:
        assert np.all(np.abs(got - G.scalar(case["expected_fill"], dt)) <= eps)
    else:
        exp = G.vec({"array": case["expected_array"]}, dt)
        assert np.all(np.abs(got - exp) <= eps)


@pytest.mark.parametrize("case", G.load("solver_kat.json")["cases"], ids=lambda c: c["name"])
def test_solver_kat(oracle, case):
    """The reference asserts only Ok; we additionally assert the exact solution its
    construction implies."""
    p = G.solver_problem(case)
    fn = getattr(oracle, case["solver"])
    x0 = np.zeros_like(p["rhs"])
    r = fn(p["indptr"], p["indices"], p["data"], p["rhs"], x0, case["max_iter"], case["tol"],
           precond_diag=p["diag"])
    assert r.status == oracle.OK, r
    err = np.max(np.abs(r.x - p["exact"]))
    scale = max(1.0, np.max(np.abs(p["exact"])))
    assert err / scale < 1e-9, (r, err)
    # true residual
    res = np.linalg.norm(oracle.spmv(p["indptr"], p["indices"], p["data"], r.x) - p["rhs"]) / np.linalg.norm(p["rhs"])
    assert res < 1e-10, res


@pytest.mark.parametrize("name", ["test_minres", "minres_ident"])
def test_csminres_is_minres_for_real_scalars(oracle, name):
    """CSMinRes has no test in the reference (tests/test_minres.rs:14-15 are commented out).  For REAL T `conj` is the
    identity, so cs_minres.rs:90-154 is arithmetically minres.rs:90-169 — same Lanczos products in the same order,
    same Givens formulas.  The oracle's two restatements must therefore agree BIT FOR BIT on the reference's own two
    MINRES problems (tests/test_minres.rs:1-60): that ties the Saunders code path to reference-held fixtures.  The
    complex-symmetric branch (conjugations live) stays "parity unpinned"."""
    case = [c for c in G.load("solver_kat.json")["cases"] if c["name"] == name][0]
    p = G.solver_problem(case)
    K = 40
    a = oracle.minres(p["indptr"], p["indices"], p["data"], p["rhs"], np.zeros_like(p["rhs"]), K, 0.0, trace_cap=K)
    b = oracle.csminres(p["indptr"], p["indices"], p["data"], p["rhs"], np.zeros_like(p["rhs"]), K, 0.0, trace_cap=K)
    assert a.status == b.status and a.its == b.its
    assert a.trace.shape == b.trace.shape and a.trace.shape[0] > 0
    assert np.array_equal(a.trace.view(np.uint64), b.trace.view(np.uint64))
    assert np.array_equal(a.x.view(np.uint64), b.x.view(np.uint64))
    # and to convergence with the reference's own tolerance
    a = oracle.minres(p["indptr"], p["indices"], p["data"], p["rhs"], np.zeros_like(p["rhs"]), case["max_iter"], case["tol"])
    b = oracle.csminres(p["indptr"], p["indices"], p["data"], p["rhs"], np.zeros_like(p["rhs"]), case["max_iter"], case["tol"])
    assert (a.status, a.its, a.res) == (b.status, b.its, b.res) and a.status == oracle.OK
    assert np.array_equal(a.x.view(np.uint64), b.x.view(np.uint64))


def test_gpu_order_reductions_are_a_reordering_only(oracle):
    """oracle/krylov_tmpl.h, "reductions as the SOLVERS call them": mode "gpu" adds the SAME terms in the order of the
    library's reduction kernels.  On the CPU this can only be checked as a reordering: sums of exactly representable
    terms are identical in any order; generic sums agree to rounding; one term or one workgroup's worth of terms folds
    exactly as the serial loop does for the first lane; the solvers reach the same solutions; the default is the
    reference's serial fold.  (Bit-for-bit equality with the GPU is tests/test_gpu_parity.py's business.)"""
    rng = np.random.default_rng(3)
    for dt in (np.float64, np.complex128, np.float32, np.complex64):
        for n in (1, 2, 255, 256, 257, 4099, 131075, 300001):
            x = rng.integers(-8, 9, n).astype(dt); y = rng.integers(-8, 9, n).astype(dt)
            if np.dtype(dt).kind == "c":
                x = x + 1j * rng.integers(-8, 9, n).astype(dt); y = y - 1j * rng.integers(-8, 9, n).astype(dt)
            assert oracle.conj_dot_gpu_order(x, y) == oracle.conj_dot(x, y), (dt, n)        # small integers: exact in any order
            assert oracle.norm2_gpu_order(x) == oracle.norm2(x)
            xr = rng.uniform(-1, 1, n).astype(dt); yr = rng.uniform(-1, 1, n).astype(dt)
            tol = (2e-4 if np.dtype(dt).itemsize in (4, 8) and np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-12) * max(1.0, float(np.sum(np.abs(xr * yr))))
            assert abs(oracle.conj_dot_gpu_order(xr, yr) - oracle.conj_dot(xr, yr)) <= tol
    # the solvers: same solution either way, and the mode is off unless asked for
    case = [c for c in G.load("solver_kat.json")["cases"] if c["name"] == "bench_laplacian_100"][0]
    p = G.solver_problem(case)
    a = oracle.bicgstab(p["indptr"], p["indices"], p["data"], p["rhs"], np.zeros_like(p["rhs"]), case["max_iter"], case["tol"], trace_cap=4)
    oracle.set_reduction_order("gpu", 512)
    try:
        b = oracle.bicgstab(p["indptr"], p["indices"], p["data"], p["rhs"], np.zeros_like(p["rhs"]), case["max_iter"], case["tol"], trace_cap=4)
    finally:
        oracle.set_reduction_order("reference")
    c = oracle.bicgstab(p["indptr"], p["indices"], p["data"], p["rhs"], np.zeros_like(p["rhs"]), case["max_iter"], case["tol"], trace_cap=4)
    assert a.status == b.status == oracle.OK and np.max(np.abs(a.x - b.x)) < 1e-9 and np.max(np.abs(b.x - p["exact"])) < 1e-9
    assert np.allclose(a.trace[0], b.trace[0], rtol=1e-12) and not np.array_equal(a.trace, b.trace)      # different order, same numbers
    assert np.array_equal(a.trace, c.trace) and np.array_equal(a.x, c.x) and a.its == c.its               # the default is untouched
