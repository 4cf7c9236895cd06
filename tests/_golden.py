"""Helpers shared by the oracle tests and the GPU parity tests: load tests/golden/*.json and
materialise their inputs as numpy arrays."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def cplx(a):
    a = np.asarray(a, dtype=np.float64)
    return (a[..., 0] + 1j * a[..., 1]).astype(np.complex128)


NP = {"f64": np.float64, "c64": np.complex128, "f32": np.float32, "c32": np.complex64}


def vec(spec, dtype):
    """{"fill": v, "n": n} or {"array": [...]} -> ndarray of dtype 'f64' / 'c64' / 'f32' / 'c32'."""
    npd = NP[dtype]
    if "fill" in spec:
        v = spec["fill"]
        if dtype in ("c64", "c32"):
            v = complex(*v) if isinstance(v, list) else complex(v)
            return np.full(spec["n"], v, dtype=npd)
        return np.full(spec["n"], float(v), dtype=npd)
    a = spec["array"]
    if dtype in ("c64", "c32"):
        return cplx(a).astype(npd)
    return np.asarray(a, dtype=npd)


def scalar(v, dtype):
    if dtype in ("c64", "c32"):
        return complex(*v) if isinstance(v, list) else complex(v)
    return float(v)


def spmv_case(case):
    dt = np.complex128 if case.get("complex") else np.float64
    idt = np.dtype(case.get("index_dtype", "int64"))
    indptr = np.asarray(case["indptr"], dtype=idt)
    indices = np.asarray(case["indices"], dtype=idt)
    if case.get("complex"):
        data, x, exp = cplx(case["data"]), cplx(case["x"]), cplx(case["expected"])
    else:
        data, x, exp = (np.asarray(case[k], dtype=dt) for k in ("data", "x", "expected"))
    return indptr, indices, data, x, exp


def solver_problem(case):
    """-> dict(indptr, indices, data, rhs, diag, exact) for a tests/golden/solver_kat.json case."""
    from sprsolve_amd import gen
    rows, cols = case["shape"]
    g = case["gen"]
    diag = None
    if g == "grid_laplacian_dirichlet":
        indptr, indices, data = gen.grid_laplacian_dirichlet(rows, cols)
        rhs = gen.dirichlet_rhs(rows, cols)
    elif g == "minres_grid_laplacian":
        indptr, indices, data, rhs = gen.minres_grid_laplacian(rows, cols)
    elif g == "minres_simple_diag":
        indptr, indices, data, rhs = gen.minres_simple_diag(rows, cols)
    elif g == "complex_hermitian_grid":
        indptr, indices, data, rhs, diag = gen.complex_hermitian_grid(rows, cols)
    elif g == "complex_symmetric_grid":
        indptr, indices, data, rhs, diag = gen.complex_symmetric_grid(rows, cols)
    else:
        raise KeyError(g)
    i, j = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    ex = case["exact"]
    if ex == "i+j":
        exact = (i + j).ravel().astype(data.dtype)
    elif ex == "0.5":
        exact = np.full(rows * cols, 0.5)
    elif ex == "i+j*1i":
        exact = (i + 1j * j).ravel().astype(np.complex128)
    else:
        raise KeyError(ex)
    if case["precond"] is None:
        diag = None
    return dict(indptr=indptr, indices=indices, data=data, rhs=rhs, diag=diag, exact=exact)


def gs_problem(case):
    """-> dict(indptr, indices, data, rhs, exact) for a tests/golden/gs_kat.json case (dtype applied)."""
    from sprsolve_amd import gen
    rows, cols = case["shape"]
    dt = np.float64 if case["dtype"] == "f64" else np.float32
    indptr, indices, data = gen.grid_laplacian_dirichlet(rows, cols)
    rhs = gen.dirichlet_rhs(rows, cols)
    i, j = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    return dict(indptr=indptr, indices=indices, data=data.astype(dt), rhs=rhs.astype(dt), exact=(i + j).ravel().astype(dt))
