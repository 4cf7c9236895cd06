"""The reference shares the operator between threads (`BiCGStab<T: Send + Sync>` holds `A: &M`, src/bicg_stab.rs:17-18;
the rayon SpMV itself fans out through a `Sync` pointer, src/mat.rs:156-161): concurrent `&self` calls of
`mul_vec` / `mul_vec_dot` on ONE handle must be safe.  Four host threads hammer one handle through the C ABI (ctypes
releases the GIL for the duration of the call) with different inputs; every result must be bit-identical to the oracle's
fold — a torn staging buffer or a shared-scratch race shows up as a wrong y or a wrong dot."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sa():
    import sprsolve_amd
    from sprsolve_amd import _lib
    _lib.lib()
    sprsolve_amd.default_ctx(0)
    return sprsolve_amd


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.mark.parametrize("dtype", [np.float64, np.complex128], ids=["f64", "c64"])
def test_concurrent_mul_vec_on_one_handle(sa, oracle, dtype):
    from sprsolve_amd import gen
    R = 48
    indptr, indices, data = gen.grid_laplacian_dirichlet(R, R)
    n = R * R
    if np.dtype(dtype).kind == "c":
        data = data * (1.0 - 0.25j)
    A = sa.HipCsr.new((n, n), indptr, indices, data.astype(dtype))
    NT, CALLS, NX = 4, 200, 8
    rng = np.random.default_rng(7)
    xs, refs, dots = [], [], []
    for t in range(NT):
        row = []
        for k in range(NX):
            x = rng.uniform(-1, 1, n).astype(dtype)
            if np.dtype(dtype).kind == "c":
                x = x + 1j * rng.uniform(-1, 1, n)
            row.append(x)
        xs.append(row)
        refs.append([oracle.spmv(indptr, indices, data.astype(dtype), x) for x in row])
        dots.append([oracle.conj_dot(x, y) for x, y in zip(row, refs[-1])])
    errors = []
    start = threading.Barrier(NT)

    def work(t):
        try:
            y = np.empty(n, dtype=dtype)
            start.wait()
            for c in range(CALLS):
                k = (c * 3 + t) % NX
                y[:] = 0
                if c % 2 == 0:
                    A.mul_vec(xs[t][k], y)
                else:
                    d = A.mul_vec_dot(xs[t][k], y)
                    e = dots[t][k]
                    if abs(d - e) > 1e-12 * max(1.0, abs(e)):
                        errors.append(("dot", t, c, d, e))
                if not np.array_equal(_bits(y), _bits(refs[t][k])):
                    errors.append(("y", t, c, int(np.sum(_bits(y) != _bits(refs[t][k])))))
        except Exception as ex:     # noqa: BLE001 - reported through the list, the thread must not die silently
            errors.append(("exception", t, repr(ex)))
    th = [threading.Thread(target=work, args=(t,)) for t in range(NT)]
    for h in th:
        h.start()
    for h in th:
        h.join()
    assert not errors, errors[:5]


def test_concurrent_solves_and_reductions_on_one_context(sa, oracle):
    """Two solver handles on the same operator + stand-alone reductions from other threads: every entry point that
    uses per-context scratch queues on the context's mutex; results equal the single-threaded ones bit for bit."""
    from sprsolve_amd import gen
    R = 40
    indptr, indices, data = gen.grid_laplacian_dirichlet(R, R)
    rhs = gen.dirichlet_rhs(R, R)
    n = R * R
    A = sa.HipCsr.new((n, n), indptr, indices, data)
    P = sa.DiagPrecond.new(np.where(np.diff(indptr) == 1, 1.0, -4.0))

    def solve_once():
        s = sa.BiCGStab.new(A, n)
        x = np.zeros(n)
        its, res = s.precond_solve(P, rhs, x, 5000, 1e-10)
        return its, res, x
    its0, res0, x0 = solve_once()
    v = np.random.default_rng(3).uniform(-1, 1, n)
    nrm0 = sa.vecalg.norm2(v)
    out, errors = {}, []

    def solver_thread(t):
        try:
            out[t] = solve_once()
        except Exception as ex:     # noqa: BLE001
            errors.append(repr(ex))

    def norm_thread():
        try:
            for _ in range(300):
                if sa.vecalg.norm2(v) != nrm0:
                    errors.append("norm2 changed under concurrency")
                    return
        except Exception as ex:     # noqa: BLE001
            errors.append(repr(ex))
    th = [threading.Thread(target=solver_thread, args=(t,)) for t in range(2)] + [threading.Thread(target=norm_thread) for _ in range(2)]
    for h in th:
        h.start()
    for h in th:
        h.join()
    assert not errors, errors
    for t in range(2):
        its, res, x = out[t]
        assert (its, res) == (its0, res0) and np.array_equal(_bits(x), _bits(x0))
