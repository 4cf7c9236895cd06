"""TEST INFRASTRUCTURE ONLY — ctypes front end of the CPU oracle (oracle/sprs_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker.  The product (sprsolve_amd/) never does.

Every function takes / returns numpy arrays (float64 or complex128; complex128 is
layout-identical to the reference's ``Complex<f64>`` = {re, im}).  Index arrays are widened
to int64, the reference's ``usize`` (src/mat.rs:199).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsprs_oracle.so")

OK, INCOMPATIBLE_RHS, INCOMPATIBLE_X, INSUFFICIENT_ITER, BREAKDOWN, INVALID_PRECOND, DIM_MISMATCH = range(7)
ZERO_DIAG = 8


def build(force=False):
    """Compile the C restatement (gcc). Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("sprs_oracle.c", "krylov_tmpl.h", "scalar.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_norm2_d.restype = C.c_double
        _lib.orc_norm2_z.restype = C.c_double
        _lib.orc_dot_d.restype = C.c_double
        _lib.orc_conj_dot_d.restype = C.c_double
        _lib.orc_spmv_csr_dot_d.restype = C.c_double
        _lib.orc_dot_z.restype = _C64
        _lib.orc_conj_dot_z.restype = _C64
        _lib.orc_spmv_csr_dot_z.restype = _C64
        _lib.orc_max_threads.restype = C.c_int
        _lib.orc_norm2_s.restype = C.c_float
        _lib.orc_norm2_c.restype = C.c_float
        _lib.orc_dot_s.restype = C.c_float
        _lib.orc_conj_dot_s.restype = C.c_float
        _lib.orc_spmv_csr_dot_s.restype = C.c_float
        _lib.orc_dot_c.restype = _C32
        _lib.orc_conj_dot_c.restype = _C32
        _lib.orc_spmv_csr_dot_c.restype = _C32
        for sfx, rt, ct in (("d", C.c_double, C.c_double), ("z", C.c_double, _C64), ("s", C.c_float, C.c_float), ("c", C.c_float, _C32)):
            getattr(_lib, "orc_norm2_gpu_order_" + sfx).restype = rt
            getattr(_lib, "orc_conj_dot_gpu_order_" + sfx).restype = ct
    return _lib


class _C64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class _C32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


def set_threads(n):
    lib().orc_set_threads(C.c_int(int(n)))


def max_threads():
    return int(lib().orc_max_threads())


def set_reduction_order(mode, grid=512):
    """Order in which the SOLVERS add the terms of their dot products / norms: "reference" = the serial left folds of
    vecalg.rs:563-568,601-605 (default); "gpu" = the same terms in the order of libsprsolve_hip's stand-alone reduction
    kernels with at most `grid` workgroups (the context knob "grid": 2 per CU).  The library's literal solver mode must
    then reproduce the oracle's recurrence bit for bit."""
    lib().orc_set_reduction_order(C.c_int({"reference": 0, "gpu": 1}[mode]), C.c_int(int(grid)))


def conj_dot_gpu_order(x, y):
    s = _sfx(x.dtype); x = _arr(x, x.dtype); y = _arr(y, x.dtype)
    return _ret(getattr(lib(), "orc_conj_dot_gpu_order_" + s)(C.c_int64(x.size), _p(x), _p(y)), s)


def norm2_gpu_order(x):
    s = _sfx(x.dtype); x = _arr(x, x.dtype)
    return float(getattr(lib(), "orc_norm2_gpu_order_" + s)(C.c_int64(x.size), _p(x)))


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "d"
    if dtype == np.complex128:
        return "z"
    if dtype == np.float32:
        return "s"
    if dtype == np.complex64:
        return "c"
    raise TypeError("oracle supports float32/64 and complex64/128, got %s" % dtype)


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _scalar(v, sfx):
    if sfx == "d":
        return C.c_double(float(v))
    if sfx == "s":
        return C.c_float(float(v))
    v = complex(v)
    return _C64(v.real, v.imag) if sfx == "z" else _C32(v.real, v.imag)


def _real(v, sfx):
    return C.c_double(float(v)) if sfx in "dz" else C.c_float(float(v))


def _ret(v, sfx):
    return float(v) if sfx in "ds" else complex(v.re, v.im)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# ----------------------------------------------------------------------------- vecalg
def dot(x, y):
    s = _sfx(x.dtype); x = _arr(x, x.dtype); y = _arr(y, x.dtype)
    assert x.shape == y.shape
    return _ret(getattr(lib(), "orc_dot_" + s)(C.c_int64(x.size), _p(x), _p(y)), s)


def conj_dot(x, y):
    s = _sfx(x.dtype); x = _arr(x, x.dtype); y = _arr(y, x.dtype)
    assert x.shape == y.shape
    return _ret(getattr(lib(), "orc_conj_dot_" + s)(C.c_int64(x.size), _p(x), _p(y)), s)


def norm2(x):
    s = _sfx(x.dtype); x = _arr(x, x.dtype)
    return float(getattr(lib(), "orc_norm2_" + s)(C.c_int64(x.size), _p(x)))


def axpy(a, x, y):
    """y += x * a  (in place on y; y must be a contiguous ndarray)."""
    s = _sfx(y.dtype); x = _arr(x, y.dtype)
    assert y.flags.c_contiguous and x.shape == y.shape
    if s in "zc" and isinstance(a, (float, int, np.floating)):
        if s == "z":
            lib().orc_axpy_zd(C.c_int64(y.size), C.c_double(float(a)), _p(x), _p(y))
        else:
            lib().orc_axpy_cs(C.c_int64(y.size), C.c_float(float(a)), _p(x), _p(y))
    else:
        getattr(lib(), "orc_axpy_" + s)(C.c_int64(y.size), _scalar(a, s), _p(x), _p(y))
    return y


def axpby(a, x, b, y):
    """y = x * a + y * b  (in place on y)."""
    s = _sfx(y.dtype); x = _arr(x, y.dtype)
    assert y.flags.c_contiguous and x.shape == y.shape
    getattr(lib(), "orc_axpby_" + s)(C.c_int64(y.size), _scalar(a, s), _p(x), _scalar(b, s), _p(y))
    return y


def scale(a, v):
    s = _sfx(v.dtype); assert v.flags.c_contiguous
    getattr(lib(), "orc_scale_" + s)(C.c_int64(v.size), _scalar(a, s), _p(v))
    return v


def rscale(a, v):
    s = _sfx(v.dtype); assert v.flags.c_contiguous
    getattr(lib(), "orc_rscale_" + s)(C.c_int64(v.size), _real(a, s), _p(v))
    return v


def conj(x):
    s = _sfx(x.dtype); x = _arr(x, x.dtype); out = np.empty_like(x)
    getattr(lib(), "orc_conj_" + s)(C.c_int64(x.size), _p(x), _p(out))
    return out


# ----------------------------------------------------------------------------- mat / precond
def spmv(indptr, indices, data, x, parallel=False):
    s = _sfx(data.dtype); data = _arr(data, data.dtype); x = _arr(x, data.dtype)
    indptr = _i64(indptr); indices = _i64(indices)
    n = indptr.size - 1
    y = np.empty(n, dtype=data.dtype)
    getattr(lib(), "orc_spmv_csr_" + s)(C.c_int64(n), _p(indptr), _p(indices), _p(data), _p(x), _p(y),
                                        C.c_int(1 if parallel else 0))
    return y


def spmv_csc(nrows, indptr, indices, data, x):
    s = _sfx(data.dtype); data = _arr(data, data.dtype); x = _arr(x, data.dtype)
    indptr = _i64(indptr); indices = _i64(indices)
    y = np.empty(nrows, dtype=data.dtype)
    getattr(lib(), "orc_spmv_csc_" + s)(C.c_int64(nrows), C.c_int64(indptr.size - 1), _p(indptr), _p(indices),
                                        _p(data), _p(x), _p(y))
    return y


def spmv_dot(indptr, indices, data, x, parallel=False):
    s = _sfx(data.dtype); data = _arr(data, data.dtype); x = _arr(x, data.dtype)
    indptr = _i64(indptr); indices = _i64(indices)
    n = indptr.size - 1
    y = np.empty(n, dtype=data.dtype)
    r = getattr(lib(), "orc_spmv_csr_dot_" + s)(C.c_int64(n), _p(indptr), _p(indices), _p(data), _p(x), _p(y),
                                                C.c_int(1 if parallel else 0))
    return y, _ret(r, s)


def diag_inv(diag):
    """DiagPrecond::new (src/precond.rs:20-29): 1/diag, real or complex V."""
    diag = np.ascontiguousarray(diag)
    out = np.empty_like(diag)
    if diag.dtype == np.float64:
        lib().orc_diag_inv_real_d(C.c_int64(diag.size), _p(diag), _p(out))
    elif diag.dtype == np.float32:
        lib().orc_diag_inv_real_s(C.c_int64(diag.size), _p(diag), _p(out))
    elif diag.dtype == np.complex128:
        lib().orc_diag_inv_complex(C.c_int64(diag.size), _p(diag), _p(out))
    elif diag.dtype == np.complex64:
        lib().orc_diag_inv_complex_f(C.c_int64(diag.size), _p(diag), _p(out))
    else:
        raise TypeError(diag.dtype)
    return out


def diag_apply(dinv, v):
    s = _sfx(v.dtype); v = _arr(v, v.dtype); dinv = np.ascontiguousarray(dinv)
    out = np.empty_like(v)
    getattr(lib(), "orc_diag_apply_" + s)(C.c_int64(v.size), _p(dinv), C.c_int(int(dinv.dtype.kind == 'c')),
                                          _p(v), _p(out))
    return out


# ----------------------------------------------------------------------------- solvers
class Result:
    def __init__(self, status, its, res, x, trace):
        self.status, self.its, self.res, self.x, self.trace = status, its, res, x, trace

    def __repr__(self):
        return "Result(status=%d, its=%d, res=%.3e)" % (self.status, self.its, self.res)


def _solve(kind, indptr, indices, data, rhs, x0, max_iter, tol, precond_diag=None, parallel=False,
           trace_cap=0, size=None):
    s = _sfx(data.dtype)
    data = _arr(data, data.dtype); rhs = _arr(rhs, data.dtype)
    x = np.array(x0, dtype=data.dtype, copy=True)
    indptr = _i64(indptr); indices = _i64(indices)
    n = indptr.size - 1 if size is None else size
    work = np.zeros(8 * max(n, rhs.size), dtype=data.dtype)
    its = C.c_int64(0); res = (C.c_double if s in "dz" else C.c_float)(0.0); rows = C.c_int64(0)
    trace = np.zeros((max(trace_cap, 1), 8))
    pc = None; pc_c = 0
    if precond_diag is not None:
        pc = diag_inv(np.ascontiguousarray(precond_diag)); pc_c = int(pc.dtype.kind == 'c')
    pc_p = _p(pc) if pc is not None else C.c_void_p(0)
    common = [C.c_int64(n), _p(indptr), _p(indices), _p(data), C.c_int(1 if parallel else 0), pc_p, C.c_int(pc_c),
              _p(rhs), C.c_int64(rhs.size), _p(x), C.c_int64(x.size), C.c_int64(max_iter), _real(tol, s),
              _p(work), C.byref(its), C.byref(res), _p(trace), C.c_int64(trace_cap), C.byref(rows)]
    if kind == "bicgstab":
        st = getattr(lib(), "orc_bicgstab_" + s)(*common)
    else:
        st = getattr(lib(), "orc_minres_" + s)(C.c_int(1 if kind == "csminres" else 0), *common)
    return Result(int(st), int(its.value), float(res.value), x, trace[: rows.value].copy())


def bicgstab(indptr, indices, data, rhs, x0, max_iter, tol, precond_diag=None, **kw):
    """BiCGStab::solve / ::precond_solve (src/bicg_stab.rs:35-200, :204-366)."""
    return _solve("bicgstab", indptr, indices, data, rhs, x0, max_iter, tol, precond_diag, **kw)


def minres(indptr, indices, data, rhs, x0, max_iter, tol, precond_diag=None, **kw):
    """MinRes::solve / ::precond_solve (src/minres.rs:31-172, :178-341)."""
    return _solve("minres", indptr, indices, data, rhs, x0, max_iter, tol, precond_diag, **kw)


def csminres(indptr, indices, data, rhs, x0, max_iter, tol, **kw):
    """CSMinRes::solve (src/cs_minres.rs:29-158)."""
    return _solve("csminres", indptr, indices, data, rhs, x0, max_iter, tol, None, **kw)


def gauss_seidel(indptr, indices, data, rhs, x0, max_iter, eps):
    """GaussSeidel::solve (src/gauss_seidel.rs:33-140), real scalars.  Result.res is the ABSOLUTE residual norm the
    reference returns; Result.its is the row index when status == ZERO_DIAG."""
    s = _sfx(data.dtype)
    assert s in "ds", "the reference bounds GaussSeidel to T: PartialOrd (real scalars)"
    data = _arr(data, data.dtype); rhs = _arr(rhs, data.dtype)
    x = np.array(x0, dtype=data.dtype, copy=True)
    indptr = _i64(indptr); indices = _i64(indices)
    n = indptr.size - 1
    work = np.zeros(2 * max(n, 1), dtype=data.dtype)
    its = C.c_int64(0); res = (C.c_double if s == "d" else C.c_float)(0.0)
    st = getattr(lib(), "orc_gauss_seidel_" + s)(C.c_int64(n), _p(indptr), _p(indices), _p(data), _p(rhs), C.c_int64(rhs.size),
                                                  _p(x), C.c_int64(x.size), C.c_int64(max_iter), _real(eps, s), _p(work),
                                                  C.byref(its), C.byref(res))
    return Result(int(st), int(its.value), float(res.value), x, np.zeros((0, 8)))
