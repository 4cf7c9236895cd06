"""Shared plumbing of the three solver mirrors."""
import ctypes as C

import numpy as np

from . import _lib
from .device import dev_len, dev_ptr, is_device_array, pre_sync
from .error import check, solve_result


class _SolverBase:
    KIND = None       # _lib.SOLVER_*
    NAME = None       # "bicgstab" | "minres" | "csminres"

    def __init__(self, A, size):
        self.A = A                      # borrowed, like `A: &'data M` (bicg_stab.rs:18)
        self.size = int(size)
        self.dtype = A.dtype
        from .device import sfx
        self.s = sfx(self.dtype)
        h = C.c_void_p()
        st = getattr(_lib.lib(), "sprs_%s_create_%s" % (self.NAME, self.s))(A.h, self.size, C.byref(h))
        check(st, A.ctx.h)
        self.h = h
        self._trace = None

    @classmethod
    def new(cls, A, size):
        return cls(A, size)

    # ---- options (no reference analogue; instrumentation of this backend)
    def set_mode(self, mode):
        """'fused' (default) or 'literal' (one kernel per reference op, host-consumed scalars)."""
        m = {"fused": 0, "literal": 1}[mode] if isinstance(mode, str) else int(mode)
        check(_lib.lib().sprs_solver_set_mode(self.h, self.KIND, m), self.A.ctx.h)

    def set_trace(self, capacity_rows):
        if capacity_rows:
            self._trace = np.zeros((int(capacity_rows), 8))
            check(_lib.lib().sprs_solver_set_trace(self.h, self.KIND, self._trace.ctypes.data_as(C.c_void_p),
                                                   int(capacity_rows)), self.A.ctx.h)
        else:
            self._trace = None
            check(_lib.lib().sprs_solver_set_trace(self.h, self.KIND, None, 0), self.A.ctx.h)

    def trace(self):
        rows = C.c_size_t()
        check(_lib.lib().sprs_solver_trace_rows(self.h, self.KIND, C.byref(rows)), self.A.ctx.h)
        return self._trace[: rows.value].copy() if self._trace is not None else np.zeros((0, 8))

    def set_profile(self, enable=True):
        """False / 0: off; True / 1: HIP events around every SpMV launch; k >= 2: around one pair of consecutive launches in k."""
        check(_lib.lib().sprs_solver_set_profile(self.h, self.KIND, int(enable)), self.A.ctx.h)

    def profile(self):
        ms = C.c_double(); n = C.c_int64(); tot = C.c_double()
        check(_lib.lib().sprs_solver_get_profile(self.h, self.KIND, C.byref(ms), C.byref(n), C.byref(tot)), self.A.ctx.h)
        k2 = C.c_int64(); k4 = C.c_int64()
        check(_lib.lib().sprs_solver_get_fused_launches(self.h, self.KIND, C.byref(k2), C.byref(k4)), self.A.ctx.h)
        st = C.c_int64(); do = C.c_int64(); t2 = C.c_int64(); t4 = C.c_int64()
        check(_lib.lib().sprs_solver_get_profile_counts(self.h, self.KIND, C.byref(st), C.byref(do), C.byref(t2), C.byref(t4)), self.A.ctx.h)
        return dict(spmv_ms_total=ms.value, spmv_launches=n.value, solve_ms=tot.value, fused_k2=k2.value, fused_k4=k4.value,
                    steps=st.value, timed_dot_other=do.value, timed_fused_k2=t2.value, timed_fused_k4=t4.value)

    # ---- the call itself
    def _solve(self, precond, rhs, x, max_iter, tol, want_precond):
        L = _lib.lib()
        its = C.c_size_t(0); res = _lib.REAL[self.s](0.0)
        dev = is_device_array(rhs)
        if dev != is_device_array(x):
            raise TypeError("rhs and x must both be host arrays or both be device vectors")
        if dev:
            pre_sync(rhs, x)
            rp, rl, xp, xl = dev_ptr(rhs), dev_len(rhs), dev_ptr(x), dev_len(x)
            if self.NAME == "csminres":
                st = getattr(L, "sprs_csminres_solve_dev_" + self.s)(self.h, rp, rl, xp, xl, int(max_iter), float(tol),
                                                                  C.byref(its), C.byref(res))
            else:
                st = getattr(L, "sprs_%s_solve_dev_%s" % (self.NAME, self.s))(
                    self.h, precond.h if precond is not None else None, rp, rl, xp, xl, int(max_iter), float(tol),
                    C.byref(its), C.byref(res))
        else:
            rhs_a = np.ascontiguousarray(rhs, dtype=self.dtype)
            if not (isinstance(x, np.ndarray) and x.dtype == self.dtype and x.flags.c_contiguous):
                raise TypeError("x must be a contiguous %s ndarray (it is updated in place)" % self.dtype)
            rp, xp = rhs_a.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p)
            if want_precond:
                st = getattr(L, "sprs_%s_precond_solve_%s" % (self.NAME, self.s))(
                    self.h, precond.h, rp, rhs_a.size, xp, x.size, int(max_iter), float(tol), C.byref(its), C.byref(res))
            else:
                st = getattr(L, "sprs_%s_solve_%s" % (self.NAME, self.s))(
                    self.h, rp, rhs_a.size, xp, x.size, int(max_iter), float(tol), C.byref(its), C.byref(res))
        return solve_result(st, its.value, res.value, self.A.ctx.h)

    def close(self):
        if self.h:
            getattr(_lib.lib(), "sprs_%s_destroy" % self.NAME)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
