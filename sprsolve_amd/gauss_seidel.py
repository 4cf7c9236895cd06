"""`GaussSeidel<'data, T, I>` of the reference (src/gauss_seidel.rs:8-141), on the GPU.

    gs = GaussSeidel.new(A)                      # gauss_seidel.rs:13-31
    its, res = gs.solve(rhs, x, max_iter, eps)   # gauss_seidel.rs:33-140; x is updated in place

Real scalars only (the reference bounds `T: PartialOrd`).  `res` is the ABSOLUTE residual norm
||A x - b||, compared against eps * ||b|| (gauss_seidel.rs:87,106,135) — unlike the Krylov solvers,
which return the relative one.  Raises IncompatibleMatrixFormat("Not a square matrix" / "Not in CSR
format") from `new`, ZeorDiagonalElem(row) / InsufficientIterNum(max_iter) from `solve`.
The sweep is level-scheduled on the device (csrc/gs.hip) and bit-identical to the serial one.
"""
import ctypes as C

import numpy as np

from . import _lib
from .device import dev_len, dev_ptr, is_device_array, pre_sync, sfx
from .error import check, solve_result


class GaussSeidel:
    def __init__(self, A):
        self.A = A                       # borrowed, like `a: CsMatViewI<'data, T, I>` (gauss_seidel.rs:9)
        self.dtype = A.dtype
        self.s = sfx(self.dtype)
        if self.s not in ("d", "s"):
            raise TypeError("GaussSeidel needs a real scalar type (T: PartialOrd, gauss_seidel.rs:8)")
        h = C.c_void_p()
        check(_lib.lib().sprs_gauss_seidel_create(A.h, C.byref(h)), A.ctx.h)
        self.h = h

    @classmethod
    def new(cls, A):
        return cls(A)

    @property
    def levels(self):
        """Dependency levels of the sweep = kernel launches per sweep (backend detail)."""
        return int(_lib.lib().sprs_gauss_seidel_levels(self.h))

    def solve(self, rhs, x, max_iter, eps):
        L = _lib.lib()
        its = C.c_size_t(0); res = _lib.REAL[self.s](0.0)
        dev = is_device_array(rhs)
        if dev != is_device_array(x):
            raise TypeError("rhs and x must both be host arrays or both be device vectors")
        if dev:
            pre_sync(rhs, x)
            st = getattr(L, "sprs_gauss_seidel_solve_dev_" + self.s)(
                self.h, dev_ptr(rhs), dev_len(rhs), dev_ptr(x), dev_len(x), int(max_iter), float(eps),
                C.byref(its), C.byref(res))
        else:
            rhs_a = np.ascontiguousarray(rhs, dtype=self.dtype)
            if not (isinstance(x, np.ndarray) and x.dtype == self.dtype and x.flags.c_contiguous):
                raise TypeError("x must be a contiguous %s ndarray (it is updated in place)" % self.dtype)
            st = getattr(L, "sprs_gauss_seidel_solve_" + self.s)(
                self.h, rhs_a.ctypes.data_as(C.c_void_p), rhs_a.size, x.ctypes.data_as(C.c_void_p), x.size,
                int(max_iter), float(eps), C.byref(its), C.byref(res))
        return solve_result(st, its.value, res.value, self.A.ctx.h)

    def close(self):
        if self.h:
            _lib.lib().sprs_gauss_seidel_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
