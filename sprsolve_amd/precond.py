"""Jacobi preconditioner — mirror of the reference's src/precond.rs `DiagPrecond<T, V>`."""
import ctypes as C

import numpy as np

from . import _lib
from .device import default_ctx, dev_len, dev_ptr, is_device_array, pre_sync
from .error import check
from .mat import MatVecMul


class DiagPrecond(MatVecMul):
    """precond.rs:6-63.  `T` is the vector scalar, `V` the diagonal's (real V with complex T is
    the reference's `DiagPrecond<Complex64, f64>`, tests/test_complex_solve.rs:44)."""

    def __init__(self, handle, ctx, t_dtype, n):
        self.h, self.ctx, self.dtype, self.n = handle, ctx, np.dtype(t_dtype), n

    @classmethod
    def new(cls, diag, t_dtype=None, ctx=None):
        """precond.rs:20-29: stores 1/diag (no zero check, as upstream)."""
        ctx = ctx or default_ctx()
        diag = np.ascontiguousarray(diag)
        if diag.dtype in (np.complex128, np.complex64):
            name, t = "sprs_diag_precond_create_" + ("z" if diag.dtype == np.complex128 else "c"), diag.dtype
        elif diag.dtype in (np.float64, np.float32):
            t = np.dtype(t_dtype or diag.dtype)
            cx = t in (np.complex128, np.complex64)
            if (t in (np.complex64, np.float32)) != (diag.dtype == np.float32):
                raise TypeError("diagonal and vector precision must match (T: Mul<V>)")
            name = "sprs_diag_precond_create_" + {(np.float64, False): "d", (np.float64, True): "zd",
                                                  (np.float32, False): "s", (np.float32, True): "cs"}[(diag.dtype.type, cx)]
        else:
            raise TypeError(diag.dtype)
        h = C.c_void_p()
        check(getattr(_lib.lib(), name)(ctx.h, diag.size, diag.ctypes.data_as(C.c_void_p), C.byref(h)), ctx.h)
        return cls(h, ctx, t, diag.size)

    def _s(self):
        from .device import sfx
        return sfx(self.dtype)

    def mul_vec(self, v_in, v_out):
        """precond.rs:37-45 (checked)."""
        if is_device_array(v_in):
            if self.n != dev_len(v_in) or self.n != dev_len(v_out):
                from .error import DimensionMismatch
                raise DimensionMismatch("Dimension mismatch")
            return self.mul_vec_unchecked(v_in, v_out)
        x = np.ascontiguousarray(v_in, dtype=self.dtype)
        assert isinstance(v_out, np.ndarray) and v_out.dtype == self.dtype
        check(getattr(_lib.lib(), "sprs_diag_mul_vec_" + self._s())(
            self.h, x.ctypes.data_as(C.c_void_p), x.size, v_out.ctypes.data_as(C.c_void_p), v_out.size), self.ctx.h)

    def mul_vec_unchecked(self, v_in, v_out):
        """precond.rs:48-52 on device vectors."""
        pre_sync(v_in, v_out)
        check(getattr(_lib.lib(), "sprs_diag_mul_vec_dev_" + self._s())(self.h, dev_ptr(v_in), dev_ptr(v_out)), self.ctx.h)
        self.ctx.sync()

    def mul_vec_dot(self, v_in, v_out):
        raise NotImplementedError("unimplemented!() upstream (precond.rs:55-57)")

    def mul_vec_dot_unchecked(self, v_in, v_out):
        raise NotImplementedError("unimplemented!() upstream (precond.rs:60-62)")

    def close(self):
        if self.h:
            _lib.lib().sprs_diag_precond_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
