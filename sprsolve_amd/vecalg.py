"""BLAS-1 on the GPU — mirror of the reference's src/vecalg.rs public functions.

Arguments may be device vectors (DevVec / torch CUDA tensors: computed in place, nothing
crosses PCIe) or numpy arrays (uploaded, computed by the HIP kernel, downloaded — so the
reference's own unit tests can be replayed verbatim against the kernels).
"""
import ctypes as C

import numpy as np

from . import _lib
from .device import DevVec, default_ctx, dev_len, dev_ptr, dev_sfx, is_device_array, pre_sync
from .error import check


class _Staged:
    """Context manager: numpy array -> temporary DevVec (copied back on exit when `out`)."""

    def __init__(self, a, out=False, ctx=None):
        self.a, self.out, self.tmp = a, out, None
        self.ctx = ctx

    def __enter__(self):
        if is_device_array(self.a):
            pre_sync(self.a)
            return self.a
        if not isinstance(self.a, np.ndarray):
            self.a = np.asarray(self.a)
            if self.out:
                raise TypeError("output vector must be a numpy array or a device vector")
        self.tmp = DevVec.from_numpy(self.a, self.ctx)
        return self.tmp

    def __exit__(self, *exc):
        if self.tmp is not None:
            if self.out and exc[0] is None:
                self.a[...] = self.tmp.to_numpy().reshape(self.a.shape)
            self.tmp.free()
        return False


def _ctx(ctx):
    return ctx or default_ctx()


def _scalar(a, s):
    return _lib.SCALAR[s](float(a)) if s in "ds" else _lib.SCALAR[s].of(a)


def _same_len(*vs):
    n = dev_len(vs[0])
    for v in vs[1:]:
        assert dev_len(v) == n, "assertion failed: vec1.len() == vec2.len()"   # vecalg.rs:30,57
    return n


def dot(vec1, vec2, ctx=None):
    """vecalg.rs:24-32: sum x*y — no conjugate."""
    return _dot("sprs_dot_", vec1, vec2, ctx)


def conj_dot(vec1, vec2, ctx=None):
    """vecalg.rs:51-59: sum conj(x)*y — conjugate-linear in the first argument."""
    return _dot("sprs_conj_dot_", vec1, vec2, ctx)


def _dot(name, vec1, vec2, ctx):
    ctx = _ctx(ctx)
    with _Staged(vec1, ctx=ctx) as x, _Staged(vec2, ctx=ctx) as y:
        s = dev_sfx(x)
        n = _same_len(x, y)
        out = _lib.SCALAR[s]()
        check(getattr(_lib.lib(), name + s)(ctx.h, n, dev_ptr(x), dev_ptr(y), C.byref(out)), ctx.h)
        return out.value if s in "ds" else out.py()


def norm2(vec, ctx=None):
    """vecalg.rs:63-69: sqrt(sum |x|^2), unscaled."""
    ctx = _ctx(ctx)
    with _Staged(vec, ctx=ctx) as x:
        s = dev_sfx(x)
        out = _lib.REAL[s]()
        check(getattr(_lib.lib(), "sprs_norm2_" + s)(ctx.h, dev_len(x), dev_ptr(x), C.byref(out)), ctx.h)
        return out.value


def scale(a, vec, ctx=None):
    """vecalg.rs:74-81: vec *= a."""
    ctx = _ctx(ctx)
    with _Staged(vec, out=True, ctx=ctx) as x:
        s = dev_sfx(x)
        check(getattr(_lib.lib(), "sprs_scale_" + s)(ctx.h, dev_len(x), _scalar(a, s), dev_ptr(x)), ctx.h)
        ctx.sync()


def rscale(a, vec, ctx=None):
    """vecalg.rs:86-92: vec = vec.mul_real(a)."""
    ctx = _ctx(ctx)
    with _Staged(vec, out=True, ctx=ctx) as x:
        check(getattr(_lib.lib(), "sprs_rscale_" + dev_sfx(x))(ctx.h, dev_len(x), float(a), dev_ptr(x)), ctx.h)
        ctx.sync()


def conj(vec_in, vec_out, ctx=None):
    """vecalg.rs:96-104: vec_out = conj(vec_in)."""
    ctx = _ctx(ctx)
    with _Staged(vec_in, ctx=ctx) as x, _Staged(vec_out, out=True, ctx=ctx) as y:
        n = _same_len(x, y)
        check(getattr(_lib.lib(), "sprs_conj_" + dev_sfx(x))(ctx.h, n, dev_ptr(x), dev_ptr(y)), ctx.h)
        ctx.sync()


def axpy(a, vec1, vec2, ctx=None):
    """vecalg.rs:109-118: vec2 += vec1 * a.  `a` may be real for complex vectors (S=f64, T=c64)."""
    ctx = _ctx(ctx)
    with _Staged(vec1, ctx=ctx) as x, _Staged(vec2, out=True, ctx=ctx) as y:
        s = dev_sfx(y)
        n = _same_len(x, y)
        if s in "zc" and isinstance(a, (int, float, np.floating, np.integer)):
            fn = _lib.lib().sprs_axpy_zd if s == "z" else _lib.lib().sprs_axpy_cs
            check(fn(ctx.h, n, float(a), dev_ptr(x), dev_ptr(y)), ctx.h)
        else:
            check(getattr(_lib.lib(), "sprs_axpy_" + s)(ctx.h, n, _scalar(a, s), dev_ptr(x), dev_ptr(y)), ctx.h)
        ctx.sync()


def axpby(a, vec1, b, vec2, ctx=None):
    """vecalg.rs:135-144: vec2 = vec1*a + vec2*b."""
    ctx = _ctx(ctx)
    with _Staged(vec1, ctx=ctx) as x, _Staged(vec2, out=True, ctx=ctx) as y:
        s = dev_sfx(y)
        n = _same_len(x, y)
        check(getattr(_lib.lib(), "sprs_axpby_" + s)(ctx.h, n, _scalar(a, s), dev_ptr(x), _scalar(b, s), dev_ptr(y)), ctx.h)
        ctx.sync()
