"""`MatVecMul` and the device CSR operator that implements it.

Mirror of the reference's src/mat.rs: the trait has the four methods of mat.rs:12-37; `HipCsr`
is the MI355X backend handle — created from CSR (or CSC) arrays the way `MklMat::new` consumes
a `CsMatI<T,i32>` (src/mkl_mat.rs:32-74) and released like its `Drop` (mkl_mat.rs:322-333).
"""
import ctypes as C

import numpy as np

from . import _lib
from .device import DevVec, default_ctx, dev_len, dev_ptr, dev_sfx, is_device_array, pre_sync, sfx
from .error import check


class MatVecMul:
    """src/mat.rs:12-37."""

    def mul_vec(self, v_in, v_out):
        raise NotImplementedError

    def mul_vec_dot(self, v_in, v_out):
        raise NotImplementedError

    def mul_vec_unchecked(self, v_in, v_out):
        raise NotImplementedError

    def mul_vec_dot_unchecked(self, v_in, v_out):
        raise NotImplementedError


class HipCsr(MatVecMul):
    """Device-resident CSR matrix; `impl MatVecMul<T>` (src/mat.rs:47-153)."""

    def __init__(self, handle, ctx, dtype, shape, keepalive=None):
        self.h, self.ctx, self.dtype, self.shape = handle, ctx, np.dtype(dtype), tuple(shape)
        self._keep = keepalive

    # -------------------------------------------------------------- constructors
    @classmethod
    def new(cls, shape, indptr, indices, data, storage="CSR", ctx=None):
        """From host arrays.  Index dtypes i32 / u32 / i64 / u64 / usize (mat.rs:196-199);
        storage "CSR" or "CSC" (mat.rs:75,130-142)."""
        ctx = ctx or default_ctx()
        data = np.ascontiguousarray(data)
        s = sfx(data.dtype)
        indptr = np.ascontiguousarray(indptr); indices = np.ascontiguousarray(indices)
        nrows, ncols = int(shape[0]), int(shape[1])
        csc = 1 if storage.upper() == "CSC" else 0
        h = C.c_void_p()
        L = _lib.lib()
        if indptr.dtype == np.int32 and indices.dtype == np.int32:
            fn = getattr(L, "sprs_csr_create_" + s)
        else:
            if indptr.dtype.kind not in "iu" or indices.dtype.kind not in "iu":
                raise TypeError("index arrays must be integer")
            if indptr.dtype == np.uint64 and indptr.size and int(indptr.max()) > 2**62:
                raise ValueError("index out of range")
            indptr = indptr.astype(np.int64); indices = indices.astype(np.int64)
            fn = getattr(L, "sprs_csr_create_i64_" + s)
        st = fn(ctx.h, nrows, ncols, int(data.size), indptr.ctypes.data_as(C.c_void_p),
                indices.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p), csc, C.byref(h))
        check(st, ctx.h)
        return cls(h, ctx, data.dtype, (nrows, ncols))

    @classmethod
    def from_scipy(cls, m, ctx=None):
        m = m.tocsr()
        return cls.new(m.shape, m.indptr, m.indices, m.data, "CSR", ctx)

    @classmethod
    def from_device(cls, shape, nnz, indptr_dev, indices_dev, data_dev, adopt=True, ctx=None):
        """From i32 CSR arrays already in HBM (DevVec or torch CUDA tensors).  adopt=True keeps
        a reference to the caller's arrays instead of copying them."""
        ctx = ctx or default_ctx()
        s = dev_sfx(data_dev)
        h = C.c_void_p()
        st = getattr(_lib.lib(), "sprs_csr_create_dev_" + s)(
            ctx.h, int(shape[0]), int(shape[1]), int(nnz), dev_ptr(indptr_dev), dev_ptr(indices_dev),
            dev_ptr(data_dev), 1 if adopt else 0, C.byref(h))
        check(st, ctx.h)
        from .device import NP_OF
        return cls(h, ctx, NP_OF[s], shape, keepalive=(indptr_dev, indices_dev, data_dev) if adopt else None)

    # -------------------------------------------------------------- accessors
    def rows(self):
        return self.shape[0]

    def cols(self):
        return self.shape[1]

    def nnz(self):
        return int(_lib.lib().sprs_csr_nnz(self.h))

    def stream_format(self):
        """(mode, n_offsets, n_pairs): 0 plain CSR, 1 offset codes + values, 2 (offset, value) pair codes (csrc/spmv_dict.hip)."""
        no, nv = C.c_int(0), C.c_int(0)
        m = _lib.lib().sprs_csr_stream_format(self.h, C.byref(no), C.byref(nv))
        return int(m), no.value, nv.value

    def wide_blocks(self):
        """(n_blocks, n_uniform) of the two-rows-per-lane kernel's 128-row blocks (f64 pair codes), else (0, 0)."""
        nb, nu = C.c_int64(0), C.c_int64(0)
        check(_lib.lib().sprs_csr_wide_blocks(self.h, C.byref(nb), C.byref(nu)), self.ctx.h)
        return nb.value, nu.value

    def tile_plan(self):
        """(n_tiles, n_tile_blocks, n_other_blocks) of the f64 pair-code stream's LDS-window tiles (knob spmv_tile), else zeros."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(_lib.lib().sprs_csr_tile_plan(self.h, C.byref(a), C.byref(b), C.byref(c)), self.ctx.h)
        return a.value, b.value, c.value

    def chain_plan(self):
        """(n_tiles, n_segments, n_chains, n_other_blocks) of the f64 pair-code stream's plane-streaming chains (knob spmv_chain;
        they take precedence over the LDS-window tiles), else zeros."""
        a, b, c, d = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(_lib.lib().sprs_csr_chain_plan(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), self.ctx.h)
        return a.value, b.value, c.value, d.value

    # -------------------------------------------------------------- MatVecMul
    def _s(self):
        return sfx(self.dtype)

    def mul_vec(self, v_in, v_out):
        """mat.rs:49-56: checked; raises DimensionMismatch where the reference panics."""
        if is_device_array(v_in):
            if self.cols() != dev_len(v_in) or dev_len(v_in) != dev_len(v_out):
                from .error import DimensionMismatch
                raise DimensionMismatch("Dimension mismatch")
            return self.mul_vec_unchecked(v_in, v_out)
        x = np.ascontiguousarray(v_in, dtype=self.dtype)
        assert isinstance(v_out, np.ndarray) and v_out.dtype == self.dtype and v_out.flags.c_contiguous
        st = getattr(_lib.lib(), "sprs_mul_vec_" + self._s())(
            self.h, x.ctypes.data_as(C.c_void_p), x.size, v_out.ctypes.data_as(C.c_void_p), v_out.size)
        check(st, self.ctx.h)

    def mul_vec_dot(self, v_in, v_out):
        """mat.rs:58-64: v_out = A v_in ; returns conj(v_in).v_out."""
        if is_device_array(v_in):
            if self.cols() != dev_len(v_in) or dev_len(v_in) != dev_len(v_out):
                from .error import DimensionMismatch
                raise DimensionMismatch("Dimension mismatch")
            return self.mul_vec_dot_unchecked(v_in, v_out)
        x = np.ascontiguousarray(v_in, dtype=self.dtype)
        assert isinstance(v_out, np.ndarray) and v_out.dtype == self.dtype and v_out.flags.c_contiguous
        s = self._s()
        out = _lib.SCALAR[s]()
        st = getattr(_lib.lib(), "sprs_mul_vec_dot_" + s)(
            self.h, x.ctypes.data_as(C.c_void_p), x.size, v_out.ctypes.data_as(C.c_void_p), v_out.size, C.byref(out))
        check(st, self.ctx.h)
        return out.value if s in "ds" else out.py()

    def mul_vec_unchecked(self, v_in, v_out):
        """mat.rs:68-143 on device vectors (no dimension check, nothing crosses PCIe).  Blocking."""
        pre_sync(v_in, v_out)
        st = getattr(_lib.lib(), "sprs_mul_vec_dev_" + self._s())(self.h, dev_ptr(v_in), dev_ptr(v_out))
        check(st, self.ctx.h)
        self.ctx.sync()

    def mul_vec_dot_unchecked(self, v_in, v_out):
        """mat.rs:145-152 on device vectors."""
        s = self._s()
        pre_sync(v_in, v_out)
        out = _lib.SCALAR[s]()
        st = getattr(_lib.lib(), "sprs_mul_vec_dot_dev_" + s)(self.h, dev_ptr(v_in), dev_ptr(v_out), C.byref(out))
        check(st, self.ctx.h)
        return out.value if s in "ds" else out.py()

    def time_mul_vec(self, v_in, v_out, reps=20):
        """Mean device milliseconds of one SpMV launch (HIP events on the library's stream)."""
        pre_sync(v_in, v_out)
        ms = C.c_double()
        st = getattr(_lib.lib(), "sprs_mul_vec_dev_timed_" + self._s())(self.h, dev_ptr(v_in), dev_ptr(v_out), int(reps),
                                                                       C.byref(ms))
        check(st, self.ctx.h)
        return ms.value

    def algorithmic_bytes(self):
        """nnz*(s+4) + (n+1)*4 + n*s (x) + n*s (y)."""
        s = self.dtype.itemsize
        n = self.rows()
        return self.nnz() * (s + 4) + (n + 1) * 4 + 2 * n * s

    def close(self):
        if self.h:
            _lib.lib().sprs_csr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
