"""Device context and device vectors (thin wrappers over the C ABI's ctx / malloc / memcpy)."""
import ctypes as C

import numpy as np

from . import _lib
from .error import check

_DTYPES = {np.dtype(np.float64): "d", np.dtype(np.complex128): "z", np.dtype(np.float32): "s", np.dtype(np.complex64): "c"}
NP_OF = {"d": np.float64, "z": np.complex128, "s": np.float32, "c": np.complex64}


def sfx(dtype):
    try:
        return _DTYPES[np.dtype(dtype)]
    except KeyError:
        raise TypeError("sprsolve_amd implements f32 / f64 / Complex<f32> / Complex<f64> (got %s)" % dtype)


class Context:
    """One GPU + one HIP stream (sprs_ctx)."""

    def __init__(self, device=0, stream=None):
        L = _lib.lib()
        h = C.c_void_p()
        st = L.sprs_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        check(st)
        self.h = h
        self.device = int(device)

    def set(self, key, value):
        check(_lib.lib().sprs_ctx_set(self.h, key.encode(), int(value)), self.h)

    def get(self, key):
        return int(_lib.lib().sprs_ctx_get(self.h, key.encode()))

    def sync(self):
        check(_lib.lib().sprs_ctx_sync(self.h), self.h)

    def close(self):
        if self.h:
            _lib.lib().sprs_ctx_destroy(self.h)
            self.h = None


_default = {}


def default_ctx(device=0):
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]


class DevVec:
    """A vector in HBM owned through sprs_malloc / sprs_free."""

    def __init__(self, n, dtype, ctx=None):
        self.ctx = ctx or default_ctx()
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        sfx(self.dtype)
        p = C.c_void_p()
        check(_lib.lib().sprs_malloc(self.ctx.h, self.n * self.dtype.itemsize, C.byref(p)), self.ctx.h)
        self.ptr = p

    @classmethod
    def from_numpy(cls, a, ctx=None):
        a = np.ascontiguousarray(a)
        v = cls(a.size, a.dtype, ctx)
        v.upload(a)
        return v

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size == self.n
        check(_lib.lib().sprs_memcpy_h2d(self.ctx.h, self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes), self.ctx.h)

    def to_numpy(self):
        out = np.empty(self.n, dtype=self.dtype)
        check(_lib.lib().sprs_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes), self.ctx.h)
        return out

    def zero(self):
        check(_lib.lib().sprs_memset_zero(self.ctx.h, self.ptr, self.n * self.dtype.itemsize), self.ctx.h)

    def free(self):
        if self.ptr:
            _lib.lib().sprs_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __len__(self):
        return self.n

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def is_device_array(a):
    """DevVec, or a torch CUDA tensor (torch is optional plumbing, never imported here)."""
    if isinstance(a, DevVec):
        return True
    return hasattr(a, "data_ptr") and getattr(a, "is_cuda", False)


def pre_sync(*arrays):
    """torch tensors are produced on torch's stream while the library runs on its own: make sure
    pending torch work on them has finished before the library touches them."""
    for a in arrays:
        if a is not None and type(a).__module__.split(".")[0] == "torch":
            import torch
            torch.cuda.current_stream(a.device).synchronize()
            return


def dev_ptr(a):
    if isinstance(a, DevVec):
        return a.ptr
    return C.c_void_p(a.data_ptr())


def dev_len(a):
    if isinstance(a, DevVec):
        return a.n
    return int(a.numel())


def dev_sfx(a):
    if isinstance(a, DevVec):
        return sfx(a.dtype)
    name = str(a.dtype)
    for key, s in (("float64", "d"), ("complex128", "z"), ("float32", "s"), ("complex64", "c")):
        if name.endswith(key):
            return s
    raise TypeError("unsupported device dtype %s" % name)
