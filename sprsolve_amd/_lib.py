"""ctypes binding of libsprsolve_hip.so (the C ABI in include/sprsolve_hip.h).

The library is the product: there is no CPU fallback.  Importing this module never touches
the GPU; the first call that needs the library loads it and fails loudly if it is missing.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsprsolve_hip.so")
CSRC = os.path.join(_HERE, "csrc")

(OK, INCOMPATIBLE_RHS_SIZE, INCOMPATIBLE_X_SIZE, INSUFFICIENT_ITER, BREAKDOWN, INVALID_PRECOND, DIM_MISMATCH,
 INVALID_ARGUMENT) = range(8)
ZERO_DIAGONAL, NOT_SQUARE, NOT_CSR = 8, 9, 10
ERR_HIP, ERR_RCCL, ERR_NO_DEVICE = 100, 101, 102
SOLVER_BICGSTAB, SOLVER_MINRES, SOLVER_CSMINRES = 1, 2, 3


class c64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]

    @classmethod
    def of(cls, v):
        v = complex(v)
        return cls(v.real, v.imag)

    def py(self):
        return complex(self.re, self.im)


class c32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]

    @classmethod
    def of(cls, v):
        v = complex(v)
        return cls(v.real, v.imag)

    def py(self):
        return complex(self.re, self.im)


# per-suffix C types: scalar passed by value, T::Real, pointer to T::Real
SCALAR = {"d": C.c_double, "z": c64, "s": C.c_float, "c": c32}
REAL = {"d": C.c_double, "z": C.c_double, "s": C.c_float, "c": C.c_float}


def build(force=False, jobs=4):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s", "-j%d" % jobs]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


_lib = None

_vp, _i32, _i64, _sz, _dbl, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_double, C.c_int
_pp = C.POINTER(C.c_void_p)
_pd = C.POINTER(C.c_double)
_psz = C.POINTER(C.c_size_t)


def _protos():
    P = {}
    P["sprs_ctx_create"] = [_int, _vp, _pp]
    P["sprs_ctx_destroy"] = [_vp]
    P["sprs_ctx_sync"] = [_vp]
    P["sprs_ctx_set"] = [_vp, C.c_char_p, _i64]
    P["sprs_malloc"] = [_vp, _sz, _pp]
    P["sprs_free"] = [_vp, _vp]
    for n in ("h2d", "d2h", "d2d"):
        P["sprs_memcpy_" + n] = [_vp, _vp, _vp, _sz]
    P["sprs_memset_zero"] = [_vp, _vp, _sz]
    for s in ("d", "z", "s", "c"):
        sc, re_, pre = SCALAR[s], REAL[s], C.POINTER(REAL[s])
        P["sprs_csr_create_" + s] = [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _pp]
        P["sprs_csr_create_i64_" + s] = [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _pp]
        P["sprs_csr_create_dev_" + s] = [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _pp]
        P["sprs_mul_vec_" + s] = [_vp, _vp, _sz, _vp, _sz]
        P["sprs_mul_vec_dot_" + s] = [_vp, _vp, _sz, _vp, _sz, _vp]
        P["sprs_mul_vec_dev_" + s] = [_vp, _vp, _vp]
        P["sprs_mul_vec_dot_dev_" + s] = [_vp, _vp, _vp, _vp]
        P["sprs_mul_vec_dev_timed_" + s] = [_vp, _vp, _vp, _int, _pd]
        P["sprs_dot_" + s] = [_vp, _sz, _vp, _vp, _vp]
        P["sprs_conj_dot_" + s] = [_vp, _sz, _vp, _vp, _vp]
        P["sprs_norm2_" + s] = [_vp, _sz, _vp, pre]
        P["sprs_scale_" + s] = [_vp, _sz, sc, _vp]
        P["sprs_rscale_" + s] = [_vp, _sz, re_, _vp]
        P["sprs_conj_" + s] = [_vp, _sz, _vp, _vp]
        P["sprs_axpy_" + s] = [_vp, _sz, sc, _vp, _vp]
        P["sprs_axpby_" + s] = [_vp, _sz, sc, _vp, sc, _vp]
        P["sprs_diag_mul_vec_" + s] = [_vp, _vp, _sz, _vp, _sz]
        P["sprs_diag_mul_vec_dev_" + s] = [_vp, _vp, _vp]
        for k in ("bicgstab", "minres", "csminres"):
            P["sprs_%s_create_%s" % (k, s)] = [_vp, _sz, _pp]
            P["sprs_%s_solve_%s" % (k, s)] = [_vp, _vp, _sz, _vp, _sz, _sz, re_, _psz, pre]
        for k in ("bicgstab", "minres"):
            P["sprs_%s_precond_solve_%s" % (k, s)] = [_vp, _vp, _vp, _sz, _vp, _sz, _sz, re_, _psz, pre]
            P["sprs_%s_solve_dev_%s" % (k, s)] = [_vp, _vp, _vp, _sz, _vp, _sz, _sz, re_, _psz, pre]
        P["sprs_csminres_solve_dev_" + s] = [_vp, _vp, _sz, _vp, _sz, _sz, re_, _psz, pre]
    P["sprs_axpy_zd"] = [_vp, _sz, _dbl, _vp, _vp]
    P["sprs_axpy_cs"] = [_vp, _sz, C.c_float, _vp, _vp]
    P["sprs_csr_destroy"] = [_vp]
    for k in ("bicgstab", "minres", "csminres"):
        P["sprs_%s_destroy" % k] = [_vp]
    for s in ("d", "zd", "z", "s", "cs", "c"):
        P["sprs_diag_precond_create_" + s] = [_vp, _sz, _vp, _pp]
    P["sprs_diag_precond_destroy"] = [_vp]
    P["sprs_comm_unique_id"] = [_vp]
    P["sprs_comm_create"] = [_vp, _int, _int, _vp, _pp]
    P["sprs_comm_destroy"] = [_vp]
    P["sprs_comm_count"] = [_vp, C.POINTER(_int)]
    P["sprs_comm_p2p"] = [_vp, C.POINTER(_int)]
    P["sprs_comm_allreduce_sum_f64"] = [_vp, _vp, _sz]
    P["sprs_comm_allreduce_timed_f64"] = [_vp, _vp, _sz, _int, _pd]
    for s in ("d", "z", "s", "c"):
        P["sprs_dist_csr_create_dev_" + s] = [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _pp]
        P["sprs_dist_mul_vec_dev_" + s] = [_vp, _vp, _vp]
        P["sprs_dist_csr_create_allgather_dev_" + s] = [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _pp]
        P["sprs_dist_csr_create_global_dev_" + s] = [_vp, _vp, _i64, _vp, _vp, _vp, _int, _int, _pp]
    P["sprs_dist_csr_info"] = [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int), C.POINTER(_i64), C.POINTER(_i64)]
    P["sprs_dist_csr_peers"] = [_vp, _int, _vp, _vp, _vp]
    P["sprs_dist_csr_send_idx"] = [_vp, _i64, _vp]
    P["sprs_csr_stream_format"] = [_vp, C.POINTER(_int), C.POINTER(_int)]
    P["sprs_csr_wide_blocks"] = [_vp, C.POINTER(_i64), C.POINTER(_i64)]
    P["sprs_csr_tile_plan"] = [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]
    P["sprs_csr_chain_plan"] = [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]
    P["sprs_gauss_seidel_create"] = [_vp, _pp]
    P["sprs_gauss_seidel_destroy"] = [_vp]
    for s in ("d", "s"):
        P["sprs_gauss_seidel_solve_" + s] = [_vp, _vp, _sz, _vp, _sz, _sz, REAL[s], _psz, C.POINTER(REAL[s])]
        P["sprs_gauss_seidel_solve_dev_" + s] = [_vp, _vp, _sz, _vp, _sz, _sz, REAL[s], _psz, C.POINTER(REAL[s])]
    P["sprs_solver_set_mode"] = [_vp, _int, _int]
    P["sprs_solver_set_trace"] = [_vp, _int, _vp, _sz]
    P["sprs_solver_trace_rows"] = [_vp, _int, _psz]
    P["sprs_solver_set_profile"] = [_vp, _int, _int]
    P["sprs_solver_get_profile"] = [_vp, _int, _pd, C.POINTER(_i64), _pd]
    P["sprs_solver_get_fused_launches"] = [_vp, _int, C.POINTER(_i64), C.POINTER(_i64)]
    P["sprs_solver_get_profile_counts"] = [_vp, _int, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]
    return P


PROTOTYPES = _protos()
# entry points declared in include/sprsolve_hip.h that return something other than int
_SPECIAL_RET = {"sprs_last_error": C.c_char_p, "sprs_status_str": C.c_char_p, "sprs_ctx_get": _i64,
                "sprs_gauss_seidel_levels": _i64, "sprs_csr_rows": _i64, "sprs_csr_cols": _i64, "sprs_csr_nnz": _i64, "sprs_version": _int}


def lib():
    """Load libsprsolve_hip.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libsprsolve_hip.so is missing at %s — run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (or make -C sprsolve_amd/csrc). There is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64.so (soname libamdhip64.so.7) and its libraries ask
        # for it by the unversioned name: if this library pulled in /opt/rocm's copy first, a later
        # `import torch` would start a SECOND HIP runtime in the process.  Loading torch first makes
        # both share one runtime (same soname), so torch tensors and our kernels see the same device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, args in PROTOTYPES.items():
            f = getattr(L, name)
            f.argtypes = args
            f.restype = _int
        L.sprs_last_error.argtypes = [_vp]; L.sprs_last_error.restype = C.c_char_p
        L.sprs_status_str.argtypes = [_int]; L.sprs_status_str.restype = C.c_char_p
        L.sprs_ctx_get.argtypes = [_vp, C.c_char_p]; L.sprs_ctx_get.restype = _i64
        for n in ("rows", "cols", "nnz"):
            f = getattr(L, "sprs_csr_" + n); f.argtypes = [_vp]; f.restype = _i64
        L.sprs_version.argtypes = []; L.sprs_version.restype = _int
        L.sprs_gauss_seidel_levels.argtypes = [_vp]; L.sprs_gauss_seidel_levels.restype = _i64
        _lib = L
    return _lib


def all_symbols():
    return sorted(list(PROTOTYPES) + list(_SPECIAL_RET))
