// extern "C" surface of libsprsolve_hip.so — see include/sprsolve_hip.h for the contract and
// the reference interface each entry point replaces.  Nothing here throws.
#include <sys/mman.h>

#include <cstdlib>
#include <memory>
#include <new>
#include <string>

#include "krylov.hpp"

using namespace sprs;

static_assert(sizeof(sprs_c64) == sizeof(cplx), "Complex<f64> layout");
static_assert(sizeof(sprs_c32) == sizeof(cplxf), "Complex<f32> layout");

#define SPRS_GUARD_BEGIN try {
#define SPRS_GUARD_END                         \
    }                                          \
    catch (const std::bad_alloc &) {           \
        return SPRS_ERR_HIP;                   \
    }                                          \
    catch (...) {                              \
        return SPRS_ERR_HIP;                   \
    }

extern "C" {

// ------------------------------------------------------------------------------ context
int sprs_version(void) { return 100; }

const char *sprs_status_str(int s) {
    switch (s) {
        case SPRS_OK: return "Ok";
        case SPRS_INCOMPATIBLE_RHS_SIZE: return "Incompatible input matrix format: Input vec dimension doesn't match the matrix size";
        case SPRS_INCOMPATIBLE_X_SIZE: return "Incompatible input matrix format: Input and output vec dimension do not match";
        case SPRS_INSUFFICIENT_ITER: return "Insufficient interation #";
        case SPRS_BREAKDOWN: return "Solver break down";
        case SPRS_INVALID_PRECOND: return "Invalid preconditioner";
        case SPRS_DIM_MISMATCH: return "Dimension mismatch";
        case SPRS_INVALID_ARGUMENT: return "Invalid argument";
        case SPRS_ZERO_DIAGONAL: return "Matrix has zero diagonal element";
        case SPRS_NOT_SQUARE: return "Incompatible input matrix format: Not a square matrix";
        case SPRS_NOT_CSR: return "Incompatible input matrix format: Not in CSR format";
        case SPRS_ERR_HIP: return "HIP runtime error";
        case SPRS_ERR_RCCL: return "RCCL error";
        case SPRS_ERR_NO_DEVICE: return "No usable GPU device";
        default: return "Unknown status";
    }
}

int sprs_ctx_create(int device, void *stream, sprs_ctx **out) {
    SPRS_GUARD_BEGIN
    if (!out) return SPRS_INVALID_ARGUMENT;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SPRS_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return SPRS_ERR_NO_DEVICE;
    sprs_ctx *c = new sprs_ctx();
    c->device = device;
    auto fail = [&](int st) { delete c; return st; };
    if (hipSetDevice(device) != hipSuccess) return fail(SPRS_ERR_NO_DEVICE);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(SPRS_ERR_HIP);
        c->own_stream = true;
    }
    // 4 workgroups of 256 lanes per CU: enough loads in flight to saturate HBM while the
    // number of reduction partials (== grid) stays small enough for the fused prologues
    // streaming / reduction kernels: 2 workgroups per CU (measured on the cfg-5 solve: 256 -> 1.99, 384 -> 1.90,
    // 512 -> 1.86, 768 -> 1.86, 1024 -> 2.00, 2048 -> 2.06 ms per BiCGStab iteration; profiles/r01_tuning.md)
    c->grid = ((c->num_cu * 2 + 7) / 8) * 8;
    if (c->grid > MAX_GRID) c->grid = MAX_GRID;
    if (hipMalloc((void **)&c->d_part, sizeof(double) * 2 * MAX_GRID) != hipSuccess) return fail(SPRS_ERR_HIP);
    if (hipMalloc((void **)&c->d_scal, 256) != hipSuccess) return fail(SPRS_ERR_HIP);
    if (hipHostMalloc((void **)&c->h_scal, 256, hipHostMallocDefault) != hipSuccess) return fail(SPRS_ERR_HIP);
    *out = c;
    return SPRS_OK;
    SPRS_GUARD_END
}

int sprs_ctx_destroy(sprs_ctx *c) {
    if (!c) return SPRS_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->d_part) (void)hipFree(c->d_part);
    if (c->d_scal) (void)hipFree(c->d_scal);
    if (c->h_scal) (void)hipHostFree(c->h_scal);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SPRS_OK;
}

int sprs_ctx_sync(sprs_ctx *c) {
    if (!c) return SPRS_INVALID_ARGUMENT;
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPRS_OK;
}

const char *sprs_last_error(const sprs_ctx *c) { return c ? c->err : "null context"; }

int sprs_ctx_set(sprs_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return SPRS_INVALID_ARGUMENT;
    std::string k(key);
    if (k == "grid") { if (value < 8 || value > MAX_GRID) return SPRS_INVALID_ARGUMENT; c->grid = (int)(value & ~7); }
    else if (k == "spmv_grid") { if (value > MAX_GRID / 2 || (value >= 0 && value < 8)) return SPRS_INVALID_ARGUMENT; c->spmv_grid = value < 0 ? -1 : (int)(value & ~7); }
    else if (k == "xcd_chunk") c->xcd_chunk = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "spmv_dict") { if (value < -1 || value > 2) return SPRS_INVALID_ARGUMENT; c->spmv_dict = (int)value; }
    else if (k == "spmv_wide") c->spmv_wide = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "spmv_uniform") c->spmv_uniform = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "spmv_period") c->spmv_period = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "spmv_triple") c->spmv_triple = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "spmv_seam") c->spmv_seam = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "spmv_tile") c->spmv_tile = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "spmv_chain") c->spmv_chain = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "spmv_fuse") c->spmv_fuse = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "p2p_allreduce") c->p2p_allreduce = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "p2p_timeout_ms") c->p2p_timeout_ms = value < 1 ? 1 : value;
    else if (k == "ew_chunk") c->ew_chunk = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "stream_nt") c->stream_nt = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "spmv_eqrows") c->spmv_eqrows = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "spmv_wideload") c->spmv_wideload = value < 0 ? -1 : (value ? 1 : 0);
    else if (k == "halo_overlap") c->halo_overlap = value ? 1 : 0;
    else if (k == "gs_graph") c->gs_graph = value ? 1 : 0;
    else if (k == "poll") { if (value < 1) return SPRS_INVALID_ARGUMENT; c->poll = (int)value; }
    else return SPRS_INVALID_ARGUMENT;
    return SPRS_OK;
}
int64_t sprs_ctx_get(const sprs_ctx *c, const char *key) {
    if (!c || !key) return -1;
    std::string k(key);
    if (k == "grid") return c->grid;
    if (k == "xcd_chunk") return c->xcd_chunk;
    if (k == "spmv_grid") return c->spmv_grid;
    if (k == "spmv_dict") return c->spmv_dict;
    if (k == "spmv_wide") return c->spmv_wide;
    if (k == "spmv_uniform") return c->spmv_uniform;
    if (k == "spmv_period") return c->spmv_period;
    if (k == "spmv_triple") return c->spmv_triple;
    if (k == "spmv_seam") return c->spmv_seam;
    if (k == "spmv_tile") return c->spmv_tile;
    if (k == "spmv_chain") return c->spmv_chain;
    if (k == "spmv_fuse") return c->spmv_fuse;
    if (k == "p2p_allreduce") return c->p2p_allreduce;
    if (k == "p2p_timeout_ms") return c->p2p_timeout_ms;
    if (k == "ew_chunk") return c->ew_chunk;
    if (k == "stream_nt") return c->stream_nt;
    if (k == "spmv_eqrows") return c->spmv_eqrows;
    if (k == "spmv_wideload") return c->spmv_wideload;
    if (k == "halo_overlap") return c->halo_overlap;
    if (k == "gs_graph") return c->gs_graph;
    if (k == "poll") return c->poll;
    if (k == "num_cu") return c->num_cu;
    if (k == "device") return c->device;
    return -1;
}

int sprs_malloc(sprs_ctx *c, size_t bytes, void **dev_out) {
    if (!c || !dev_out) return SPRS_INVALID_ARGUMENT;
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    SPRS_HIP_TRY(c, hipMalloc(dev_out, bytes ? bytes : 16));
    return SPRS_OK;
}
int sprs_free(sprs_ctx *c, void *dev) {
    if (!c) return SPRS_INVALID_ARGUMENT;
    if (!dev) return SPRS_OK;
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    SPRS_HIP_TRY(c, hipFree(dev));
    return SPRS_OK;
}
int sprs_memcpy_h2d(sprs_ctx *c, void *d, const void *h, size_t bytes) {
    if (!c) return SPRS_INVALID_ARGUMENT;
    SPRS_HIP_TRY(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPRS_OK;
}
int sprs_memcpy_d2h(sprs_ctx *c, void *h, const void *d, size_t bytes) {
    if (!c) return SPRS_INVALID_ARGUMENT;
    SPRS_HIP_TRY(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPRS_OK;
}
int sprs_memcpy_d2d(sprs_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c) return SPRS_INVALID_ARGUMENT;
    SPRS_HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return SPRS_OK;
}
int sprs_memset_zero(sprs_ctx *c, void *d, size_t bytes) {
    if (!c) return SPRS_INVALID_ARGUMENT;
    SPRS_HIP_TRY(c, hipMemsetAsync(d, 0, bytes, c->stream));
    return SPRS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------ CSR operator
namespace {

template <class I>
bool narrow_check(const I *a, int64_t n, int64_t lo, int64_t hi) {
    for (int64_t i = 0; i < n; ++i)
        if ((int64_t)a[i] < lo || (int64_t)a[i] > hi) return false;
    return true;
}

// Build the device CSR from host arrays (CSR or CSC, any integer index type).
template <class T, class I>
int csr_create_host(sprs_ctx *c, int64_t nrows, int64_t ncols, int64_t nnz, const I *ptr, const I *idx, const T *val,
                    int storage_csc, sprs_csr **out) {
    if (!c || !out || nrows < 0 || ncols < 0 || nnz < 0) return SPRS_INVALID_ARGUMENT;
    if (!ptr || (nnz > 0 && (!idx || !val))) return SPRS_INVALID_ARGUMENT;
    CtxLock lock(c);
    *out = nullptr;
    if (nrows >= INT32_MAX || ncols >= INT32_MAX || nnz >= INT32_MAX) return SPRS_INVALID_ARGUMENT;
    const int64_t nouter = storage_csc ? ncols : nrows, ninner = storage_csc ? nrows : ncols;
    // validate: the kernels index x and y with these arrays
    if ((int64_t)ptr[0] != 0 || (int64_t)ptr[nouter] != nnz) return SPRS_INVALID_ARGUMENT;
    for (int64_t i = 0; i < nouter; ++i)
        if ((int64_t)ptr[i + 1] < (int64_t)ptr[i]) return SPRS_INVALID_ARGUMENT;
    if (!narrow_check(idx, nnz, 0, ninner - 1)) return SPRS_INVALID_ARGUMENT;

    std::vector<int32_t> rp((size_t)nrows + 1), ci((size_t)nnz);
    std::vector<T> vv;
    const T *vsrc = val;
    if (!storage_csc) {
        for (int64_t i = 0; i <= nrows; ++i) rp[i] = (int32_t)ptr[i];
        for (int64_t k = 0; k < nnz; ++k) ci[k] = (int32_t)idx[k];
    } else {
        // CSC -> CSR, stable in column order: per row the terms keep the order in which the
        // reference's serial scatter (mat.rs:135-141) adds them, so y has the same bits.
        vv.resize((size_t)nnz);
        std::fill(rp.begin(), rp.end(), 0);
        for (int64_t k = 0; k < nnz; ++k) rp[(size_t)idx[k] + 1]++;
        for (int64_t i = 0; i < nrows; ++i) rp[i + 1] += rp[i];
        std::vector<int32_t> fill(rp.begin(), rp.end() - 1);
        for (int64_t col = 0; col < ncols; ++col)
            for (int64_t k = (int64_t)ptr[col]; k < (int64_t)ptr[col + 1]; ++k) {
                int32_t dst = fill[(size_t)idx[k]]++;
                ci[dst] = (int32_t)col;
                vv[dst] = val[k];
            }
        vsrc = vv.data();
    }
    sprs_csr *A = new sprs_csr();
    A->ctx = c; A->dtype = dtype_of<T>::value;
    A->nrows = nrows; A->ncols = ncols; A->nnz = nnz; A->owns_arrays = true; A->was_csc = storage_csc != 0;
    auto fail = [&](int st) { sprs_csr_destroy(A); return st; };
    if (hipSetDevice(c->device) != hipSuccess) return fail(SPRS_ERR_HIP);
    if (hipMalloc((void **)&A->row_ptr, sizeof(int32_t) * ((size_t)nrows + 1)) != hipSuccess) return fail(SPRS_ERR_HIP);
    if (hipMalloc((void **)&A->col_idx, sizeof(int32_t) * (size_t)(nnz ? nnz : 1)) != hipSuccess) return fail(SPRS_ERR_HIP);
    if (hipMalloc((void **)&A->val, sizeof(T) * (size_t)(nnz ? nnz : 1)) != hipSuccess) return fail(SPRS_ERR_HIP);
    if (hipMemcpyAsync(A->row_ptr, rp.data(), sizeof(int32_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail(SPRS_ERR_HIP);
    if (nnz) {
        if (hipMemcpyAsync(A->col_idx, ci.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail(SPRS_ERR_HIP);
        if (hipMemcpyAsync(A->val, vsrc, sizeof(T) * (size_t)nnz, hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail(SPRS_ERR_HIP);
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(SPRS_ERR_HIP);
    int st = build_rowblocks(A, rp.data());
    if (st != SPRS_OK) return fail(st);
    *out = A;
    return SPRS_OK;
}

template <class T>
int csr_create_dev(sprs_ctx *c, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t *d_rp, const int32_t *d_ci,
                   const T *d_val, int adopt, sprs_csr **out) {
    if (!c || !out || !d_rp || nrows < 0 || ncols < 0 || nnz < 0) return SPRS_INVALID_ARGUMENT;
    if (nnz > 0 && (!d_ci || !d_val)) return SPRS_INVALID_ARGUMENT;
    CtxLock lock(c);
    *out = nullptr;
    if (nrows >= INT32_MAX || ncols >= INT32_MAX || nnz >= INT32_MAX) return SPRS_INVALID_ARGUMENT;
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    CreateTrace tr;
    // row_ptr stays in HBM: build_rowblocks validates it (and copies it to the host only for irregular matrices)
    sprs_csr *A = new sprs_csr();
    A->ctx = c; A->dtype = dtype_of<T>::value;
    A->nrows = nrows; A->ncols = ncols; A->nnz = nnz;
    auto fail = [&](int st) { sprs_csr_destroy(A); return st; };
    if (adopt) {
        A->owns_arrays = false;
        A->row_ptr = const_cast<int32_t *>(d_rp); A->col_idx = const_cast<int32_t *>(d_ci);
        A->val = const_cast<T *>(d_val);
    } else {
        A->owns_arrays = true;
        if (hipMalloc((void **)&A->row_ptr, sizeof(int32_t) * ((size_t)nrows + 1)) != hipSuccess) return fail(SPRS_ERR_HIP);
        if (hipMalloc((void **)&A->col_idx, sizeof(int32_t) * (size_t)(nnz ? nnz : 1)) != hipSuccess) return fail(SPRS_ERR_HIP);
        if (hipMalloc((void **)&A->val, sizeof(T) * (size_t)(nnz ? nnz : 1)) != hipSuccess) return fail(SPRS_ERR_HIP);
        if (hipMemcpyAsync(A->row_ptr, d_rp, sizeof(int32_t) * ((size_t)nrows + 1), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return fail(SPRS_ERR_HIP);
        if (nnz) {
            if (hipMemcpyAsync(A->col_idx, d_ci, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return fail(SPRS_ERR_HIP);
            if (hipMemcpyAsync(A->val, d_val, sizeof(T) * (size_t)nnz, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return fail(SPRS_ERR_HIP);
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(SPRS_ERR_HIP);
    }
    int st = validate_cols_device(A);
    if (st != SPRS_OK) return fail(st);
    tr.lap("column range check (device)");
    st = build_rowblocks(A, nullptr);
    if (st != SPRS_OK) return fail(st);
    tr.lap("build_rowblocks total");
    *out = A;
    return SPRS_OK;
}

template <class T>
int ensure_tmp(const sprs_csr *Ac) {
    sprs_csr *A = const_cast<sprs_csr *>(Ac);
    sprs_ctx *c = A->ctx;
    const size_t m = (size_t)(A->nrows > A->ncols ? A->nrows : A->ncols) + 2;
    if (!A->x_tmp) SPRS_HIP_TRY(c, hipMalloc(&A->x_tmp, sizeof(T) * m));
    if (!A->y_tmp) SPRS_HIP_TRY(c, hipMalloc(&A->y_tmp, sizeof(T) * m));
    if (!A->part) SPRS_HIP_TRY(c, hipMalloc((void **)&A->part, sizeof(double) * 2 * MAX_GRID));
    return SPRS_OK;
}

// MatVecMul::mul_vec / mul_vec_dot over host slices (mat.rs:49-64)
template <class T>
int mul_vec_host(const sprs_csr *A, const T *x, size_t x_len, T *y, size_t y_len, T *dot_out) {
    if (!A || !x || !y) return SPRS_INVALID_ARGUMENT;
    if (A->dtype != dtype_of<T>::value) return SPRS_INVALID_ARGUMENT;
    if ((size_t)A->ncols != x_len || x_len != y_len) return SPRS_DIM_MISMATCH;   // mat.rs:50-52
    sprs_ctx *c = A->ctx;
    CtxLock lock(c);   // x_tmp / y_tmp / part are per-handle staging: `&self` calls from several threads queue here
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    SPRS_TRY(ensure_tmp<T>(A));
    T *dx = (T *)A->x_tmp, *dy = (T *)A->y_tmp;
    SPRS_HIP_TRY(c, hipMemcpyAsync(dx, x, sizeof(T) * x_len, hipMemcpyHostToDevice, c->stream));
    T *part = (T *)A->part;
    SPRS_TRY(launch_spmv<T>(A, dx, dy, dot_out ? 1 : 0, dx, part, nullptr, nullptr));
    const size_t ncopy = (size_t)A->nrows < y_len ? (size_t)A->nrows : y_len;
    SPRS_HIP_TRY(c, hipMemcpyAsync(y, dy, sizeof(T) * ncopy, hipMemcpyDeviceToHost, c->stream));
    if (dot_out) SPRS_TRY(reduce_partials_host<T>(c, part, spmv_num_partials(A), dot_out));   // mat.rs:151
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPRS_OK;
}

template <class T>
int mul_vec_dev(const sprs_csr *A, const T *dx, T *dy, T *dot_out) {
    if (!A || !dx || !dy) return SPRS_INVALID_ARGUMENT;
    if (A->dtype != dtype_of<T>::value) return SPRS_INVALID_ARGUMENT;
    sprs_ctx *c = A->ctx;
    CtxLock lock(c);
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    if (!dot_out) return launch_spmv<T>(A, dx, dy, 0, nullptr, nullptr, nullptr, nullptr);
    SPRS_TRY(ensure_tmp<T>(A));
    T *part = (T *)A->part;
    SPRS_TRY(launch_spmv<T>(A, dx, dy, 1, dx, part, nullptr, nullptr));
    return reduce_partials_host<T>(c, part, spmv_num_partials(A), dot_out);
}

template <class T>
int mul_vec_timed(const sprs_csr *A, const T *dx, T *dy, int reps, double *ms) {
    if (!A || !dx || !dy || !ms || reps < 1) return SPRS_INVALID_ARGUMENT;
    if (A->dtype != dtype_of<T>::value) return SPRS_INVALID_ARGUMENT;
    sprs_ctx *c = A->ctx;
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    CtxLock lock(c);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Ev { hipEvent_t &a, &b; ~Ev() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } guard{e0, e1};   // every return path
    SPRS_HIP_TRY(c, hipEventCreate(&e0));
    SPRS_HIP_TRY(c, hipEventCreate(&e1));
    SPRS_HIP_TRY(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; ++i) SPRS_TRY(launch_spmv<T>(A, dx, dy, 0, nullptr, nullptr, nullptr, nullptr));
    SPRS_HIP_TRY(c, hipEventRecord(e1, c->stream));
    SPRS_HIP_TRY(c, hipEventSynchronize(e1));
    float t = 0.f;
    SPRS_HIP_TRY(c, hipEventElapsedTime(&t, e0, e1));
    *ms = (double)t / reps;
    return SPRS_OK;
}

}  // namespace

// ------------------------------------------------------------------------------ Jacobi preconditioner
namespace {
template <class V>
int diag_create(sprs_ctx *c, size_t n, const V *diag_host, int t_dtype, sprs_diag **out) {
    if (!c || !out || (!diag_host && n)) return SPRS_INVALID_ARGUMENT;
    CtxLock lock(c);
    *out = nullptr;
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    sprs_diag *P = new sprs_diag();
    P->ctx = c; P->n = n; P->t_dtype = t_dtype; P->v_complex = is_complex<V>::value ? 1 : 0;
    auto fail = [&](int st) { sprs_diag_precond_destroy(P); return st; };
    V *tmp = nullptr;
    const size_t np = ((n + 31) & ~(size_t)31) + 32;
    if (hipMalloc(&P->dinv, sizeof(V) * np) != hipSuccess) return fail(SPRS_ERR_HIP);
    if (hipMalloc((void **)&tmp, sizeof(V) * np) != hipSuccess) return fail(SPRS_ERR_HIP);
    int st = SPRS_OK;
    if (hipMemcpyAsync(tmp, diag_host, sizeof(V) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) st = SPRS_ERR_HIP;
    if (st == SPRS_OK) st = launch_diag_inv<V>(c, n, tmp, (V *)P->dinv);   // precond.rs:22-24
    if (st == SPRS_OK && hipStreamSynchronize(c->stream) != hipSuccess) st = SPRS_ERR_HIP;
    (void)hipFree(tmp);
    if (st != SPRS_OK) return fail(st);
    *out = P;
    return SPRS_OK;
}

template <class T>
int diag_apply_dev(const sprs_diag *P, const T *in, T *out) {
    if (!P || !in || !out) return SPRS_INVALID_ARGUMENT;
    if (P->t_dtype != dtype_of<T>::value) return SPRS_INVALID_ARGUMENT;
    if (P->v_complex) {
        if constexpr (is_complex<T>::value) return launch_diag_apply<T, T>(P->ctx, P->n, (const T *)P->dinv, in, out);
        else return SPRS_INVALID_ARGUMENT;
    }
    return launch_diag_apply<T, Real<T>>(P->ctx, P->n, (const Real<T> *)P->dinv, in, out);
}

template <class T>
int diag_apply_host(const sprs_diag *Pc, const T *in, size_t in_len, T *out, size_t out_len) {
    if (!Pc || !in || !out) return SPRS_INVALID_ARGUMENT;
    if (Pc->n != in_len || Pc->n != out_len) return SPRS_DIM_MISMATCH;     // precond.rs:39-41
    sprs_diag *P = const_cast<sprs_diag *>(Pc);
    sprs_ctx *c = P->ctx;
    CtxLock lock(c);   // in_tmp / out_tmp are per-handle staging
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    if (!P->in_tmp) SPRS_HIP_TRY(c, hipMalloc(&P->in_tmp, sizeof(T) * (P->n + 2)));
    if (!P->out_tmp) SPRS_HIP_TRY(c, hipMalloc(&P->out_tmp, sizeof(T) * (P->n + 2)));
    SPRS_HIP_TRY(c, hipMemcpyAsync(P->in_tmp, in, sizeof(T) * in_len, hipMemcpyHostToDevice, c->stream));
    SPRS_TRY(diag_apply_dev<T>(P, (const T *)P->in_tmp, (T *)P->out_tmp));
    SPRS_HIP_TRY(c, hipMemcpyAsync(out, P->out_tmp, sizeof(T) * out_len, hipMemcpyDeviceToHost, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPRS_OK;
}

// ------------------------------------------------------------------------------ solver handles
// one opaque handle layout for the three solver kinds: { dtype, impl }
template <template <class> class S, class F>
int with_solver(int dtype, void *impl, F &&f) {
    switch (dtype) {
        case DT_D: return f((S<double> *)impl);
        case DT_Z: return f((S<cplx> *)impl);
        case DT_S: return f((S<float> *)impl);
        case DT_C: return f((S<cplxf> *)impl);
    }
    return SPRS_INVALID_ARGUMENT;
}

template <class T, class H, template <class> class S, class Mk>
int solver_create(const sprs_csr *A, size_t size, H **out, Mk mk) {
    if (!A || !out) return SPRS_INVALID_ARGUMENT;
    *out = nullptr;
    if (A->dtype != dtype_of<T>::value) return SPRS_INVALID_ARGUMENT;
    // the solvers multiply size-vectors by A in place: the reference leaves a mismatch to UB
    // (mul_vec_unchecked); we refuse it here rather than index out of bounds on the GPU
    if ((int64_t)size != A->nrows || (A->dist ? A->ncols < A->nrows : (int64_t)size != A->ncols)) return SPRS_DIM_MISMATCH;
    H *h = new H();
    h->dtype = dtype_of<T>::value;
    auto *s = new S<T>();
    h->impl = s;
    int st = mk(s);
    if (st != SPRS_OK) { s->destroy(); delete s; delete h; return st; }
    *out = h;
    return SPRS_OK;
}

template <class H, template <class> class S>
int solver_destroy(H *h) {
    if (!h) return SPRS_OK;
    with_solver<S>(h->dtype, h->impl, [](auto *s) { (void)hipStreamSynchronize(s->ctx->stream); s->destroy(); delete s; return (int)SPRS_OK; });
    delete h;
    return SPRS_OK;
}

// host-slice solve: copy rhs/x in, run the device solve, copy x back
template <class T, class SolverT>
int solve_host(SolverT *s, const sprs_diag *P, const T *rhs, size_t rl, T *x, size_t xl, size_t max_iter, Real<T> tol,
               size_t *its, Real<T> *res) {
    if (!s || !rhs || !x) return SPRS_INVALID_ARGUMENT;
    CtxLock lock(s->ctx);   // one solve at a time per context (its stream, its scratch)
    return s->solve_host(rhs, rl, x, xl, [&](T *drhs, T *dx) {
        return s->solve_dev(P, drhs, rl, dx, xl, max_iter, tol, its, res);
    });
}

// device-vector solve; stage through aligned buffers when the caller's vectors are not 16-byte aligned
template <class T, class SolverT>
int solve_dev(SolverT *s, const sprs_diag *P, const T *rhs, size_t rl, T *x, size_t xl, size_t max_iter, Real<T> tol,
              size_t *its, Real<T> *res) {
    if (!s || !rhs || !x) return SPRS_INVALID_ARGUMENT;
    if (rl != s->n) return SPRS_INCOMPATIBLE_RHS_SIZE;
    if (xl != s->n) return SPRS_INCOMPATIBLE_X_SIZE;
    sprs_ctx *c = s->ctx;
    CtxLock lock(c);
    const bool al = ((reinterpret_cast<uintptr_t>(rhs) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
    if (al) return s->solve_dev(P, rhs, rl, x, xl, max_iter, tol, its, res);
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    if (!s->rhs_buf) SPRS_HIP_TRY(c, hipMalloc((void **)&s->rhs_buf, sizeof(T) * s->stride));
    if (!s->x_buf) SPRS_HIP_TRY(c, hipMalloc((void **)&s->x_buf, sizeof(T) * s->stride));
    SPRS_HIP_TRY(c, hipMemcpyAsync(s->rhs_buf, rhs, sizeof(T) * rl, hipMemcpyDeviceToDevice, c->stream));
    SPRS_HIP_TRY(c, hipMemcpyAsync(s->x_buf, x, sizeof(T) * xl, hipMemcpyDeviceToDevice, c->stream));
    int st = s->solve_dev(P, s->rhs_buf, rl, s->x_buf, xl, max_iter, tol, its, res);
    if (st >= SPRS_ERR_HIP) return st;
    SPRS_HIP_TRY(c, hipMemcpyAsync(x, s->x_buf, sizeof(T) * xl, hipMemcpyDeviceToDevice, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return st;
}

template <class T, class H> BicgStab<T> *bi(H *h) { return (h && h->dtype == dtype_of<T>::value) ? (BicgStab<T> *)h->impl : nullptr; }
template <class T, class H> MinRes<T> *mr(H *h) { return (h && h->dtype == dtype_of<T>::value) ? (MinRes<T> *)h->impl : nullptr; }

}  // namespace

struct sprs_bicgstab; struct sprs_minres; struct sprs_csminres;
static int base_of(void *solver, int kind, int *dtype, void **impl) {
    if (!solver) return SPRS_INVALID_ARGUMENT;
    if (kind == SPRS_SOLVER_BICGSTAB) { auto *h = (sprs_bicgstab *)solver; *dtype = h->dtype; *impl = h->impl; }
    else if (kind == SPRS_SOLVER_MINRES) { auto *h = (sprs_minres *)solver; *dtype = h->dtype; *impl = h->impl; }
    else if (kind == SPRS_SOLVER_CSMINRES) { auto *h = (sprs_csminres *)solver; *dtype = h->dtype; *impl = h->impl; }
    else return SPRS_INVALID_ARGUMENT;
    return SPRS_OK;
}
// every solver impl derives from KrylovBase<T>
template <class F>
static int with_base(void *solver, int kind, F &&f) {
    int dt; void *impl;
    SPRS_TRY(base_of(solver, kind, &dt, &impl));
    if (kind == SPRS_SOLVER_BICGSTAB)
        return with_solver<BicgStab>(dt, impl, [&](auto *s) { return f(s); });
    return with_solver<MinRes>(dt, impl, [&](auto *s) { return f(s); });
}


#define SPRS_CHK(c) do { if (!(c)) return SPRS_INVALID_ARGUMENT; } while (0)
#define SPRS_G(...) try { __VA_ARGS__ } catch (...) { return SPRS_ERR_HIP; }

extern "C" {

int sprs_csr_destroy(sprs_csr *A) {
    if (!A) return SPRS_OK;
    if (A->ctx) { (void)hipSetDevice(A->ctx->device); (void)hipStreamSynchronize(A->ctx->stream); }
    if (A->owns_arrays) {
        if (A->row_ptr) (void)hipFree(A->row_ptr);
        if (A->col_idx) (void)hipFree(A->col_idx);
        if (A->val) (void)hipFree(A->val);
    }
    if (A->rowblk) (void)hipFree(A->rowblk);
    if (A->blk_desc) (void)hipFree(A->blk_desc);
    if (A->blk_desc_eq) (void)hipFree(A->blk_desc_eq);
    if (A->tail) (void)hipFree(A->tail);
    free_dict(A);
    if (A->x_tmp) (void)hipFree(A->x_tmp);
    if (A->y_tmp) (void)hipFree(A->y_tmp);
    if (A->part) (void)hipFree(A->part);
    if (A->dist) {
        if (A->dist->send_idx) (void)hipFree(A->dist->send_idx);
        if (A->dist->send_buf) (void)hipFree(A->dist->send_buf);
        if (A->dist->ag_buf) (void)hipFree(A->dist->ag_buf);
        if (A->dist->order_int) (void)hipFree(A->dist->order_int);
        if (A->dist->order_bnd) (void)hipFree(A->dist->order_bnd);
        for (void *q : {(void *)A->dist->tile_int.list, (void *)A->dist->tile_int.xstart, (void *)A->dist->tile_int.left}) if (q) (void)hipFree(q);
        if (A->dist->order_int_w) (void)hipFree(A->dist->order_int_w);
        if (A->dist->order_bnd_w) (void)hipFree(A->dist->order_bnd_w);
        if (A->dist->ev_pack) (void)hipEventDestroy(A->dist->ev_pack);
        if (A->dist->ev_halo) (void)hipEventDestroy(A->dist->ev_halo);
        if (A->dist->comm_stream) (void)hipStreamDestroy(A->dist->comm_stream);
        delete A->dist;
    }
    delete A;
    return SPRS_OK;
}
int64_t sprs_csr_rows(const sprs_csr *A) { return A ? A->nrows : -1; }
int64_t sprs_csr_cols(const sprs_csr *A) { return A ? A->ncols : -1; }
int64_t sprs_csr_nnz(const sprs_csr *A) { return A ? A->nnz : -1; }
int sprs_csr_wide_blocks(const sprs_csr *A, int64_t *n_blocks, int64_t *n_uniform) {
    if (!A || !n_blocks || !n_uniform) return SPRS_INVALID_ARGUMENT;
    *n_blocks = 0; *n_uniform = 0;
    if (dict_mode(A) == 0) {       // plain CSR stream: the 64-row blocks; "uniform" = equal-length blocks, which read no row_ptr
        *n_blocks = A->n_rowblk; *n_uniform = A->blk_desc_eq ? A->n_eq_blocks : 0;
        return SPRS_OK;
    }
    // the descriptors the SpMV of this handle walks: 128-row blocks of the f64 pair-code stream, else the 64-row
    // blocks of the offset-code stream
    const bool wide = A->dict->wide_desc && A->dict->n_wide > 0 && dict_mode(A) == 2 && A->ctx->spmv_wide != 0;
    // (the 64-row pair-code kernel, spmv_wide = 0, walks the plain descriptors: no uniform blocks)
    const bool offs = dict_mode(A) == 1 && A->ctx->spmv_uniform != 0;
    const void *src = wide ? A->dict->wide_desc : (offs ? A->dict->off_desc : nullptr);
    const int64_t nb = wide ? A->dict->n_wide : (src ? A->n_rowblk : 0);
    if (!src || nb == 0) return SPRS_OK;
    sprs_ctx *c = A->ctx;
    CtxLock lock(c);
    std::vector<int32_t> d((size_t)nb * 4);
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    SPRS_HIP_TRY(c, hipMemcpyAsync(d.data(), src, d.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_blocks = nb;
    for (int64_t j = 0; j < nb; ++j) *n_uniform += ((uint32_t)d[(size_t)j * 4 + 1] & sprs::UNI2) != 0;
    return SPRS_OK;
}
int sprs_csr_tile_plan(const sprs_csr *A, int64_t *n_tiles, int64_t *n_tile_blocks, int64_t *n_other_blocks) {
    if (!A || !n_tiles || !n_tile_blocks || !n_other_blocks) return SPRS_INVALID_ARGUMENT;
    *n_tiles = 0; *n_tile_blocks = 0; *n_other_blocks = 0;
    const sprs_dict *D = A->dict;
    if (!D || A->ctx->spmv_tile == 0 || A->ctx->spmv_wide == 0) return SPRS_OK;
    const int dm = dict_mode(A);
    if (dm != 1 && dm != 2) return SPRS_OK;
    if (A->dist && A->dist->order_int) {
        // a distributed operator with an interior / boundary split: the interior launch's plan (boundary blocks are walked singly)
        const sprs_tile_plan &TI = A->dist->tile_int;
        if (TI.n_tile <= 0 || A->dist->tile_int_off != (dm == 1)) return SPRS_OK;
        *n_tiles = TI.n_tile; *n_tile_blocks = (int64_t)TI.n_tile * sprs::tile_blocks();
        *n_other_blocks = (A->n_rowblk + 1) / 2 - *n_tile_blocks;
        return SPRS_OK;
    }
    if (sprs::chain_plan_used(A)) return SPRS_OK;          // the chains run this handle's SpMV (sprs_csr_chain_plan)
    const sprs_tile_plan &TP = dm == 2 ? D->tile_pair : D->tile_off;
    if (TP.n_tile <= 0 || !sprs::tile_plan_used(A)) return SPRS_OK;
    // the other blocks in 128-row units (the offset stream walks them as 64-row blocks)
    *n_tiles = TP.n_tile; *n_tile_blocks = (int64_t)TP.n_tile * sprs::tile_blocks(); *n_other_blocks = dm == 2 ? TP.n_left : (TP.n_left + 1) / 2;
    return SPRS_OK;
}
int sprs_csr_chain_plan(const sprs_csr *A, int64_t *n_tiles, int64_t *n_segments, int64_t *n_chains, int64_t *n_other_blocks) {
    if (!A || !n_tiles || !n_segments || !n_chains || !n_other_blocks) return SPRS_INVALID_ARGUMENT;
    *n_tiles = 0; *n_segments = 0; *n_chains = 0; *n_other_blocks = 0;
    if (!A->dict || !sprs::chain_plan_used(A)) return SPRS_OK;
    const sprs_chain_plan &CP = A->dict->chain_pair;
    *n_tiles = CP.n_tile; *n_segments = CP.n_seg; *n_chains = CP.n_chain; *n_other_blocks = CP.n_left;
    return SPRS_OK;
}
int sprs_csr_stream_format(const sprs_csr *A, int *n_offsets, int *n_values) {
    if (!A) return -1;
    if (n_offsets) *n_offsets = A->dict ? A->dict->n_off : 0;
    if (n_values) *n_values = A->dict ? (A->dict->n_pair ? A->dict->n_pair : 0) : 0;
    return dict_mode(A);
}

int sprs_diag_precond_destroy(sprs_diag *P) {
    if (!P) return SPRS_OK;
    if (P->ctx) { (void)hipSetDevice(P->ctx->device); (void)hipStreamSynchronize(P->ctx->stream); }
    if (P->dinv) (void)hipFree(P->dinv);
    if (P->in_tmp) (void)hipFree(P->in_tmp);
    if (P->out_tmp) (void)hipFree(P->out_tmp);
    delete P;
    return SPRS_OK;
}
int sprs_bicgstab_destroy(sprs_bicgstab *S) { return solver_destroy<sprs_bicgstab, BicgStab>(S); }
int sprs_minres_destroy(sprs_minres *S) { return solver_destroy<sprs_minres, MinRes>(S); }
int sprs_csminres_destroy(sprs_csminres *S) { return solver_destroy<sprs_csminres, MinRes>(S); }

// ---- everything that exists once per scalar type.  X = suffix, T = device scalar, CT = C-ABI scalar
// (passed by value / pointer), R = T::Real
#define SPRS_API(X, T, CT, R)                                                                                          \
    int sprs_csr_create_##X(sprs_ctx *c, int64_t nr, int64_t nc, int64_t nnz, const int32_t *rp, const int32_t *ci,    \
                            const CT *v, int csc, sprs_csr **out) {                                                    \
        SPRS_G(return csr_create_host<T, int32_t>(c, nr, nc, nnz, rp, ci, (const T *)v, csc, out);)                    \
    }                                                                                                                  \
    int sprs_csr_create_i64_##X(sprs_ctx *c, int64_t nr, int64_t nc, int64_t nnz, const int64_t *rp,                   \
                                const int64_t *ci, const CT *v, int csc, sprs_csr **out) {                             \
        SPRS_G(return csr_create_host<T, int64_t>(c, nr, nc, nnz, rp, ci, (const T *)v, csc, out);)                    \
    }                                                                                                                  \
    int sprs_csr_create_dev_##X(sprs_ctx *c, int64_t nr, int64_t nc, int64_t nnz, const int32_t *rp,                   \
                                const int32_t *ci, const CT *v, int adopt, sprs_csr **out) {                           \
        SPRS_G(return csr_create_dev<T>(c, nr, nc, nnz, rp, ci, (const T *)v, adopt, out);)                            \
    }                                                                                                                  \
    int sprs_mul_vec_##X(const sprs_csr *A, const CT *x, size_t xl, CT *y, size_t yl) {                                \
        SPRS_G(return mul_vec_host<T>(A, (const T *)x, xl, (T *)y, yl, nullptr);)                                      \
    }                                                                                                                  \
    int sprs_mul_vec_dot_##X(const sprs_csr *A, const CT *x, size_t xl, CT *y, size_t yl, CT *d) {                     \
        SPRS_G(if (!d) return SPRS_INVALID_ARGUMENT; return mul_vec_host<T>(A, (const T *)x, xl, (T *)y, yl, (T *)d);) \
    }                                                                                                                  \
    int sprs_mul_vec_dev_##X(const sprs_csr *A, const CT *x, CT *y) { return mul_vec_dev<T>(A, (const T *)x, (T *)y, nullptr); } \
    int sprs_mul_vec_dot_dev_##X(const sprs_csr *A, const CT *x, CT *y, CT *d) {                                       \
        if (!d) return SPRS_INVALID_ARGUMENT;                                                                          \
        return mul_vec_dev<T>(A, (const T *)x, (T *)y, (T *)d);                                                        \
    }                                                                                                                  \
    int sprs_mul_vec_dev_timed_##X(const sprs_csr *A, const CT *x, CT *y, int reps, double *ms) {                      \
        return mul_vec_timed<T>(A, (const T *)x, (T *)y, reps, ms);                                                    \
    }                                                                                                                  \
    int sprs_dot_##X(sprs_ctx *c, size_t n, const CT *x, const CT *y, CT *o) { SPRS_CHK(c && o); return dot_host<T>(c, n, (const T *)x, (const T *)y, false, (T *)o); } \
    int sprs_conj_dot_##X(sprs_ctx *c, size_t n, const CT *x, const CT *y, CT *o) { SPRS_CHK(c && o); return dot_host<T>(c, n, (const T *)x, (const T *)y, true, (T *)o); } \
    int sprs_norm2_##X(sprs_ctx *c, size_t n, const CT *x, R *o) { SPRS_CHK(c && o); return norm2_host<T>(c, n, (const T *)x, o); } \
    int sprs_scale_##X(sprs_ctx *c, size_t n, CT a, CT *x) { SPRS_CHK(c); T aa; memcpy(&aa, &a, sizeof(T)); return launch_scale<T>(c, n, aa, (T *)x); } \
    int sprs_rscale_##X(sprs_ctx *c, size_t n, R a, CT *x) { SPRS_CHK(c); return launch_rscale<T>(c, n, a, (T *)x); }  \
    int sprs_conj_##X(sprs_ctx *c, size_t n, const CT *in, CT *out) { SPRS_CHK(c); return launch_conj<T>(c, n, (const T *)in, (T *)out); } \
    int sprs_axpy_##X(sprs_ctx *c, size_t n, CT a, const CT *x, CT *y) { SPRS_CHK(c); T aa; memcpy(&aa, &a, sizeof(T)); return launch_axpy<T, T>(c, n, aa, (const T *)x, (T *)y); } \
    int sprs_axpby_##X(sprs_ctx *c, size_t n, CT a, const CT *x, CT b, CT *y) {                                        \
        SPRS_CHK(c); T aa, bb; memcpy(&aa, &a, sizeof(T)); memcpy(&bb, &b, sizeof(T));                                 \
        return launch_axpby<T>(c, n, aa, (const T *)x, bb, (T *)y);                                                    \
    }                                                                                                                  \
    int sprs_diag_mul_vec_##X(const sprs_diag *P, const CT *in, size_t il, CT *out, size_t ol) {                       \
        SPRS_G(return diag_apply_host<T>(P, (const T *)in, il, (T *)out, ol);)                                         \
    }                                                                                                                  \
    int sprs_diag_mul_vec_dev_##X(const sprs_diag *P, const CT *in, CT *out) { return diag_apply_dev<T>(P, (const T *)in, (T *)out); } \
    int sprs_bicgstab_create_##X(const sprs_csr *A, size_t n, sprs_bicgstab **out) {                                   \
        SPRS_G(return (solver_create<T, sprs_bicgstab, BicgStab>(A, n, out, [&](auto *s) { return s->create(A, n); }));) \
    }                                                                                                                  \
    int sprs_minres_create_##X(const sprs_csr *A, size_t n, sprs_minres **out) {                                       \
        SPRS_G(return (solver_create<T, sprs_minres, MinRes>(A, n, out, [&](auto *s) { return s->create(A, n, false); }));) \
    }                                                                                                                  \
    int sprs_csminres_create_##X(const sprs_csr *A, size_t n, sprs_csminres **out) {                                   \
        SPRS_G(return (solver_create<T, sprs_csminres, MinRes>(A, n, out, [&](auto *s) { return s->create(A, n, true); }));) \
    }                                                                                                                  \
    int sprs_bicgstab_solve_##X(sprs_bicgstab *S, const CT *rhs, size_t rl, CT *x, size_t xl, size_t mi, R tol, size_t *its, R *res) { \
        SPRS_G(return solve_host<T>(bi<T>(S), nullptr, (const T *)rhs, rl, (T *)x, xl, mi, tol, its, res);)            \
    }                                                                                                                  \
    int sprs_bicgstab_precond_solve_##X(sprs_bicgstab *S, const sprs_diag *P, const CT *rhs, size_t rl, CT *x, size_t xl, size_t mi, R tol, size_t *its, R *res) { \
        SPRS_G(if (!P) return SPRS_INVALID_ARGUMENT; return solve_host<T>(bi<T>(S), P, (const T *)rhs, rl, (T *)x, xl, mi, tol, its, res);) \
    }                                                                                                                  \
    int sprs_bicgstab_solve_dev_##X(sprs_bicgstab *S, const sprs_diag *P, const CT *rhs, size_t rl, CT *x, size_t xl, size_t mi, R tol, size_t *its, R *res) { \
        SPRS_G(return solve_dev<T>(bi<T>(S), P, (const T *)rhs, rl, (T *)x, xl, mi, tol, its, res);)                   \
    }                                                                                                                  \
    int sprs_minres_solve_##X(sprs_minres *S, const CT *rhs, size_t rl, CT *x, size_t xl, size_t mi, R tol, size_t *its, R *res) { \
        SPRS_G(return solve_host<T>(mr<T>(S), nullptr, (const T *)rhs, rl, (T *)x, xl, mi, tol, its, res);)            \
    }                                                                                                                  \
    int sprs_minres_precond_solve_##X(sprs_minres *S, const sprs_diag *P, const CT *rhs, size_t rl, CT *x, size_t xl, size_t mi, R tol, size_t *its, R *res) { \
        SPRS_G(if (!P) return SPRS_INVALID_ARGUMENT; return solve_host<T>(mr<T>(S), P, (const T *)rhs, rl, (T *)x, xl, mi, tol, its, res);) \
    }                                                                                                                  \
    int sprs_minres_solve_dev_##X(sprs_minres *S, const sprs_diag *P, const CT *rhs, size_t rl, CT *x, size_t xl, size_t mi, R tol, size_t *its, R *res) { \
        SPRS_G(return solve_dev<T>(mr<T>(S), P, (const T *)rhs, rl, (T *)x, xl, mi, tol, its, res);)                   \
    }                                                                                                                  \
    int sprs_csminres_solve_##X(sprs_csminres *S, const CT *rhs, size_t rl, CT *x, size_t xl, size_t mi, R tol, size_t *its, R *res) { \
        SPRS_G(return solve_host<T>(mr<T>(S), nullptr, (const T *)rhs, rl, (T *)x, xl, mi, tol, its, res);)            \
    }                                                                                                                  \
    int sprs_csminres_solve_dev_##X(sprs_csminres *S, const CT *rhs, size_t rl, CT *x, size_t xl, size_t mi, R tol, size_t *its, R *res) { \
        SPRS_G(return solve_dev<T>(mr<T>(S), nullptr, (const T *)rhs, rl, (T *)x, xl, mi, tol, its, res);)             \
    }

SPRS_API(d, double, double, double)
SPRS_API(z, cplx, sprs_c64, double)
SPRS_API(s, float, float, float)
SPRS_API(c, cplxf, sprs_c32, float)

// complex vector, real scalar (S = Real, T = Complex; vecalg.rs:746-757)
int sprs_axpy_zd(sprs_ctx *c, size_t n, double a, const sprs_c64 *x, sprs_c64 *y) { SPRS_CHK(c); return launch_axpy<cplx, double>(c, n, a, (const cplx *)x, (cplx *)y); }
int sprs_axpy_cs(sprs_ctx *c, size_t n, float a, const sprs_c32 *x, sprs_c32 *y) { SPRS_CHK(c); return launch_axpy<cplxf, float>(c, n, a, (const cplxf *)x, (cplxf *)y); }

// DiagPrecond<T, V>::new  (precond.rs:20-29)
int sprs_diag_precond_create_d(sprs_ctx *c, size_t n, const double *d, sprs_diag **out) { SPRS_G(return diag_create<double>(c, n, d, DT_D, out);) }
int sprs_diag_precond_create_zd(sprs_ctx *c, size_t n, const double *d, sprs_diag **out) { SPRS_G(return diag_create<double>(c, n, d, DT_Z, out);) }
int sprs_diag_precond_create_z(sprs_ctx *c, size_t n, const sprs_c64 *d, sprs_diag **out) { SPRS_G(return diag_create<cplx>(c, n, (const cplx *)d, DT_Z, out);) }
int sprs_diag_precond_create_s(sprs_ctx *c, size_t n, const float *d, sprs_diag **out) { SPRS_G(return diag_create<float>(c, n, d, DT_S, out);) }
int sprs_diag_precond_create_cs(sprs_ctx *c, size_t n, const float *d, sprs_diag **out) { SPRS_G(return diag_create<float>(c, n, d, DT_C, out);) }
int sprs_diag_precond_create_c(sprs_ctx *c, size_t n, const sprs_c32 *d, sprs_diag **out) { SPRS_G(return diag_create<cplxf>(c, n, (const cplxf *)d, DT_C, out);) }

// ------------------------------------------------------------------------------ options / instrumentation
int sprs_solver_set_mode(void *solver, int kind, int mode) {
    if (mode != 0 && mode != 1) return SPRS_INVALID_ARGUMENT;
    return with_base(solver, kind, [&](auto *b) { b->mode = mode; return (int)SPRS_OK; });
}
int sprs_solver_set_trace(void *solver, int kind, double *trace_host, size_t cap) {
    return with_base(solver, kind, [&](auto *b) { b->trace = trace_host; b->trace_cap = trace_host ? cap : 0; b->trace_rows = 0; return (int)SPRS_OK; });
}
int sprs_solver_trace_rows(const void *solver, int kind, size_t *rows_out) {
    if (!rows_out) return SPRS_INVALID_ARGUMENT;
    return with_base(const_cast<void *>(solver), kind, [&](auto *b) { *rows_out = b->trace_rows; return (int)SPRS_OK; });
}
int sprs_solver_set_profile(void *solver, int kind, int enable) {
    return with_base(solver, kind, [&](auto *b) { b->profile = enable < 0 ? 0 : enable; return (int)SPRS_OK; });
}
int sprs_solver_get_profile(const void *solver, int kind, double *spmv_ms, int64_t *launches, double *solve_ms) {
    return with_base(const_cast<void *>(solver), kind, [&](auto *b) {
        if (spmv_ms) *spmv_ms = b->stats.spmv_ms;
        if (launches) *launches = b->stats.spmv_launches;
        if (solve_ms) *solve_ms = b->stats.solve_ms;
        return (int)SPRS_OK;
    });
}

int sprs_solver_get_profile_counts(const void *solver, int kind, int64_t *steps, int64_t *timed_dot_other, int64_t *timed_k2_fused,
                                   int64_t *timed_k4_fused) {
    return with_base(const_cast<void *>(solver), kind, [&](auto *b) {
        if (steps) *steps = b->stats.steps;
        if (timed_dot_other) *timed_dot_other = b->stats.timed_dot_other;
        if (timed_k2_fused) *timed_k2_fused = b->stats.timed_fused_k2;
        if (timed_k4_fused) *timed_k4_fused = b->stats.timed_fused_k4;
        return (int)SPRS_OK;
    });
}

int sprs_solver_get_fused_launches(const void *solver, int kind, int64_t *k2_fused, int64_t *k4_fused) {
    return with_base(const_cast<void *>(solver), kind, [&](auto *b) {
        if (k2_fused) *k2_fused = b->stats.fused_k2;
        if (k4_fused) *k4_fused = b->stats.fused_k4;
        return (int)SPRS_OK;
    });
}

}  // extern "C"
