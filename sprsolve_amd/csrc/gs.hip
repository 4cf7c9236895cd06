// Gauss-Seidel (reference src/gauss_seidel.rs) made data-parallel WITHOUT changing its arithmetic.
//
// The reference sweeps the rows in order; row i uses the already updated x_j for j < i and the
// previous sweep's x_j for j > i.  Rows are grouped into dependency levels at handle creation
// (level(i) = 1 + max level of the rows j < i that row i references); the rows of one level are
// independent and are updated by one kernel launch, each row by one lane that folds its entries
// left to right exactly as the reference does (`sigma += val * x[col]`, gauss_seidel.rs:116).
// The sweep writes into a second vector (x_new) and reads x_old for the entries above the
// diagonal, so a row of an earlier level never sees a value the serial sweep would not have seen.
// x after k sweeps is therefore BIT-IDENTICAL to the reference's; only the residual norm
// (a reduction) differs in summation order.  A 3-D stencil of 50 M rows has ~1200 levels of up
// to ~10^5 rows; a 2-D one has ~2n levels of <= n rows (little parallelism, still exact); a chain (tridiagonal
// matrix) has n levels of one row each — one launch per row: exact, and pointless on a GPU.
#include <algorithm>

#include "device.hpp"

using namespace sprs;

struct sprs_gauss_seidel {
    const sprs_csr *A = nullptr;   // borrowed (the caller keeps it alive while solving, like the reference's view)
    sprs_ctx *ctx = nullptr;
    int dtype = 0;
    int64_t n = 0;
    std::vector<int32_t> lvl_ptr;      // host: level l owns rows lvl_rows[lvl_ptr[l] .. lvl_ptr[l+1])
    int32_t *lvl_rows = nullptr;       // device
    void *diag = nullptr;              // device, T   (workspace[n..2n) of the reference, :81)
    void *resid = nullptr;             // device, T   (workspace[0..n), :90)
    void *x_a = nullptr, *x_b = nullptr, *rhs_buf = nullptr;   // device, T
    int *d_bad = nullptr;              // device: smallest row with a missing / too small diagonal
    hipGraphExec_t sweep_graph[2] = {nullptr, nullptr};   // [0]: x_a -> x_b, [1]: x_b -> x_a (rhs = rhs_buf)
};

namespace {

// one lane per row of the level; rows >= row_limit are left untouched (mimics the reference's early
// return from the unrolled first sweep: rows before the offending one have already been updated)
template <class T, bool FIRST>
__global__ __launch_bounds__(BLOCK) void gs_level_kernel(int count, const int32_t *__restrict__ rows, int row_limit,
                                                         const int32_t *__restrict__ row_ptr,
                                                         const int32_t *__restrict__ col_idx, const T *__restrict__ val,
                                                         const T *__restrict__ rhs, T *__restrict__ diag,
                                                         const T *__restrict__ x_old, T *__restrict__ x_new) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const int row = rows[i];
    if (row >= row_limit) return;
    T sigma = szero<T>();
    T dg = FIRST ? szero<T>() : diag[row];
    for (int k = row_ptr[row]; k < row_ptr[row + 1]; ++k) {
        const int c = col_idx[k];
        if (c != row) sigma = sadd(sigma, smul(val[k], c < row ? x_new[c] : x_old[c]));   // :66 / :116
        else if (FIRST) dg = val[k];                                                     // :69
    }
    if (FIRST) diag[row] = dg;                                                           // :81
    x_new[row] = sdiv(ssub(rhs[row], sigma), dg);                                        // :84 / :123
}

// smallest row whose diagonal is missing or has |diag|^2 < eps (:72-78)
template <class T>
__global__ __launch_bounds__(BLOCK) void gs_diag_check_kernel(int n, const int32_t *__restrict__ row_ptr,
                                                              const int32_t *__restrict__ col_idx, const T *__restrict__ val,
                                                              int *__restrict__ bad) {
    for (int row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK) {
        bool have = false; T dg = szero<T>();
        for (int k = row_ptr[row]; k < row_ptr[row + 1]; ++k)
            if (col_idx[k] == row) { have = true; dg = val[k]; }          // the last one wins, as in the reference
        if (!have || ssq(dg) < seps<Real<T>>()) atomicMin(bad, row);
    }
}

template <class T>
int gs_create(const sprs_csr *A, sprs_gauss_seidel **out) {
    sprs_ctx *c = A->ctx;
    const int64_t n = A->nrows;
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    // dependency levels need the pattern on the host
    std::vector<int32_t> rp((size_t)n + 1), ci((size_t)A->nnz);
    SPRS_HIP_TRY(c, hipMemcpyAsync(rp.data(), A->row_ptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, c->stream));
    if (A->nnz) SPRS_HIP_TRY(c, hipMemcpyAsync(ci.data(), A->col_idx, sizeof(int32_t) * ci.size(), hipMemcpyDeviceToHost, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<int32_t> level((size_t)n, 0);
    int32_t nlev = n ? 1 : 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t l = 0;
        for (int32_t k = rp[i]; k < rp[i + 1]; ++k)
            if (ci[k] < i) l = std::max(l, level[ci[k]] + 1);
        level[i] = l;
        nlev = std::max(nlev, l + 1);
    }
    auto *G = new sprs_gauss_seidel();
    G->A = A; G->ctx = c; G->dtype = A->dtype; G->n = n;
    G->lvl_ptr.assign((size_t)nlev + 1, 0);
    for (int64_t i = 0; i < n; ++i) G->lvl_ptr[(size_t)level[i] + 1]++;
    for (int32_t l = 0; l < nlev; ++l) G->lvl_ptr[l + 1] += G->lvl_ptr[l];
    std::vector<int32_t> rows((size_t)n), fill(G->lvl_ptr.begin(), G->lvl_ptr.end() - (nlev ? 1 : 0));
    for (int64_t i = 0; i < n; ++i) rows[(size_t)fill[level[i]]++] = (int32_t)i;     // ascending rows inside a level
    const size_t np = (size_t)n + 32;
    bool ok = hipMalloc((void **)&G->lvl_rows, sizeof(int32_t) * np) == hipSuccess &&
              hipMalloc(&G->diag, sizeof(T) * np) == hipSuccess && hipMalloc(&G->resid, sizeof(T) * np) == hipSuccess &&
              hipMalloc(&G->x_a, sizeof(T) * np) == hipSuccess && hipMalloc(&G->x_b, sizeof(T) * np) == hipSuccess &&
              hipMalloc(&G->rhs_buf, sizeof(T) * np) == hipSuccess && hipMalloc((void **)&G->d_bad, sizeof(int)) == hipSuccess &&
              (n == 0 || hipMemcpy(G->lvl_rows, rows.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess);
    if (!ok) { sprs_gauss_seidel_destroy(G); return SPRS_ERR_HIP; }
    *out = G;
    return SPRS_OK;
}

template <class T, bool FIRST>
int gs_sweep(sprs_gauss_seidel *G, int row_limit, const T *rhs, const T *x_old, T *x_new) {
    sprs_ctx *c = G->A->ctx;
    const T *val = reinterpret_cast<const T *>(G->A->val);
    for (size_t l = 0; l + 1 < G->lvl_ptr.size(); ++l) {
        const int cnt = G->lvl_ptr[l + 1] - G->lvl_ptr[l];
        if (!cnt) continue;
        hipLaunchKernelGGL((gs_level_kernel<T, FIRST>), dim3((cnt + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, cnt,
                           G->lvl_rows + G->lvl_ptr[l], row_limit, G->A->row_ptr, G->A->col_idx, val, rhs, (T *)G->diag,
                           x_old, x_new);
    }
    SPRS_HIP_TRY(c, hipGetLastError());
    return SPRS_OK;
}

// One later sweep (gauss_seidel.rs:110-126) between the handle's own buffers.  A sweep is ~10^3 small launches
// (one per level), so it is captured once per direction into a hipGraph and replayed.
template <class T>
int gs_sweep_buffers(sprs_gauss_seidel *G, int dir) {
    sprs_ctx *c = G->ctx;
    const T *x_old = (const T *)(dir == 0 ? G->x_a : G->x_b);
    T *x_new = (T *)(dir == 0 ? G->x_b : G->x_a);
    if (!c->gs_graph || G->lvl_ptr.size() < 8) return gs_sweep<T, false>(G, INT32_MAX, (const T *)G->rhs_buf, x_old, x_new);
    if (!G->sweep_graph[dir]) {
        hipGraph_t g = nullptr;
        SPRS_HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int st = gs_sweep<T, false>(G, INT32_MAX, (const T *)G->rhs_buf, x_old, x_new);
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (st != SPRS_OK) { if (g) (void)hipGraphDestroy(g); return st; }
        SPRS_HIP_TRY(c, e);
        e = hipGraphInstantiate(&G->sweep_graph[dir], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        SPRS_HIP_TRY(c, e);
    }
    SPRS_HIP_TRY(c, hipGraphLaunch(G->sweep_graph[dir], c->stream));
    return SPRS_OK;
}

// device solve: rhs / x device vectors of n elements
template <class T>
int gs_solve_dev(sprs_gauss_seidel *G, const T *rhs, T *x, size_t max_iter, Real<T> eps, size_t *its_out, Real<T> *res_out) {
    using R = Real<T>;
    sprs_ctx *c = G->A->ctx;
    CtxLock lock(c);
    const size_t n = (size_t)G->n;
    *its_out = 0; *res_out = 0;
    if (max_iter == 0) return SPRS_INSUFFICIENT_ITER;                       // :52-54
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    T *xa = (T *)G->x_a, *xb = (T *)G->x_b, *res_v = (T *)G->resid;
    if (rhs != (const T *)G->rhs_buf) {      // the captured sweeps read rhs from the handle's buffer
        SPRS_HIP_TRY(c, hipMemcpyAsync(G->rhs_buf, rhs, sizeof(T) * n, hipMemcpyDeviceToDevice, c->stream));
        rhs = (const T *)G->rhs_buf;
    }
    // ZeorDiagonalElem(row): the reference fails inside the first sweep, after rows < row were updated
    int bad = INT32_MAX;
    SPRS_HIP_TRY(c, hipMemcpyAsync(G->d_bad, &bad, sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (n) hipLaunchKernelGGL((gs_diag_check_kernel<T>), dim3((int)std::min<size_t>((n + BLOCK - 1) / BLOCK, 2048)), dim3(BLOCK), 0,
                              c->stream, (int)n, G->A->row_ptr, G->A->col_idx, reinterpret_cast<const T *>(G->A->val), G->d_bad);
    SPRS_HIP_TRY(c, hipMemcpyAsync(&bad, G->d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int row_limit = bad == INT32_MAX ? INT32_MAX : bad;
    // the sweep reads the previous values from x_old and writes x_new; both start as the caller's x
    SPRS_HIP_TRY(c, hipMemcpyAsync(xa, x, sizeof(T) * n, hipMemcpyDeviceToDevice, c->stream));
    SPRS_HIP_TRY(c, hipMemcpyAsync(xb, x, sizeof(T) * n, hipMemcpyDeviceToDevice, c->stream));
    T *x_old = xa, *x_new = xb;
    SPRS_TRY((gs_sweep<T, true>(G, row_limit, rhs, x_old, x_new)));         // :60-86
    if (bad != INT32_MAX) {
        SPRS_HIP_TRY(c, hipMemcpyAsync(x, x_new, sizeof(T) * n, hipMemcpyDeviceToDevice, c->stream));
        SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        *its_out = (size_t)bad;
        return SPRS_ZERO_DIAGONAL;
    }
    R b_norm = 0;
    SPRS_TRY(norm2_host<T>(c, n, rhs, &b_norm));                            // :83 accumulated, :87 sqrt
    const R tol2 = eps * b_norm;
    auto residual = [&](const T *xx, R *res) -> int {
        SPRS_TRY(launch_spmv<T>(G->A, xx, res_v, 0, nullptr, nullptr, nullptr, nullptr));     // :90 / :128
        SPRS_TRY((launch_axpy<T, T>(c, n, sneg(sone<T>()), rhs, res_v)));                     // :97 / :131
        return norm2_host<T>(c, n, res_v, res);                                               // :104 / :133
    };
    R res = 0;
    SPRS_TRY(residual(x_new, &res));
    size_t it_done = 1;
    int status = SPRS_INSUFFICIENT_ITER;
    if (res <= tol2) { status = SPRS_OK; it_done = 1; }                     // :106-108
    else {
        for (size_t it = 1; it < max_iter; ++it) {                          // :110
            // next sweep: every row is rewritten, so x_new becomes complete again; x_old must hold the
            // previous sweep's values for the entries above the diagonal
            std::swap(x_old, x_new);
            SPRS_TRY(gs_sweep_buffers<T>(G, x_old == xa ? 0 : 1));
            SPRS_TRY(residual(x_new, &res));
            if (res <= tol2) { status = SPRS_OK; it_done = it; break; }     // :135-137
        }
    }
    SPRS_HIP_TRY(c, hipMemcpyAsync(x, x_new, sizeof(T) * n, hipMemcpyDeviceToDevice, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (status == SPRS_OK) { *its_out = it_done; *res_out = res; }
    else *its_out = max_iter;                                               // :139
    return status;
}

template <class T>
int gs_solve_host(sprs_gauss_seidel *G, const T *rhs, size_t rl, T *x, size_t xl, size_t max_iter, Real<T> eps, size_t *its,
                  Real<T> *res) {
    if (!G || !rhs || !x || !its || !res) return SPRS_INVALID_ARGUMENT;
    if (G->dtype != dtype_of<T>::value) return SPRS_INVALID_ARGUMENT;
    if (rl != (size_t)G->n) return SPRS_INCOMPATIBLE_RHS_SIZE;              // :41-45
    if (rl != xl) return SPRS_INCOMPATIBLE_X_SIZE;                          // :46-50
    sprs_ctx *c = G->A->ctx;
    CtxLock lock(c);
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    T *drhs = (T *)G->rhs_buf;
    T *dx = nullptr;
    SPRS_HIP_TRY(c, hipMalloc((void **)&dx, sizeof(T) * (rl + 32)));
    int st = SPRS_ERR_HIP;
    if (hipMemcpyAsync(drhs, rhs, sizeof(T) * rl, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
        hipMemcpyAsync(dx, x, sizeof(T) * xl, hipMemcpyHostToDevice, c->stream) == hipSuccess) {
        st = gs_solve_dev<T>(G, drhs, dx, max_iter, eps, its, res);
        if (st < SPRS_ERR_HIP && (hipMemcpyAsync(x, dx, sizeof(T) * xl, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                                  hipStreamSynchronize(c->stream) != hipSuccess)) st = SPRS_ERR_HIP;
    }
    (void)hipFree(dx);
    return st;
}

}  // namespace

extern "C" {

int sprs_gauss_seidel_create(const sprs_csr *A, sprs_gauss_seidel **out) {
    try {
        if (!A || !out) return SPRS_INVALID_ARGUMENT;
        *out = nullptr;
        if (A->nrows != A->ncols) return SPRS_NOT_SQUARE;                   // gauss_seidel.rs:16-20
        if (A->was_csc) return SPRS_NOT_CSR;                                // :22-26
        if (A->dist) return SPRS_INVALID_ARGUMENT;
        if (A->dtype == DT_D) return gs_create<double>(A, out);
        if (A->dtype == DT_S) return gs_create<float>(A, out);
        return SPRS_INVALID_ARGUMENT;                                       // T: PartialOrd — real scalars only (:8)
    } catch (...) { return SPRS_ERR_HIP; }
}

int sprs_gauss_seidel_destroy(sprs_gauss_seidel *G) {
    if (!G) return SPRS_OK;
    if (G->ctx) (void)hipStreamSynchronize(G->ctx->stream);   // never touches A: handles may be destroyed in any order
    for (hipGraphExec_t g : G->sweep_graph) if (g) (void)hipGraphExecDestroy(g);
    for (void *p : {(void *)G->lvl_rows, G->diag, G->resid, G->x_a, G->x_b, G->rhs_buf, (void *)G->d_bad})
        if (p) (void)hipFree(p);
    delete G;
    return SPRS_OK;
}

int64_t sprs_gauss_seidel_levels(const sprs_gauss_seidel *G) { return G ? (int64_t)G->lvl_ptr.size() - 1 : -1; }

int sprs_gauss_seidel_solve_d(sprs_gauss_seidel *G, const double *rhs, size_t rl, double *x, size_t xl, size_t max_iter, double eps,
                              size_t *its, double *res) {
    try { return gs_solve_host<double>(G, rhs, rl, x, xl, max_iter, eps, its, res); } catch (...) { return SPRS_ERR_HIP; }
}
int sprs_gauss_seidel_solve_s(sprs_gauss_seidel *G, const float *rhs, size_t rl, float *x, size_t xl, size_t max_iter, float eps,
                              size_t *its, float *res) {
    try { return gs_solve_host<float>(G, rhs, rl, x, xl, max_iter, eps, its, res); } catch (...) { return SPRS_ERR_HIP; }
}
int sprs_gauss_seidel_solve_dev_d(sprs_gauss_seidel *G, const double *rhs, size_t rl, double *x, size_t xl, size_t max_iter,
                                  double eps, size_t *its, double *res) {
    if (!G || !rhs || !x || !its || !res || G->dtype != DT_D) return SPRS_INVALID_ARGUMENT;
    if (rl != (size_t)G->n) return SPRS_INCOMPATIBLE_RHS_SIZE;
    if (rl != xl) return SPRS_INCOMPATIBLE_X_SIZE;
    try { return gs_solve_dev<double>(G, rhs, x, max_iter, eps, its, res); } catch (...) { return SPRS_ERR_HIP; }
}
int sprs_gauss_seidel_solve_dev_s(sprs_gauss_seidel *G, const float *rhs, size_t rl, float *x, size_t xl, size_t max_iter,
                                  float eps, size_t *its, float *res) {
    if (!G || !rhs || !x || !its || !res || G->dtype != DT_S) return SPRS_INVALID_ARGUMENT;
    if (rl != (size_t)G->n) return SPRS_INCOMPATIBLE_RHS_SIZE;
    if (rl != xl) return SPRS_INCOMPATIBLE_X_SIZE;
    try { return gs_solve_dev<float>(G, rhs, x, max_iter, eps, its, res); } catch (...) { return SPRS_ERR_HIP; }
}

}  // extern "C"
