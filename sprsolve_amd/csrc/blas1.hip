// BLAS-1 kernels of the hot path (reference src/vecalg.rs:556-605, src/precond.rs:20-52) as
// stand-alone launches.  HBM-bound streaming kernels: 16 bytes per lane per access, grid-stride,
// two-stage deterministic reductions (wavefront butterfly -> LDS -> one partial per workgroup ->
// fixed-order final pass).  The fused solver kernels in krylov.hip reuse the same arithmetic.
#include "device.hpp"

namespace sprs {

// ------------------------------------------------------------------ element-wise functors
template <class T, class S>
struct AxpyF {  // vecalg.rs:570-575  y += x * a
    S a; const T *x; T *y;
    template <int PK> __device__ __forceinline__ void run(int64_t i) const {
        auto xv = ldp<T, PK>(x, i); auto yv = ldp<T, PK>(y, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) yv.v[e] = sadd(yv.v[e], smulv(xv.v[e], a));
        stp<T, PK>(y, i, yv);
    }
};
template <class T>
struct AxpbyF {  // vecalg.rs:585-590  y = x*a + y*b
    T a, b; const T *x; T *y;
    template <int PK> __device__ __forceinline__ void run(int64_t i) const {
        auto xv = ldp<T, PK>(x, i); auto yv = ldp<T, PK>(y, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) yv.v[e] = sadd(smul(xv.v[e], a), smul(yv.v[e], b));
        stp<T, PK>(y, i, yv);
    }
};
template <class T>
struct ScaleF {  // vecalg.rs:592-595  v *= a
    T a; T *x;
    template <int PK> __device__ __forceinline__ void run(int64_t i) const {
        auto xv = ldp<T, PK>(x, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) xv.v[e] = smul(xv.v[e], a);
        stp<T, PK>(x, i, xv);
    }
};
template <class T>
struct RscaleF {  // vecalg.rs:596-599  v = v.mul_real(a)
    Real<T> a; T *x;
    template <int PK> __device__ __forceinline__ void run(int64_t i) const {
        auto xv = ldp<T, PK>(x, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) xv.v[e] = smulr(xv.v[e], a);
        stp<T, PK>(x, i, xv);
    }
};
template <class T>
struct ConjF {  // vecalg.rs:577-583  out = conj(in)
    const T *in; T *out;
    template <int PK> __device__ __forceinline__ void run(int64_t i) const {
        auto xv = ldp<T, PK>(in, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) xv.v[e] = sconj(xv.v[e]);
        stp<T, PK>(out, i, xv);
    }
};
template <class T, class V>
struct DiagApplyF {  // precond.rs:48-52  out = in * diag_inv   (V may be real while T is complex)
    const V *d; const T *in; T *out;
    template <int PK> __device__ __forceinline__ void run(int64_t i) const {
        auto xv = ldp<T, PK>(in, i); auto dv = ldp<V, PK>(d, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) xv.v[e] = smulv(xv.v[e], dv.v[e]);
        stp<T, PK>(out, i, xv);
    }
};
template <class V>
struct DiagInvF {  // precond.rs:22-24  diag_inv = one / v
    const V *d; V *o;
    template <int PK> __device__ __forceinline__ void run(int64_t i) const {
#pragma unroll
        for (int e = 0; e < PK; ++e) o[i * PK + e] = sinv(d[i * PK + e]);
    }
};

template <int PK, class F>
__global__ __launch_bounds__(BLOCK) void ew_kernel(int64_t n, F f) {
    SPRS_FOREACH_PACK(n, PK, i) f.template run<PK>(i);
    if (PK > 1) {
        int64_t i = (n / PK) * PK + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
        if (i < n) f.template run<1>(i);
    }
}

template <class T, class F>
static int launch_ew(sprs_ctx *c, size_t n, bool all_aligned, F f) {
    if (n == 0) return SPRS_OK;
    constexpr int PKW = pack_width<T>::value;
    const int pk = (all_aligned && PKW > 1) ? PKW : 1;
    int64_t work = ((int64_t)n / pk + BLOCK - 1) / BLOCK;
    const int g = balanced_grid(c, work);
    if (pk == PKW && PKW > 1)
        hipLaunchKernelGGL((ew_kernel<PKW, F>), dim3(g), dim3(BLOCK), 0, c->stream, (int64_t)n, f);
    else
        hipLaunchKernelGGL((ew_kernel<1, F>), dim3(g), dim3(BLOCK), 0, c->stream, (int64_t)n, f);
    SPRS_HIP_TRY(c, hipGetLastError());
    return SPRS_OK;
}

template <class T, class S>
int launch_axpy(sprs_ctx *c, size_t n, S a, const T *x, T *y) {
    return launch_ew<T>(c, n, aligned16(x) && aligned16(y), AxpyF<T, S>{a, x, y});
}
template <class T>
int launch_axpby(sprs_ctx *c, size_t n, T a, const T *x, T b, T *y) {
    return launch_ew<T>(c, n, aligned16(x) && aligned16(y), AxpbyF<T>{a, b, x, y});
}
template <class T>
int launch_scale(sprs_ctx *c, size_t n, T a, T *x) {
    return launch_ew<T>(c, n, aligned16(x), ScaleF<T>{a, x});
}
template <class T>
int launch_rscale(sprs_ctx *c, size_t n, Real<T> a, T *x) {
    return launch_ew<T>(c, n, aligned16(x), RscaleF<T>{a, x});
}
template <class T>
int launch_conj(sprs_ctx *c, size_t n, const T *in, T *out) {
    return launch_ew<T>(c, n, aligned16(in) && aligned16(out), ConjF<T>{in, out});
}
template <class T, class V>
int launch_diag_apply(sprs_ctx *c, size_t n, const V *dinv, const T *in, T *out) {
    return launch_ew<T>(c, n, aligned16(dinv) && aligned16(in) && aligned16(out), DiagApplyF<T, V>{dinv, in, out});
}
template <class V>
int launch_diag_inv(sprs_ctx *c, size_t n, const V *diag, V *dinv) {
    return launch_ew<V>(c, n, true, DiagInvF<V>{diag, dinv});
}

// ------------------------------------------------------------------ reductions
// A read-only kernel with ONE 16-byte load in flight per lane is latency-bound at 2 workgroups per CU (nrm2 of cfg 5
// ran at 3.3 TB/s where a read stream reaches 6.3, profiles/r02_tuning.md §1): the grid-stride loop is walked four packs
// at a time — the four loads are issued together, the additions stay in the ORIGINAL order (one accumulator,
// i, i+st, i+2st, i+3st, ...), so every partial is bit-identical to the plain loop's.
// NT: operands of 72 MB and more (stream_loads_nt) are read with non-temporal loads — a read-once stream that should not
// allocate on its way: nrm2 of a cfg-5 vector 68 -> 63 us (0.73 -> 0.79 of the HBM peak), dot 137 -> 123 us (0.73 -> 0.81),
// scripts/micro/red_shape.hip / profiles/r03_tuning.md; two workgroups per CU and four packs in flight stay the best shape.
constexpr int RED_UNROLL = 4;

template <class T, bool CONJ, int PK, bool NT>
__global__ __launch_bounds__(BLOCK) void dot_kernel(int64_t n, const T *__restrict__ x, const T *__restrict__ y,
                                                    T *__restrict__ part) {
    __shared__ T smem[NWAVE];
    T acc = szero<T>();
    const int64_t np = n / PK, st = (int64_t)gridDim.x * BLOCK;
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    for (; i + (RED_UNROLL - 1) * st < np; i += RED_UNROLL * st) {
        Pack<T, PK> xv[RED_UNROLL], yv[RED_UNROLL];
#pragma unroll
        for (int u = 0; u < RED_UNROLL; ++u) { xv[u] = ldp<T, PK, NT>(x, i + u * st); yv[u] = ldp<T, PK, NT>(y, i + u * st); }
#pragma unroll
        for (int u = 0; u < RED_UNROLL; ++u)
#pragma unroll
            for (int e = 0; e < PK; ++e) acc = sadd(acc, smul(CONJ ? sconj(xv[u].v[e]) : xv[u].v[e], yv[u].v[e]));
    }
    for (; i < np; i += st) {
        auto xv = ldp<T, PK, NT>(x, i); auto yv = ldp<T, PK, NT>(y, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) acc = sadd(acc, smul(CONJ ? sconj(xv.v[e]) : xv.v[e], yv.v[e]));
    }
    if (PK > 1) {
        int64_t t = (n / PK) * PK + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
        if (t < n) acc = sadd(acc, smul(CONJ ? sconj(x[t]) : x[t], y[t]));
    }
    acc = block_sum(acc, smem);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

template <class T, int PK, bool NT>
__global__ __launch_bounds__(BLOCK) void nrm2sq_kernel(int64_t n, const T *__restrict__ x, Real<T> *__restrict__ part) {
    __shared__ Real<T> smem[NWAVE];
    Real<T> acc = 0;
    const int64_t np = n / PK, st = (int64_t)gridDim.x * BLOCK;
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    for (; i + (RED_UNROLL - 1) * st < np; i += RED_UNROLL * st) {
        Pack<T, PK> xv[RED_UNROLL];
#pragma unroll
        for (int u = 0; u < RED_UNROLL; ++u) xv[u] = ldp<T, PK, NT>(x, i + u * st);
#pragma unroll
        for (int u = 0; u < RED_UNROLL; ++u)
#pragma unroll
            for (int e = 0; e < PK; ++e) acc = acc + ssq(xv[u].v[e]);
    }
    for (; i < np; i += st) {
        auto xv = ldp<T, PK, NT>(x, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) acc = acc + ssq(xv.v[e]);
    }
    if (PK > 1) {
        int64_t t = (n / PK) * PK + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
        if (t < n) acc = acc + ssq(x[t]);
    }
    acc = block_sum(acc, smem);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

template <class T>
__global__ __launch_bounds__(BLOCK) void finalize_kernel(const T *__restrict__ part, int P, T *__restrict__ out) {
    __shared__ T smem[NWAVE];
    T v = reduce_partials(part, P, smem);
    if (threadIdx.x == 0) out[0] = v;
}

template <class T>
int reduce_partials_host(sprs_ctx *c, const T *part, int P, T *out, sprs_comm *comm) {
    CtxLock lock(c);   // d_scal / h_scal are per-context scratch
    T *d_out = reinterpret_cast<T *>(c->d_scal);
    hipLaunchKernelGGL((finalize_kernel<T>), dim3(1), dim3(BLOCK), 0, c->stream, part, P, d_out);
    SPRS_HIP_TRY(c, hipGetLastError());
    if (comm) SPRS_TRY(allreduce_sum(comm, d_out, sizeof(T) / sizeof(Real<T>), sizeof(Real<T>) == 4));
    SPRS_HIP_TRY(c, hipMemcpyAsync(c->h_scal, d_out, sizeof(T), hipMemcpyDeviceToHost, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(out, c->h_scal, sizeof(T));
    return SPRS_OK;
}

static int red_grid(sprs_ctx *c, size_t n, int pk) {
    int64_t work = ((int64_t)n / pk + BLOCK - 1) / BLOCK;
    return balanced_grid(c, work);
}

template <class T>
int dot_host(sprs_ctx *c, size_t n, const T *x, const T *y, bool conj, T *out, sprs_comm *comm) {
    CtxLock lock(c);   // d_part is per-context scratch
    constexpr int PKW = pack_width<T>::value;
    const bool al = aligned16(x) && aligned16(y);
    const int pk = (al && PKW > 1) ? PKW : 1;
    const int g = red_grid(c, n, pk);
    T *part = reinterpret_cast<T *>(c->d_part);
    const bool nt = pk > 1 && stream_loads_nt(c, n * sizeof(T));      // (16-byte packs only: the unaligned path is not a bulk path)
#define SPRS_LAUNCH_DOT(CJ, PKV, NTF) \
    hipLaunchKernelGGL((dot_kernel<T, CJ, PKV, NTF>), dim3(g), dim3(BLOCK), 0, c->stream, (int64_t)n, x, y, part)
    if (pk == 1) { if (conj) SPRS_LAUNCH_DOT(true, 1, false); else SPRS_LAUNCH_DOT(false, 1, false); }
    else if (nt) { if (conj) SPRS_LAUNCH_DOT(true, PKW, true); else SPRS_LAUNCH_DOT(false, PKW, true); }
    else         { if (conj) SPRS_LAUNCH_DOT(true, PKW, false); else SPRS_LAUNCH_DOT(false, PKW, false); }
#undef SPRS_LAUNCH_DOT
    SPRS_HIP_TRY(c, hipGetLastError());
    return reduce_partials_host<T>(c, part, g, out, comm);
}

template <class T>
int norm2_host(sprs_ctx *c, size_t n, const T *x, Real<T> *out, sprs_comm *comm) {
    CtxLock lock(c);
    constexpr int PKW = pack_width<T>::value;
    const int pk = (aligned16(x) && PKW > 1) ? PKW : 1;
    const int g = red_grid(c, n, pk);
    Real<T> *part = reinterpret_cast<Real<T> *>(c->d_part);
    if (pk == 1) hipLaunchKernelGGL((nrm2sq_kernel<T, 1, false>), dim3(g), dim3(BLOCK), 0, c->stream, (int64_t)n, x, part);
    else if (stream_loads_nt(c, n * sizeof(T))) hipLaunchKernelGGL((nrm2sq_kernel<T, PKW, true>), dim3(g), dim3(BLOCK), 0, c->stream, (int64_t)n, x, part);
    else hipLaunchKernelGGL((nrm2sq_kernel<T, PKW, false>), dim3(g), dim3(BLOCK), 0, c->stream, (int64_t)n, x, part);
    SPRS_HIP_TRY(c, hipGetLastError());
    Real<T> s = 0;
    SPRS_TRY(reduce_partials_host<Real<T>>(c, part, g, &s, comm));
    *out = ssqrt(s);  // vecalg.rs:604
    return SPRS_OK;
}

// ------------------------------------------------------------------ explicit instantiations
#define SPRS_INST_REAL(T)                                                                    \
    template int launch_axpy<T, T>(sprs_ctx *, size_t, T, const T *, T *);                   \
    template int launch_axpby<T>(sprs_ctx *, size_t, T, const T *, T, T *);                  \
    template int launch_scale<T>(sprs_ctx *, size_t, T, T *);                                \
    template int launch_rscale<T>(sprs_ctx *, size_t, Real<T>, T *);                         \
    template int launch_conj<T>(sprs_ctx *, size_t, const T *, T *);                         \
    template int launch_diag_apply<T, Real<T>>(sprs_ctx *, size_t, const Real<T> *, const T *, T *); \
    template int launch_diag_inv<T>(sprs_ctx *, size_t, const T *, T *);                     \
    template int dot_host<T>(sprs_ctx *, size_t, const T *, const T *, bool, T *, sprs_comm *); \
    template int norm2_host<T>(sprs_ctx *, size_t, const T *, Real<T> *, sprs_comm *);       \
    template int reduce_partials_host<T>(sprs_ctx *, const T *, int, T *, sprs_comm *);
#define SPRS_INST_CPLX(T)                                                                    \
    SPRS_INST_REAL(T)                                                                        \
    template int launch_axpy<T, Real<T>>(sprs_ctx *, size_t, Real<T>, const T *, T *);       \
    template int launch_diag_apply<T, T>(sprs_ctx *, size_t, const T *, const T *, T *);
SPRS_INST_REAL(double)
SPRS_INST_REAL(float)
SPRS_INST_CPLX(cplx)
SPRS_INST_CPLX(cplxf)

}  // namespace sprs
