// Round-4 micro-benchmark (not part of the product): what does computing the vector update that PRODUCES an SpMV's input
// inside that SpMV's window staging buy?  
//
//   K3 + K4 of BiCGStab (bicg_stab.rs:172-175):   s = r + v * (-alpha) ;  t = A s ;  partials of t.t and t.s
//     unfused  : upd2 (2 reads + 1 write, in place, non-temporal)  then  win<DOT 2> on s
//     fused    : winf2 — the tile stages s = r + v * na for its window (two window loads per lane and piece), combines the far
//                pairs from r and v, stores s for its OWN rows into a second buffer (in place would race with other tiles'
//                windows), folds from LDS
//   K1 + K2 (bicg_stab.rs:155-160):               p = (v * a + p * beta) + r ;  v' = A p ;  partials of r0.v'
//     unfused  : upd3 (3 reads + 1 write)  then  win<DOT 1, u = r0>
//     fused    : winf3 — three window loads per piece, own rows of p to a second buffer
//
// The matrix is the 7-diagonal constant-coefficient operator of scripts/micro/stencil_window.hip, rows [P, n - P), tiles of 4096
// rows dealt to the XCDs by their phase within the plane period (the product's tile plan).  Every variant is compared bit for
// bit with the unfused pair (same element-wise rounding sequence, same fold).
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/micro/fused_window.hip -o /tmp/fused_window && /tmp/fused_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#include <algorithm>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256;
struct alignas(16) D2 { double lo, hi; };
typedef unsigned u4w __attribute__((ext_vector_type(4)));

__device__ inline double hval(long k) { unsigned long h = (unsigned long)k * 0x9E3779B97F4A7C15ull; h ^= h >> 29; return (double)(h & 0xfffff) / 1048576.0 - 0.5; }
__global__ void fill_vec(long n, double *x, unsigned seed) {
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n; g += (long)gridDim.x * blockDim.x) x[g] = hval(g * 3 + seed);
}
struct Coef { double v[7]; };
__device__ __forceinline__ double fold7(const Coef &c, double a0, double a1, double a2, double a3, double a4, double a5, double a6) {
    double acc = 0.0;
    acc = acc + a0 * c.v[0]; acc = acc + a1 * c.v[1]; acc = acc + a2 * c.v[2]; acc = acc + a3 * c.v[3];
    acc = acc + a4 * c.v[4]; acc = acc + a5 * c.v[5]; acc = acc + a6 * c.v[6];
    return acc;
}
__global__ void cmp_kernel(long r0, long r1, const double *a, const double *b, unsigned long long *bad) {
    unsigned long long c = 0;
    for (long g = r0 + blockIdx.x * (long)blockDim.x + threadIdx.x; g < r1; g += (long)gridDim.x * blockDim.x)
        c += (__double_as_longlong(a[g]) != __double_as_longlong(b[g]));
    if (c) atomicAdd(bad, c);
}
__device__ __forceinline__ D2 ldg2(const double *p) { return *reinterpret_cast<const D2 *>(p); }
__device__ __forceinline__ D2 ldnt2(const double *p) { u4w w = __builtin_nontemporal_load(reinterpret_cast<const u4w *>(p)); D2 r; __builtin_memcpy(&r, &w, 16); return r; }
__device__ __forceinline__ void stnt2(double *p, D2 v) { u4w w; __builtin_memcpy(&w, &v, 16); __builtin_nontemporal_store(w, reinterpret_cast<u4w *>(p)); }
__device__ __forceinline__ D2 as_d2(u4w w) { D2 r; __builtin_memcpy(&r, &w, 16); return r; }
__device__ __forceinline__ u4w as_u4(D2 d) { u4w w; __builtin_memcpy(&w, &d, 16); return w; }

// the product's fused_kernel<BicgK3>: r = r + v * na, 16 bytes per lane, non-temporal, grid-stride (out == r: in place)
__global__ __launch_bounds__(BLOCK) void upd2(long n, double na, const double *__restrict__ v, const double *r, double *out) {
    const long n2 = n >> 1;
    for (long g = blockIdx.x * (long)BLOCK + threadIdx.x; g < n2; g += (long)gridDim.x * BLOCK) {
        const D2 vv = ldnt2(v + 2 * g); D2 rv = ldnt2(r + 2 * g);
        rv.lo = rv.lo + vv.lo * na; rv.hi = rv.hi + vv.hi * na;
        stnt2(out + 2 * g, rv);
    }
}
// fused_kernel<BicgK1>: p = (v * a + p * beta) + r * 1
__global__ __launch_bounds__(BLOCK) void upd3(long n, double a, double beta, const double *__restrict__ v, const double *__restrict__ r, const double *p, double *out) {
    const long n2 = n >> 1;
    for (long g = blockIdx.x * (long)BLOCK + threadIdx.x; g < n2; g += (long)gridDim.x * BLOCK) {
        const D2 vv = ldnt2(v + 2 * g), pv = ldnt2(p + 2 * g), rv = ldnt2(r + 2 * g);
        D2 o;
        o.lo = (vv.lo * a + pv.lo * beta) + rv.lo * 1.0; o.hi = (vv.hi * a + pv.hi * beta) + rv.hi * 1.0;
        stnt2(out + 2 * g, o);
    }
}

// ---- the product's tile kernel shape (spmv_tile_kernel<DOT, UX, 7, 1, 1, 512> without seams / descriptors / left blocks)
template <int T, int W, int DOT>
__global__ __launch_bounds__(BLOCK) void k_win(long r_begin, int nx, long P, Coef c, const double *__restrict__ x, double *__restrict__ y,
                                               const double *__restrict__ u, double *__restrict__ part, const int *__restrict__ order, const int *__restrict__ xstart) {
    constexpr int NW = (T + 2 * W) / 2 / BLOCK, NQ = T / 512;
    __shared__ __attribute__((aligned(16))) double win[T + 2 * W];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double d0 = 0.0, d1 = 0.0;
    long s = xstart[blockIdx.x & 7] + (blockIdx.x >> 3);
    const long send = xstart[(blockIdx.x & 7) + 1], sstep = gridDim.x >> 3;
    for (; s < send; s += sstep) {
        const long ts = r_begin + (long)order[s] * T;
        u4w wreg[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) wreg[i] = *reinterpret_cast<const u4w *>(x + ts - W + 2 * (long)(tid + i * BLOCK));
        D2 fm[NQ], fp[NQ], uu[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const long r0 = ts + ((q * 4 + wv) << 7) + 2 * lane;
            fm[q] = ldg2(x + r0 - P); fp[q] = ldg2(x + r0 + P);
            if (DOT == 1) uu[q] = ldnt2(u + r0);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NW; ++i) *reinterpret_cast<u4w *>(&win[2 * (tid + i * BLOCK)]) = wreg[i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int li = W + ((q * 4 + wv) << 7) + 2 * lane;
            const D2 cc = *reinterpret_cast<const D2 *>(&win[li]);
            const D2 a = *reinterpret_cast<const D2 *>(&win[li - nx]);
            const D2 b = *reinterpret_cast<const D2 *>(&win[li + nx]);
            const double xl = win[li - 1], xr = win[li + 2];
            D2 o;
            o.lo = fold7(c, fm[q].lo, a.lo, xl, cc.lo, cc.hi, b.lo, fp[q].lo);
            o.hi = fold7(c, fm[q].hi, a.hi, cc.lo, cc.hi, xr, b.hi, fp[q].hi);
            stnt2(y + ts + ((q * 4 + wv) << 7) + 2 * lane, o);
            if (DOT == 1) { d0 = d0 + uu[q].lo * o.lo; d0 = d0 + uu[q].hi * o.hi; }
            if (DOT == 2) { d0 = d0 + o.lo * o.lo; d1 = d1 + o.lo * cc.lo; d0 = d0 + o.hi * o.hi; d1 = d1 + o.hi * cc.hi; }
        }
    }
    if (DOT) {
        for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
        if (lane == 0) { part[blockIdx.x * 4 + wv] = d0; part[4096 + blockIdx.x * 4 + wv] = d1; }
    }
}

// ---- fused: NV input vectors combined on the fly.  NV = 2: s = r + v * c0 (K3 -> K4, DOT 2 on s);  NV = 3: p = (v * c0 + p * c1) + r (K1 -> K2, DOT 1 on u)
// in0 = r, in1 = v, in2 = p (NV = 3).  own: the combined vector for the tile's own rows.  FB: far pairs of FB row blocks per batch.
template <int NV>
__device__ __forceinline__ double comb(double r, double v, double p, double c0, double c1) {
    if (NV == 2) return r + v * c0;
    return (v * c0 + p * c1) + r * 1.0;
}
template <int T, int W, int NV, int FB, int WB>
__global__ __launch_bounds__(BLOCK) void k_winf(long r_begin, int nx, long P, Coef c, double c0, double c1, const double *__restrict__ in0,
                                                const double *__restrict__ in1, const double *__restrict__ in2, double *__restrict__ own,
                                                double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part,
                                                const int *__restrict__ order, const int *__restrict__ xstart) {
    constexpr int NW = (T + 2 * W) / 2 / BLOCK, NQ = T / 512;
    static_assert(NQ % FB == 0 && NW % WB == 0, "batches");
    constexpr int DOT = NV == 2 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) double win[T + 2 * W];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double d0 = 0.0, d1 = 0.0;
    long s = xstart[blockIdx.x & 7] + (blockIdx.x >> 3);
    const long send = xstart[(blockIdx.x & 7) + 1], sstep = gridDim.x >> 3;
    for (; s < send; s += sstep) {
        const long ts = r_begin + (long)order[s] * T;
        __syncthreads();                                                // the previous tile's window has been read
        // ---- the window, WB pieces per lane at a time: load NV vectors, combine, store to LDS
#pragma unroll
        for (int i0 = 0; i0 < NW; i0 += WB) {
            u4w w0[WB], w1[WB], w2[NV == 3 ? WB : 1];
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const long g = ts - W + 2 * (long)(tid + (i0 + i) * BLOCK);
                w0[i] = *reinterpret_cast<const u4w *>(in0 + g); w1[i] = *reinterpret_cast<const u4w *>(in1 + g);
                if (NV == 3) w2[i] = *reinterpret_cast<const u4w *>(in2 + g);
            }
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const D2 a = as_d2(w0[i]), b = as_d2(w1[i]), p = NV == 3 ? as_d2(w2[i]) : D2{0.0, 0.0};
                const D2 o{comb<NV>(a.lo, b.lo, p.lo, c0, c1), comb<NV>(a.hi, b.hi, p.hi, c0, c1)};
                *reinterpret_cast<D2 *>(&win[2 * (tid + (i0 + i) * BLOCK)]) = o;
            }
        }
        __syncthreads();
        // ---- the row blocks, FB at a time: far pairs of NV vectors (and the dot operand), combine, fold
#pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += FB) {
            D2 m0[FB], m1[FB], m2[NV == 3 ? FB : 1], p0[FB], p1[FB], p2[NV == 3 ? FB : 1], uu[DOT == 1 ? FB : 1];
#pragma unroll
            for (int k = 0; k < FB; ++k) {
                const long r0 = ts + (((q0 + k) * 4 + wv) << 7) + 2 * lane;
                m0[k] = ldg2(in0 + r0 - P); m1[k] = ldg2(in1 + r0 - P); p0[k] = ldg2(in0 + r0 + P); p1[k] = ldg2(in1 + r0 + P);
                if (NV == 3) { m2[k] = ldg2(in2 + r0 - P); p2[k] = ldg2(in2 + r0 + P); }
                if (DOT == 1) uu[k] = ldnt2(u + r0);
            }
#pragma unroll
            for (int k = 0; k < FB; ++k) {
                const int q = q0 + k;
                const int li = W + ((q * 4 + wv) << 7) + 2 * lane;
                const D2 cc = *reinterpret_cast<const D2 *>(&win[li]);
                const D2 a = *reinterpret_cast<const D2 *>(&win[li - nx]);
                const D2 b = *reinterpret_cast<const D2 *>(&win[li + nx]);
                const double xl = win[li - 1], xr = win[li + 2];
                const D2 z{0.0, 0.0};
                const D2 fm{comb<NV>(m0[k].lo, m1[k].lo, (NV == 3 ? m2[k] : z).lo, c0, c1), comb<NV>(m0[k].hi, m1[k].hi, (NV == 3 ? m2[k] : z).hi, c0, c1)};
                const D2 fp{comb<NV>(p0[k].lo, p1[k].lo, (NV == 3 ? p2[k] : z).lo, c0, c1), comb<NV>(p0[k].hi, p1[k].hi, (NV == 3 ? p2[k] : z).hi, c0, c1)};
                D2 o;
                o.lo = fold7(c, fm.lo, a.lo, xl, cc.lo, cc.hi, b.lo, fp.lo);
                o.hi = fold7(c, fm.hi, a.hi, cc.lo, cc.hi, xr, b.hi, fp.hi);
                const long r0 = ts + ((q * 4 + wv) << 7) + 2 * lane;
                stnt2(y + r0, o);
                stnt2(own + r0, cc);
                if (DOT == 1) { d0 = d0 + uu[k].lo * o.lo; d0 = d0 + uu[k].hi * o.hi; }
                if (DOT == 2) { d0 = d0 + o.lo * o.lo; d1 = d1 + o.lo * cc.lo; d0 = d0 + o.hi * o.hi; d1 = d1 + o.hi * cc.hi; }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
    if (lane == 0) { part[blockIdx.x * 4 + wv] = d0; part[4096 + blockIdx.x * 4 + wv] = d1; }
}

// ---- the same with EVERY load of a tile issued before anything is consumed (the product kernel's shape: one exposed round trip
// per tile): raw window pieces and raw far pairs of all NV vectors in registers together — only fits for smaller tiles
template <int T, int W, int NV>
__global__ __launch_bounds__(BLOCK) void k_winf_up(long r_begin, int nx, long P, Coef c, double c0, double c1, const double *__restrict__ in0,
                                                   const double *__restrict__ in1, const double *__restrict__ in2, double *__restrict__ own,
                                                   double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part,
                                                   const int *__restrict__ order, const int *__restrict__ xstart) {
    constexpr int NW = (T + 2 * W) / 2 / BLOCK, NQ = T / 512;
    constexpr int DOT = NV == 2 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) double win[T + 2 * W];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double d0 = 0.0, d1 = 0.0;
    long s = xstart[blockIdx.x & 7] + (blockIdx.x >> 3);
    const long send = xstart[(blockIdx.x & 7) + 1], sstep = gridDim.x >> 3;
    for (; s < send; s += sstep) {
        const long ts = r_begin + (long)order[s] * T;
        u4w w0[NW], w1[NW], w2[NV == 3 ? NW : 1];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const long g = ts - W + 2 * (long)(tid + i * BLOCK);
            w0[i] = *reinterpret_cast<const u4w *>(in0 + g); w1[i] = *reinterpret_cast<const u4w *>(in1 + g);
            if (NV == 3) w2[i] = *reinterpret_cast<const u4w *>(in2 + g);
        }
        D2 m0[NQ], m1[NQ], m2[NV == 3 ? NQ : 1], p0[NQ], p1[NQ], p2[NV == 3 ? NQ : 1], uu[DOT == 1 ? NQ : 1];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const long r0 = ts + ((k * 4 + wv) << 7) + 2 * lane;
            m0[k] = ldg2(in0 + r0 - P); m1[k] = ldg2(in1 + r0 - P); p0[k] = ldg2(in0 + r0 + P); p1[k] = ldg2(in1 + r0 + P);
            if (NV == 3) { m2[k] = ldg2(in2 + r0 - P); p2[k] = ldg2(in2 + r0 + P); }
            if (DOT == 1) uu[k] = ldnt2(u + r0);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const D2 a = as_d2(w0[i]), b = as_d2(w1[i]), p = NV == 3 ? as_d2(w2[i]) : D2{0.0, 0.0};
            *reinterpret_cast<D2 *>(&win[2 * (tid + i * BLOCK)]) = D2{comb<NV>(a.lo, b.lo, p.lo, c0, c1), comb<NV>(a.hi, b.hi, p.hi, c0, c1)};
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int li = W + ((q * 4 + wv) << 7) + 2 * lane;
            const D2 cc = *reinterpret_cast<const D2 *>(&win[li]);
            const D2 a = *reinterpret_cast<const D2 *>(&win[li - nx]);
            const D2 b = *reinterpret_cast<const D2 *>(&win[li + nx]);
            const double xl = win[li - 1], xr = win[li + 2];
            const D2 z{0.0, 0.0};
            const D2 fm{comb<NV>(m0[q].lo, m1[q].lo, (NV == 3 ? m2[q] : z).lo, c0, c1), comb<NV>(m0[q].hi, m1[q].hi, (NV == 3 ? m2[q] : z).hi, c0, c1)};
            const D2 fp{comb<NV>(p0[q].lo, p1[q].lo, (NV == 3 ? p2[q] : z).lo, c0, c1), comb<NV>(p0[q].hi, p1[q].hi, (NV == 3 ? p2[q] : z).hi, c0, c1)};
            D2 o;
            o.lo = fold7(c, fm.lo, a.lo, xl, cc.lo, cc.hi, b.lo, fp.lo);
            o.hi = fold7(c, fm.hi, a.hi, cc.lo, cc.hi, xr, b.hi, fp.hi);
            const long r0 = ts + ((q * 4 + wv) << 7) + 2 * lane;
            stnt2(y + r0, o);
            stnt2(own + r0, cc);
            if (DOT == 1) { d0 = d0 + uu[q].lo * o.lo; d0 = d0 + uu[q].hi * o.hi; }
            if (DOT == 2) { d0 = d0 + o.lo * o.lo; d1 = d1 + o.lo * cc.lo; d0 = d0 + o.hi * o.hi; d1 = d1 + o.hi * cc.hi; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
    if (lane == 0) { part[blockIdx.x * 4 + wv] = d0; part[4096 + blockIdx.x * 4 + wv] = d1; }
}

// ---- plane streaming ("chains"): a workgroup owns a COLUMN of tiles — rows [c T, (c + 1) T) of consecutive planes — and keeps the
// windows of planes z - 1, z, z + 1 in LDS: the -P / +P operands of plane z are the centres of its neighbours' windows, so no far
// load exists at all and every element of the input crosses the L1 once per window ((T + 2W) / T loads per lane and 128 rows
// instead of (T + 2W) / T + 2 per vector), independent of what survives in an L2.  The loads of window z + 2 fly over the fold of
// plane z.  NV input vectors are combined while the window is staged (NV = 1: a plain SpMV).
template <int T, int W, int NV, int DOT>
__global__ __launch_bounds__(BLOCK) void k_chain(long r_begin, long n, int nx, long P, int ncol, int nzi, int zc, Coef c, double c0, double c1,
                                                 const double *__restrict__ in0, const double *__restrict__ in1, const double *__restrict__ in2,
                                                 double *__restrict__ own, double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part) {
    constexpr int WL = T + 2 * W, NW = WL / 2 / BLOCK, NQ = T / 512;
    static_assert(WL % (2 * BLOCK) == 0, "shape");
    __shared__ __attribute__((aligned(16))) double win[3][WL];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double d0 = 0.0, d1 = 0.0;
    // work item = (column, chunk of zc planes); XCD x owns a contiguous eighth of the columns (neighbouring windows overlap by 2W)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int c_lo = (int)((long)ncol * xcd / 8), c_hi = (int)((long)ncol * (xcd + 1) / 8), ncx = c_hi - c_lo;
    const int nchunk = (nzi + zc - 1) / zc;
    for (int item = j; item < ncx * nchunk; item += gridDim.x >> 3) {
        const int col = c_lo + item % ncx, ch = item / ncx;
        const int z0 = ch * zc, z1 = min(nzi, z0 + zc);                  // interior planes [z0, z1) of this chunk (plane 0 = r_begin)
        const long base = r_begin + (long)col * T - W;                  // first element of the window of interior plane 0
        u4w w0[NW], w1[NV >= 2 ? NW : 1], w2[NV == 3 ? NW : 1];
        auto issue = [&](int z) {                                       // raw window of plane z (z = -1 and nzi are the boundary planes: they exist in x)
            const long g0 = base + (long)z * P;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                long g = g0 + 2 * (long)(tid + i * BLOCK);
                g = g < 0 ? 0 : (g > n - 2 ? n - 2 : g);               // the margins of the first / last plane's windows (never folded) may leave the vectors
                w0[i] = *reinterpret_cast<const u4w *>(in0 + g);
                if (NV >= 2) w1[i] = *reinterpret_cast<const u4w *>(in1 + g);
                if (NV == 3) w2[i] = *reinterpret_cast<const u4w *>(in2 + g);
            }
        };
        auto stage = [&](int slot) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                D2 o = as_d2(w0[i]);
                if (NV >= 2) {
                    const D2 a = as_d2(w0[i]), b = as_d2(w1[i]), p = NV == 3 ? as_d2(w2[i]) : D2{0.0, 0.0};
                    o = D2{comb<NV>(a.lo, b.lo, p.lo, c0, c1), comb<NV>(a.hi, b.hi, p.hi, c0, c1)};
                }
                *reinterpret_cast<D2 *>(&win[slot][2 * (tid + i * BLOCK)]) = o;
            }
        };
        __syncthreads();                                                // the previous item's windows have been read
        issue(z0 - 1); stage((z0 + 2) % 3);                             // slot of plane z is (z + 3) % 3
        issue(z0); stage((z0 + 3) % 3);
        issue(z0 + 1);
        for (int z = z0; z < z1; ++z) {
            stage((z + 4) % 3);                                         // plane z + 1 (its loads flew over the previous fold)
            __syncthreads();
            if (z + 1 < z1) issue(z + 2);
            const double *wm = win[(z + 2) % 3], *wc = win[(z + 3) % 3], *wp = win[(z + 4) % 3];
            const long ts = r_begin + (long)z * P + (long)col * T;
            D2 uu[DOT == 1 ? NQ : 1];
            if (DOT == 1) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) uu[q] = ldnt2(u + ts + ((q * 4 + wv) << 7) + 2 * lane);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int li = W + ((q * 4 + wv) << 7) + 2 * lane;
                const D2 cc = *reinterpret_cast<const D2 *>(&wc[li]);
                const D2 a = *reinterpret_cast<const D2 *>(&wc[li - nx]);
                const D2 b = *reinterpret_cast<const D2 *>(&wc[li + nx]);
                const D2 fm = *reinterpret_cast<const D2 *>(&wm[li]);
                const D2 fp = *reinterpret_cast<const D2 *>(&wp[li]);
                const double xl = wc[li - 1], xr = wc[li + 2];
                D2 o;
                o.lo = fold7(c, fm.lo, a.lo, xl, cc.lo, cc.hi, b.lo, fp.lo);
                o.hi = fold7(c, fm.hi, a.hi, cc.lo, cc.hi, xr, b.hi, fp.hi);
                const long r0 = ts + ((q * 4 + wv) << 7) + 2 * lane;
                stnt2(y + r0, o);
                if (NV >= 2 && own != nullptr) stnt2(own + r0, cc);
                if (DOT == 1) { d0 = d0 + uu[q].lo * o.lo; d0 = d0 + uu[q].hi * o.hi; }
                if (DOT == 2) { d0 = d0 + o.lo * o.lo; d1 = d1 + o.lo * cc.lo; d0 = d0 + o.hi * o.hi; d1 = d1 + o.hi * cc.hi; }
            }
            __syncthreads();                                            // plane z - 1's slot is free for plane z + 2
        }
    }
    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
    if (lane == 0) { part[blockIdx.x * 4 + wv] = d0; part[4096 + blockIdx.x * 4 + wv] = d1; }
}

// ---- the consumer that follows K4 in the iteration: fused_kernel<BicgK5> — x += y na + s nw ; r = s + t nw ; partials of |r|^2, r0.r
// (5 reads + 2 writes, non-temporal, grid-stride).  REV: the same pass from the last element to the first.
template <bool REV, bool NTL>
__global__ __launch_bounds__(BLOCK) void k5_like(long n, double na, double nw, const double *__restrict__ yv, const double *__restrict__ tv,
                                                 const double *__restrict__ r0v, double *xv, double *sv, double *__restrict__ part) {
    const long n2 = n >> 1;
    double d0 = 0.0, d1 = 0.0;
    for (long g0 = blockIdx.x * (long)BLOCK + threadIdx.x; g0 < n2; g0 += (long)gridDim.x * BLOCK) {
        const long g = REV ? n2 - 1 - g0 : g0;
        D2 x2 = NTL ? ldnt2(xv + 2 * g) : ldg2(xv + 2 * g), y2 = NTL ? ldnt2(yv + 2 * g) : ldg2(yv + 2 * g), s2 = NTL ? ldnt2(sv + 2 * g) : ldg2(sv + 2 * g);
        const D2 t2 = NTL ? ldnt2(tv + 2 * g) : ldg2(tv + 2 * g), q2 = NTL ? ldnt2(r0v + 2 * g) : ldg2(r0v + 2 * g);
        x2.lo = (x2.lo + y2.lo * na) + s2.lo * nw; x2.hi = (x2.hi + y2.hi * na) + s2.hi * nw;
        s2.lo = s2.lo + t2.lo * nw; s2.hi = s2.hi + t2.hi * nw;
        d0 = d0 + s2.lo * s2.lo; d0 = d0 + s2.hi * s2.hi; d1 = d1 + q2.lo * s2.lo; d1 = d1 + q2.hi * s2.hi;
        stnt2(xv + 2 * g, x2); stnt2(sv + 2 * g, s2);
    }
    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
    if ((threadIdx.x & 63) == 0) { part[8192 + blockIdx.x * 4 + (threadIdx.x >> 6)] = d0; part[12288 + blockIdx.x * 4 + (threadIdx.x >> 6)] = d1; }
}

// the same consumer when the fused K4 does NOT store s: s = r + v na is formed again here (6 reads + 2 writes; r' in place of r)
__global__ __launch_bounds__(BLOCK) void k5_like6(long n, double na, double nal, double nw, const double *__restrict__ yv, const double *__restrict__ tv,
                                                  const double *__restrict__ r0v, const double *__restrict__ vv, double *xv, double *rv, double *__restrict__ part) {
    const long n2 = n >> 1;
    double d0 = 0.0, d1 = 0.0;
    for (long g = blockIdx.x * (long)BLOCK + threadIdx.x; g < n2; g += (long)gridDim.x * BLOCK) {
        D2 x2 = ldnt2(xv + 2 * g), y2 = ldnt2(yv + 2 * g), r2 = ldnt2(rv + 2 * g);
        const D2 v2 = ldnt2(vv + 2 * g), t2 = ldnt2(tv + 2 * g), q2 = ldnt2(r0v + 2 * g);
        D2 s2{r2.lo + v2.lo * nal, r2.hi + v2.hi * nal};
        x2.lo = (x2.lo + y2.lo * na) + s2.lo * nw; x2.hi = (x2.hi + y2.hi * na) + s2.hi * nw;
        s2.lo = s2.lo + t2.lo * nw; s2.hi = s2.hi + t2.hi * nw;
        d0 = d0 + s2.lo * s2.lo; d0 = d0 + s2.hi * s2.hi; d1 = d1 + q2.lo * s2.lo; d1 = d1 + q2.hi * s2.hi;
        stnt2(xv + 2 * g, x2); stnt2(rv + 2 * g, s2);
    }
    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
    if ((threadIdx.x & 63) == 0) { part[8192 + blockIdx.x * 4 + (threadIdx.x >> 6)] = d0; part[12288 + blockIdx.x * 4 + (threadIdx.x >> 6)] = d1; }
}

int main(int argc, char **argv) {
    const std::string filt = argc > 1 ? argv[1] : "";
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nx = 500, ny = 500, nz = argc > 3 ? atoi(argv[3]) : 200;
    const long P = (long)nx * ny, n = P * nz;
    constexpr int T = 4096, W = 512;
    const long r_begin = P, r_end = r_begin + (n - 2 * P) / T * T;
    double *r, *v, *p, *r0v, *s_ref, *s_out, *y, *y_ref, *part, *junk_a, *junk_b; unsigned long long *bad;
    for (double **q : {&r, &v, &p, &r0v, &s_ref, &s_out, &y, &y_ref, &junk_a, &junk_b}) CK(hipMalloc(q, n * 8));
    CK(hipMalloc(&part, 1 << 20)); CK(hipMalloc(&bad, 8));
    fill_vec<<<2048, 256>>>(n, r, 1); fill_vec<<<2048, 256>>>(n, v, 7); fill_vec<<<2048, 256>>>(n, p, 13); fill_vec<<<2048, 256>>>(n, r0v, 29);
    fill_vec<<<2048, 256>>>(n, junk_a, 3); fill_vec<<<2048, 256>>>(n, junk_b, 5);
    Coef c{{-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0}};
    const double na = -0.37, ca = -0.81, cb = 0.93;
    // tile order: XCD sections by phase within the plane period, as the product's tile plan
    const long ntile = (r_end - r_begin) / T;
    std::vector<int> ord, xs(9, 0);
    for (int xc = 0; xc < 8; ++xc) { xs[xc] = (int)ord.size(); for (long t = 0; t < ntile; ++t) if ((int)(((t * T) % P) * 8 / P) == xc) ord.push_back((int)t); }
    xs[8] = (int)ord.size();
    int *d_ord, *d_xs;
    CK(hipMalloc(&d_ord, ord.size() * 4)); CK(hipMemcpy(d_ord, ord.data(), ord.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_xs, 36)); CK(hipMemcpy(d_xs, xs.data(), 36, hipMemcpyHostToDevice));
    auto make_order = [&](int TT, int **d_o, int **d_x) {
        const long nt = (r_end - r_begin) / TT;
        std::vector<int> o2, x2(9, 0);
        for (int xc = 0; xc < 8; ++xc) { x2[xc] = (int)o2.size(); for (long t = 0; t < nt; ++t) if ((int)(((t * TT) % P) * 8 / P) == xc) o2.push_back((int)t); }
        x2[8] = (int)o2.size();
        CK(hipMalloc(d_o, o2.size() * 4)); CK(hipMemcpy(*d_o, o2.data(), o2.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(d_x, 36)); CK(hipMemcpy(*d_x, x2.data(), 36, hipMemcpyHostToDevice));
    };
    int *o2048, *x2048, *o1024, *x1024;
    make_order(2048, &o2048, &x2048); make_order(1024, &o1024, &x1024);
    const double rows = (double)(r_end - r_begin);
    printf("rows in tiles %.0f of %ld (%ld tiles)\n", rows, n, ntile);
    printf("%-34s %10s %10s  %s\n", "variant", "us", "us(altern)", "check");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // `launch` runs the whole step (update + SpMV or the fused kernel) leaving s in `sres` and y in y; compared with (s_ref, y_ref)
    auto run = [&](const std::string &name, double *sres, std::function<void()> launch) {
        if (!filt.empty() && name.find(filt) == std::string::npos) return;
        CK(hipMemset(y, 0xff, n * 8));
        launch(); CK(hipDeviceSynchronize());
        unsigned long long hb = 0, hb2 = 0;
        if (s_ref != sres || true) {
            CK(hipMemset(bad, 0, 8)); cmp_kernel<<<2048, 256>>>(r_begin, r_end, y, y_ref, bad); CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
            CK(hipMemset(bad, 0, 8)); cmp_kernel<<<2048, 256>>>(r_begin, r_end, sres, s_ref, bad); CK(hipMemcpy(&hb2, bad, 8, hipMemcpyDeviceToHost));
        }
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double b2b = ms * 1e3 / reps;
        double alt = 0;                 // with a 5-read 2-write streaming pass (K5's shape) between the steps, timing the step alone
        for (int i = 0; i < reps; ++i) {
            upd3<<<512, 256>>>(n, 0.5, 0.25, junk_a, junk_b, junk_a, junk_a);
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); alt += ms * 1e3;
        }
        printf("%-34s %10.1f %10.1f  %s\n", name.c_str(), b2b, alt / reps, (hb || hb2) ? "MISMATCH" : "bit-exact");
        fflush(stdout);
    };
    // chains cover the first ncol T rows of every interior plane: compare those rows only
    unsigned long long *bad2; CK(hipMalloc(&bad2, 8));
    auto run_chain = [&](const std::string &name, int TT, double *sres, bool has_own, std::function<void()> launch) {
        if (!filt.empty() && name.find(filt) == std::string::npos) return;
        CK(hipMemset(y, 0xff, n * 8));
        launch(); CK(hipDeviceSynchronize());
        const long ncol = P / TT; const int nzi = nz - 2;
        CK(hipMemset(bad2, 0, 8));
        for (int z = 0; z < nzi; ++z) {
            const long lo = r_begin + (long)z * P, hi = std::min(lo + ncol * TT, r_end);        // (the reference pair stops at r_end)
            cmp_kernel<<<64, 256>>>(lo, hi, y, y_ref, bad2);
            if (has_own) cmp_kernel<<<64, 256>>>(lo, hi, sres, s_ref, bad2);
        }
        unsigned long long hb = 0; CK(hipMemcpy(&hb, bad2, 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double b2b = ms * 1e3 / reps;
        double alt = 0;
        for (int i = 0; i < reps; ++i) {
            upd3<<<512, 256>>>(n, 0.5, 0.25, junk_a, junk_b, junk_a, junk_a);
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); alt += ms * 1e3;
        }
        printf("%-34s %10.1f %10.1f  %s  (covers %.2f %% of the tile rows)\n", name.c_str(), b2b, alt / reps, hb ? ("MISMATCH " + std::to_string(hb)).c_str() : "bit-exact",
               100.0 * (double)(ncol * TT) * nzi / rows);
        fflush(stdout);
    };
    // ================= K3 -> K4
    upd2<<<512, 256>>>(n, na, v, r, s_ref); k_win<T, W, 2><<<512, 256>>>(r_begin, nx, P, c, s_ref, y_ref, nullptr, part, d_ord, d_xs);
    CK(hipDeviceSynchronize());
    for (int grid : {512, 768, 1024}) {
        const std::string g = "/" + std::to_string(grid);
        run("K3+K4 unfused (upd2 + win)" + g, s_out, [&] { upd2<<<512, 256>>>(n, na, v, r, s_out); k_win<T, W, 2><<<grid, 256>>>(r_begin, nx, P, c, s_out, y, nullptr, part, d_ord, d_xs); });
        run("K4 alone (win dot2)" + g, s_ref, [&] { k_win<T, W, 2><<<grid, 256>>>(r_begin, nx, P, c, s_ref, y, nullptr, part, d_ord, d_xs); });
#define F2(FB, WB) run("K3+K4 fused fb" #FB " wb" #WB + g, s_out, [&] { k_winf<T, W, 2, FB, WB><<<grid, 256>>>(r_begin, nx, P, c, na, 0.0, r, v, nullptr, s_out, y, nullptr, part, d_ord, d_xs); });
        F2(8, 10) F2(4, 5) F2(2, 5) F2(1, 5) F2(1, 2)
        run("K3+K4 fused upfront T2048" + g, s_out, [&] { k_winf_up<2048, W, 2><<<grid, 256>>>(r_begin, nx, P, c, na, 0.0, r, v, nullptr, s_out, y, nullptr, part, o2048, x2048); });
        run("K3+K4 fused upfront T1024" + g, s_out, [&] { k_winf_up<1024, W, 2><<<grid, 256>>>(r_begin, nx, P, c, na, 0.0, r, v, nullptr, s_out, y, nullptr, part, o1024, x1024); });
        run("K3+K4 fused T2048 fb2 wb3" + g, s_out, [&] { k_winf<2048, W, 2, 2, 3><<<grid, 256>>>(r_begin, nx, P, c, na, 0.0, r, v, nullptr, s_out, y, nullptr, part, o2048, x2048); });
        run("K3+K4 fused T2048 fb4 wb6" + g, s_out, [&] { k_winf<2048, W, 2, 4, 6><<<grid, 256>>>(r_begin, nx, P, c, na, 0.0, r, v, nullptr, s_out, y, nullptr, part, o2048, x2048); });
    }
#define CH(TT, NVV, DOTT, ZC, GRID, NAME, OWN, I0, I1, I2, C0, C1, UU) \
    run_chain(std::string(NAME) + " T" #TT " zc" #ZC "/" #GRID, TT, s_out, OWN, [&] { k_chain<TT, W, NVV, DOTT><<<GRID, 256>>>(r_begin, n, nx, P, (int)(P / TT), nz - 2, ZC, c, C0, C1, I0, I1, I2, s_out, y, UU, part); });
    CH(2048, 1, 2, 50, 512, "K4 alone chain", false, s_ref, nullptr, nullptr, 0.0, 0.0, nullptr)
    CH(2048, 1, 2, 25, 512, "K4 alone chain", false, s_ref, nullptr, nullptr, 0.0, 0.0, nullptr)
    CH(2048, 1, 2, 25, 1024, "K4 alone chain", false, s_ref, nullptr, nullptr, 0.0, 0.0, nullptr)
    CH(1024, 1, 2, 25, 1024, "K4 alone chain", false, s_ref, nullptr, nullptr, 0.0, 0.0, nullptr)
    CH(1024, 1, 2, 50, 1024, "K4 alone chain", false, s_ref, nullptr, nullptr, 0.0, 0.0, nullptr)
    CH(4096, 1, 2, 50, 256, "K4 alone chain", false, s_ref, nullptr, nullptr, 0.0, 0.0, nullptr)
    CH(2048, 2, 2, 50, 512, "K3+K4 fused chain", true, r, v, nullptr, na, 0.0, nullptr)
    CH(2048, 2, 2, 25, 512, "K3+K4 fused chain", true, r, v, nullptr, na, 0.0, nullptr)
    CH(2048, 2, 2, 25, 1024, "K3+K4 fused chain", true, r, v, nullptr, na, 0.0, nullptr)
    CH(1024, 2, 2, 25, 1024, "K3+K4 fused chain", true, r, v, nullptr, na, 0.0, nullptr)
    CH(1024, 2, 2, 50, 1024, "K3+K4 fused chain", true, r, v, nullptr, na, 0.0, nullptr)
    CH(4096, 2, 2, 50, 256, "K3+K4 fused chain", true, r, v, nullptr, na, 0.0, nullptr)
    // ================= K1 -> K2
    upd3<<<512, 256>>>(n, ca, cb, v, r, p, s_ref); k_win<T, W, 1><<<512, 256>>>(r_begin, nx, P, c, s_ref, y_ref, r0v, part, d_ord, d_xs);
    CK(hipDeviceSynchronize());
    for (int grid : {512, 768, 1024}) {
        const std::string g = "/" + std::to_string(grid);
        run("K1+K2 unfused (upd3 + win)" + g, s_out, [&] { upd3<<<512, 256>>>(n, ca, cb, v, r, p, s_out); k_win<T, W, 1><<<grid, 256>>>(r_begin, nx, P, c, s_out, y, r0v, part, d_ord, d_xs); });
        run("K2 alone (win dot1)" + g, s_ref, [&] { k_win<T, W, 1><<<grid, 256>>>(r_begin, nx, P, c, s_ref, y, r0v, part, d_ord, d_xs); });
#define F3(FB, WB) run("K1+K2 fused fb" #FB " wb" #WB + g, s_out, [&] { k_winf<T, W, 3, FB, WB><<<grid, 256>>>(r_begin, nx, P, c, ca, cb, r, v, p, s_out, y, r0v, part, d_ord, d_xs); });
        F3(4, 5) F3(2, 5) F3(1, 5) F3(1, 2)
        run("K1+K2 fused upfront T2048" + g, s_out, [&] { k_winf_up<2048, W, 3><<<grid, 256>>>(r_begin, nx, P, c, ca, cb, r, v, p, s_out, y, r0v, part, o2048, x2048); });
        run("K1+K2 fused upfront T1024" + g, s_out, [&] { k_winf_up<1024, W, 3><<<grid, 256>>>(r_begin, nx, P, c, ca, cb, r, v, p, s_out, y, r0v, part, o1024, x1024); });
        run("K1+K2 fused T2048 fb2 wb3" + g, s_out, [&] { k_winf<2048, W, 3, 2, 3><<<grid, 256>>>(r_begin, nx, P, c, ca, cb, r, v, p, s_out, y, r0v, part, o2048, x2048); });
    }
    CH(2048, 1, 1, 50, 512, "K2 alone chain", false, s_ref, nullptr, nullptr, 0.0, 0.0, r0v)
    CH(2048, 1, 1, 25, 1024, "K2 alone chain", false, s_ref, nullptr, nullptr, 0.0, 0.0, r0v)
    CH(1024, 1, 1, 25, 1024, "K2 alone chain", false, s_ref, nullptr, nullptr, 0.0, 0.0, r0v)
    CH(2048, 3, 1, 50, 512, "K1+K2 fused chain", true, r, v, p, ca, cb, r0v)
    CH(2048, 3, 1, 25, 512, "K1+K2 fused chain", true, r, v, p, ca, cb, r0v)
    CH(2048, 3, 1, 25, 1024, "K1+K2 fused chain", true, r, v, p, ca, cb, r0v)
    CH(1024, 3, 1, 25, 1024, "K1+K2 fused chain", true, r, v, p, ca, cb, r0v)
    CH(1024, 3, 1, 50, 1024, "K1+K2 fused chain", true, r, v, p, ca, cb, r0v)
    // ================= what the consumer K5 pays behind each producer (time of the K5-like pass alone, producer launched right before it)
    if (filt.empty() || filt.find("k5") != std::string::npos) {
        double *xx; CK(hipMalloc(&xx, n * 8)); fill_vec<<<2048, 256>>>(n, xx, 41);
        auto after = [&](const std::string &name, std::function<void()> producer, std::function<void()> consumer) {
            for (int i = 0; i < 2; ++i) { producer(); consumer(); }
            double tot = 0; float ms = 0;
            for (int i = 0; i < reps; ++i) {
                producer();
                CK(hipEventRecord(e0)); consumer(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms * 1e3;
            }
            printf("%-64s %8.1f us\n", name.c_str(), tot / reps); fflush(stdout);
        };
        auto k5f = [&] { k5_like<false, true><<<512, 256>>>(n, -0.3, -0.2, p, y, r0v, xx, s_out, part); };
        auto k5r = [&] { k5_like<true, true><<<512, 256>>>(n, -0.3, -0.2, p, y, r0v, xx, s_out, part); };
        auto k5fa = [&] { k5_like<false, false><<<512, 256>>>(n, -0.3, -0.2, p, y, r0v, xx, s_out, part); };
        auto k5ra = [&] { k5_like<true, false><<<512, 256>>>(n, -0.3, -0.2, p, y, r0v, xx, s_out, part); };
        auto prod_unf = [&] { upd2<<<512, 256>>>(n, na, v, r, s_out); k_chain<2048, W, 1, 2><<<512, 256>>>(r_begin, n, nx, P, (int)(P / 2048), nz - 2, 50, c, 0.0, 0.0, s_out, nullptr, nullptr, s_out, y, nullptr, part); };
        auto prod_fus = [&] { k_chain<2048, W, 2, 2><<<512, 256>>>(r_begin, n, nx, P, (int)(P / 2048), nz - 2, 50, c, na, 0.0, r, v, nullptr, s_out, y, nullptr, part); };
        auto prod_str = [&] { upd3<<<512, 256>>>(n, 0.5, 0.25, junk_a, junk_b, junk_a, junk_a); };
        auto prod_fus_nos = [&] { k_chain<2048, W, 2, 2><<<512, 256>>>(r_begin, n, nx, P, (int)(P / 2048), nz - 2, 50, c, na, 0.0, r, v, nullptr, nullptr, y, nullptr, part); };
        auto k6 = [&] { k5_like6<<<512, 256>>>(n, -0.3, na, -0.2, p, y, r0v, v, xx, r, part); };
        {   // the producers alone, for the sums
            auto tm = [&](const char *nm, std::function<void()> f) {
                for (int i = 0; i < 3; ++i) f();
                CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-64s %8.1f us (back to back)\n", nm, ms * 1e3 / reps);
            };
            tm("producer: update + chain SpMV (five-launch flow)", prod_unf);
            tm("producer: fused chain SpMV storing s", prod_fus);
            tm("producer: fused chain SpMV NOT storing s", prod_fus_nos);
        }
        after("k5 6R+2W (s formed again from r, v) | cold", prod_str, k6);
        after("k5 6R+2W (s formed again from r, v) | behind the fused chain SpMV NOT storing s", prod_fus_nos, k6);
        after("k5 forward, nt loads | behind a streaming pass (cold)", prod_str, k5f);
        after("k5 forward, nt loads | behind update + chain SpMV reading s (five-launch flow)", prod_unf, k5f);
        after("k5 forward, nt loads | behind the fused chain SpMV writing s (three-launch flow)", prod_fus, k5f);
        after("k5 REVERSE, nt loads | behind update + chain SpMV reading s", prod_unf, k5r);
        after("k5 REVERSE, nt loads | behind the fused chain SpMV writing s", prod_fus, k5r);
        after("k5 forward, plain loads | behind update + chain SpMV reading s", prod_unf, k5fa);
        after("k5 forward, plain loads | behind the fused chain SpMV writing s", prod_fus, k5fa);
        after("k5 REVERSE, plain loads | behind the fused chain SpMV writing s", prod_fus, k5ra);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
