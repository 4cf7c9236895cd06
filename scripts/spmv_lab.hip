// Round-2 SpMV laboratory (not part of the product): variants of the plain-CSR kernel on the cfg-5 matrix
// (500x500x200 7-point, built in HBM), each checked bit for bit against a row-per-thread fold and timed per
// launch with HIP events, back to back and alternating with a 3-vector streaming kernel (the in-solve state).
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/spmv_lab.hip -o scripts/spmv_lab
//   scripts/spmv_lab [filter-substring] [reps]
//
// Variant name = <kernel>/<walk>/<grid>; walk rr = row blocks dealt round-robin over the workgroups,
// xcN = XCD-chunked walk with chunks of P/N rows (P = nx*ny; xc8 puts rows r and r +- P on one XCD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256, WAVE = 64, NWAVE = 4, CAP = 512, ITEMS = 8;
struct alignas(16) Desc { int ra, rb, pa, nn; };

__global__ void gen_rowptr(int nx, int ny, int nz, int *cnt) {
    long n = (long)nx * ny * nz;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n; g += (long)gridDim.x * blockDim.x) {
        int x = g % nx, y = (g / nx) % ny, z = g / ((long)nx * ny);
        cnt[g] = 1 + (x > 0) + (x < nx - 1) + (y > 0) + (y < ny - 1) + (z > 0) + (z < nz - 1);
    }
}
__device__ inline double hval(long k) { unsigned long h = (unsigned long)k * 0x9E3779B97F4A7C15ull; h ^= h >> 29; return (double)(h & 0xfffff) / 1048576.0 - 0.5; }
__global__ void gen_fill(int nx, int ny, int nz, const int *rp, int *ci, double *val, int randomv) {
    long n = (long)nx * ny * nz, P = (long)nx * ny;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n; g += (long)gridDim.x * blockDim.x) {
        int x = g % nx, y = (g / nx) % ny, z = g / P;
        int p = rp[g];
        auto put = [&](long c, double v) { ci[p] = (int)c; val[p] = randomv ? (c == g ? 6.5 + hval(p) : hval(p)) : v; ++p; };
        if (z > 0) put(g - P, -1);
        if (y > 0) put(g - nx, -1);
        if (x > 0) put(g - 1, -1);
        put(g, 6);
        if (x < nx - 1) put(g + 1, -1);
        if (y < ny - 1) put(g + nx, -1);
        if (z < nz - 1) put(g + P, -1);
    }
}
__global__ void fill_vec(long n, double *x, unsigned seed) {
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n; g += (long)gridDim.x * blockDim.x) x[g] = hval(g * 3 + seed);
}
__global__ void ref_spmv(long n, const int *rp, const int *ci, const double *val, const double *x, double *y) {
    for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < n; r += (long)gridDim.x * blockDim.x) {
        double acc = 0;
        for (int k = rp[r]; k < rp[r + 1]; ++k) acc = acc + x[ci[k]] * val[k];
        y[r] = acc;
    }
}
__global__ void cmp_kernel(long n, const double *a, const double *b, unsigned long long *bad) {
    unsigned long long c = 0;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n; g += (long)gridDim.x * blockDim.x)
        c += (__double_as_longlong(a[g]) != __double_as_longlong(b[g]));
    if (c) atomicAdd(bad, c);
}
// stand-in for the BLAS-1 kernels between two SpMV launches of the solve: r = r + a*v, 16 B per lane (2R + 1W)
__global__ __launch_bounds__(BLOCK) void triad(long n2, const double2 *__restrict__ a, double2 *__restrict__ b, double s) {
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n2; g += (long)gridDim.x * blockDim.x) {
        double2 u = a[g], v = b[g]; v.x = v.x + u.x * s; v.y = v.y + u.y * s; b[g] = v;
    }
}
__global__ __launch_bounds__(BLOCK) void read_only(long n4, const int4 *__restrict__ a, int *sink) {
    int acc = 0;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n4; g += (long)gridDim.x * blockDim.x) { int4 v = a[g]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678) *sink = acc;
}
__global__ void xcc_probe(int *out) {
    if (threadIdx.x == 0) { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); out[blockIdx.x] = v; }
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_down(v, off, 64);
    return v;
}
template <bool NT, class U> __device__ __forceinline__ U lds_(const U *p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }

// ---- block walks.  Every wavefront owns whole row blocks.
struct WalkArgs { int nblk; int nchunk; const int *cb; };
template <int WALK> struct Walker;
template <> struct Walker<0> {   // round-robin
    int b, step, end;
    __device__ Walker(const WalkArgs &a, int wv) : b(blockIdx.x * NWAVE + wv), step(gridDim.x * NWAVE), end(a.nblk) {}
    __device__ int next() { const int r = b < end ? b : -1; b += step; return r; }
};
template <> struct Walker<1> {   // XCD-chunked: XCD x (= blockIdx & 7) walks chunks x, x+8, ... as one concatenated list
    const int *cb; int c, nchunk, pos, nwx, c0, c1;
    __device__ Walker(const WalkArgs &a, int wv) : cb(a.cb), c(blockIdx.x & 7), nchunk(a.nchunk), pos((blockIdx.x >> 3) * NWAVE + wv), nwx((gridDim.x >> 3) * NWAVE) {
        if (c < nchunk) { c0 = cb[c]; c1 = cb[c + 1]; } else { c0 = c1 = 0; }
    }
    __device__ int next() {
        while (c < nchunk) {
            if (c0 + pos < c1) { const int r = c0 + pos; pos += nwx; return r; }
            pos -= c1 - c0; c += 8;
            if (c < nchunk) { c0 = cb[c]; c1 = cb[c + 1]; }
        }
        return -1;
    }
};

struct Stream { int cidx[ITEMS]; double vv[ITEMS]; int s, e; double uu; };
template <bool NT, bool NOU = false>
__device__ __forceinline__ void load_stream(Stream &S, const Desc d, int lane, const int *__restrict__ row_ptr, const int *__restrict__ col_idx,
                                            const double *__restrict__ val, const double *__restrict__ u) {
    const int rb = d.rb, pa = d.pa, last = max(d.nn - 1, 0);
    const int rcl = min(d.ra + lane, rb - 1);
    S.s = lds_<NT>(row_ptr + rcl) - pa; S.e = lds_<NT>(row_ptr + rcl + 1) - pa;
    if constexpr (!NOU) S.uu = lds_<NT>(u + rcl); else S.uu = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int k = min(lane + i * WAVE, last);
        S.cidx[i] = lds_<NT>(col_idx + pa + k);
        S.vv[i] = lds_<NT>(val + pa + k);
    }
}
__device__ __forceinline__ Desc uniform_desc(const Desc *__restrict__ desc, int b) {
    Desc d = desc[b];
    d.ra = __builtin_amdgcn_readfirstlane(d.ra); d.rb = __builtin_amdgcn_readfirstlane(d.rb);
    d.pa = __builtin_amdgcn_readfirstlane(d.pa); d.nn = __builtin_amdgcn_readfirstlane(d.nn);
    return d;
}
// ST: 0 plain store, 1 non-temporal, 2 sc1 (agent-scope write-through), 3 sc0 sc1 (system scope), 4 no store (ablation)
template <int ST, bool NOU = false>
__device__ __forceinline__ void fold_store(const Stream &S, const Desc d, int lane, const double *prod, double *__restrict__ y, double &d0) {
    const int r = d.ra + lane;
    if (r < d.rb) {
        double acc = 0; const int len = S.e - S.s;
        if (len <= 8) {
            double pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = prod[min(S.s + j, CAP - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < len) acc = acc + pv[j];
        } else for (int k = S.s; k < S.e; ++k) acc = acc + prod[k];
        if constexpr (ST == 1) __builtin_nontemporal_store(acc, y + r);
        else if constexpr (ST == 2) __hip_atomic_store(y + r, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if constexpr (ST == 3) __hip_atomic_store(y + r, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if constexpr (ST == 4) { if (acc == 1.2345e300) y[r] = acc; }
        else y[r] = acc;
        d0 = d0 + (NOU ? 1.0 : S.uu) * acc;
    }
}

// the round-1 product kernel's structure (DOT = 1), with selectable walk
template <int WALK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_base(WalkArgs wa, const Desc *__restrict__ desc, const int *__restrict__ row_ptr,
                                                const int *__restrict__ col_idx, const double *__restrict__ val, const double *__restrict__ x,
                                                double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part) {
    __shared__ double prod_all[NWAVE][CAP]; __shared__ double red[NWAVE];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *prod = prod_all[wv]; double d0 = 0;
    Walker<WALK> w(wa, wv);
    for (int b = w.next(); b >= 0; b = w.next()) {
        const Desc d = uniform_desc(desc, b);
        Stream S; load_stream<NT>(S, d, lane, row_ptr, col_idx, val, u);
        double xg[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) xg[i] = x[S.cidx[i]];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { const int k = lane + i * WAVE; if (k < d.nn) prod[k] = xg[i] * S.vv[i]; }
        wave_lds_fence();
        fold_store<(NT ? 1 : 0)>(S, d, lane, prod, y, d0);
        wave_lds_fence();
    }
    d0 = wave_sum(d0);
    if (lane == 0) red[wv] = d0;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// software-pipelined: the stream of block i+1 and the descriptor of block i+2 are requested before the
// products of block i are formed; all loads unconditional on clamped (valid) addresses so the loop body is
// straight-line code and the waits are counted, not drained
template <int WALK, bool NT, int ST = (NT ? 1 : 0), int ABL = 0>
__global__ __launch_bounds__(BLOCK) void k_pipe(WalkArgs wa, const Desc *__restrict__ desc, const int *__restrict__ row_ptr,
                                                const int *__restrict__ col_idx, const double *__restrict__ val, const double *__restrict__ x,
                                                double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part) {
    __shared__ double prod_all[NWAVE][CAP]; __shared__ double red[NWAVE];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *prod = prod_all[wv]; double d0 = 0;
    Walker<WALK> w(wa, wv);
    int b = w.next();
    if (b >= 0) {
        Desc d = uniform_desc(desc, b);
        Stream S; load_stream<NT, (ABL & 1) != 0>(S, d, lane, row_ptr, col_idx, val, u);
        int bn = w.next();
        Desc dn = uniform_desc(desc, bn >= 0 ? bn : b);
        while (true) {
            double xg[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) xg[i] = (ABL & 2) ? 1.0 + S.cidx[i] * 1e-30 : x[S.cidx[i]];
            Stream SN; load_stream<NT, (ABL & 1) != 0>(SN, dn, lane, row_ptr, col_idx, val, u);
            const int bnn = bn >= 0 ? w.next() : -1;
            const Desc dnn = desc[bnn >= 0 ? bnn : b];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) prod[lane + i * WAVE] = xg[i] * S.vv[i];   // slots >= nn hold junk nobody reads
            wave_lds_fence();
            fold_store<ST, (ABL & 1) != 0>(S, d, lane, prod, y, d0);
            wave_lds_fence();
            if (bn < 0) break;
            S = SN; d = dn; b = bn; bn = bnn;
            dn.ra = __builtin_amdgcn_readfirstlane(dnn.ra); dn.rb = __builtin_amdgcn_readfirstlane(dnn.rb);
            dn.pa = __builtin_amdgcn_readfirstlane(dnn.pa); dn.nn = __builtin_amdgcn_readfirstlane(dnn.nn);
        }
    }
    d0 = wave_sum(d0);
    if (lane == 0) red[wv] = d0;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}


// ---- round 3: the TA (vector-memory address unit) is ~87 % busy in the product kernel (profiles/r03_pmc_l2_ta.json), and a
// 64-lane load costs it about the same for 4, 8 or 16 bytes per lane (scripts/micro/ta_rate.hip).  k_wide reads the
// stream with 16 bytes per lane: lane l takes entries 4l .. 4l+3 of the block's 16-byte-aligned window (2 loads of
// col_idx, 4 of val per 512 entries instead of 8 + 8), gathers and multiplies the same entries, writes the products to
// LDS in window order; the fold is the product kernel's.  EQ: rows of equal length take their extents from the descriptor.
template <int WALK, int ST, bool EQ, int NTS>
__global__ __launch_bounds__(BLOCK) void k_wide(WalkArgs wa, const Desc *__restrict__ desc, const int *__restrict__ row_ptr,
                                                const int *__restrict__ col_idx, const double *__restrict__ val, const double *__restrict__ x,
                                                double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part) {
    __shared__ __attribute__((aligned(16))) double prod_all[NWAVE][CAP + 8]; __shared__ double red[NWAVE];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *prod = prod_all[wv]; double d0 = 0;
    Walker<WALK> w(wa, wv);
    for (int b = w.next(); b >= 0; b = w.next()) {
        Desc d = uniform_desc(desc, b);
        const bool eq = EQ && ((d.rb >> 30) & 1);                                // (scalar) flagged copy of the descriptors: rows of equal length
        d.rb &= 0x3fffffff;
        const int shift = d.pa & 3, base = d.pa - shift, tot = d.nn + shift;     // the aligned window [base, base + tot)
        const int rcl = min(d.ra + lane, d.rb - 1);
        int s, e;
        if (eq) { const int L = d.nn / (d.rb - d.ra); s = (rcl - d.ra) * L; e = s + L; }
        else { s = row_ptr[rcl] - d.pa; e = row_ptr[rcl + 1] - d.pa; }
        const double uu = u[rcl];
        const int lastq = max(tot - 1, 0) >> 2;                                  // last 4-entry group of the window
        int4 c[2]; double2 v[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = min(lane + i * WAVE, lastq);
            const int4 *cp = reinterpret_cast<const int4 *>(col_idx + base) + q;
            const double2 *vp = reinterpret_cast<const double2 *>(val + base) + 2 * q;
            if (NTS) {
                typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                const u4v a = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(cp));
                const u4v b0 = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(vp)), b1 = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(vp + 1));
                __builtin_memcpy(&c[i], &a, 16); __builtin_memcpy(&v[i][0], &b0, 16); __builtin_memcpy(&v[i][1], &b1, 16);
            } else { c[i] = *cp; v[i][0] = vp[0]; v[i][1] = vp[1]; }
        }
        double xg[8];
#pragma unroll
        for (int i = 0; i < 2; ++i) { xg[4 * i] = x[c[i].x]; xg[4 * i + 1] = x[c[i].y]; xg[4 * i + 2] = x[c[i].z]; xg[4 * i + 3] = x[c[i].w]; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = 4 * (lane + i * WAVE);
            double2 p0, p1;
            p0.x = xg[4 * i] * v[i][0].x; p0.y = xg[4 * i + 1] * v[i][0].y; p1.x = xg[4 * i + 2] * v[i][1].x; p1.y = xg[4 * i + 3] * v[i][1].y;
            if (k < tot) { *reinterpret_cast<double2 *>(prod + k) = p0; *reinterpret_cast<double2 *>(prod + k + 2) = p1; }
        }
        wave_lds_fence();
        const int r = d.ra + lane;
        if (r < d.rb) {
            double acc = 0; const int len = e - s; const double *pr = prod + shift;
            if (len <= 8) {
                double pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = pr[min(s + j, CAP + 3)];
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < len) acc = acc + pv[j];
            } else for (int k = s; k < e; ++k) acc = acc + pr[k];
            if constexpr (ST == 1) __builtin_nontemporal_store(acc, y + r); else y[r] = acc;
            d0 = d0 + uu * acc;
        }
        wave_lds_fence();
    }
    d0 = wave_sum(d0);
    if (lane == 0) red[wv] = d0;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// the product kernel with the equal-length shortcut (what csrc/spmv.hip runs on cfg 5): the baseline of this round
template <int WALK>
__global__ __launch_bounds__(BLOCK) void k_prod(WalkArgs wa, const Desc *__restrict__ desc, const int *__restrict__ row_ptr,
                                                const int *__restrict__ col_idx, const double *__restrict__ val, const double *__restrict__ x,
                                                double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part) {
    __shared__ double prod_all[NWAVE][CAP]; __shared__ double red[NWAVE];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *prod = prod_all[wv]; double d0 = 0;
    Walker<WALK> w(wa, wv);
    for (int b = w.next(); b >= 0; b = w.next()) {
        Desc d = uniform_desc(desc, b);
        const bool eq = (d.rb >> 30) & 1;
        d.rb &= 0x3fffffff;
        const int rcl = min(d.ra + lane, d.rb - 1);
        Stream S;
        if (eq) { const int L = d.nn / (d.rb - d.ra); S.s = (rcl - d.ra) * L; S.e = S.s + L; }
        else { S.s = row_ptr[rcl] - d.pa; S.e = row_ptr[rcl + 1] - d.pa; }
        S.uu = u[rcl];
        const int last = max(d.nn - 1, 0);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { const int k = min(lane + i * WAVE, last); S.cidx[i] = col_idx[d.pa + k]; S.vv[i] = val[d.pa + k]; }
        double xg[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) xg[i] = x[S.cidx[i]];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { const int k = lane + i * WAVE; if (k < d.nn) prod[k] = xg[i] * S.vv[i]; }
        wave_lds_fence();
        fold_store<0>(S, d, lane, prod, y, d0);
        wave_lds_fence();
    }
    d0 = wave_sum(d0);
    if (lane == 0) red[wv] = d0;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// 128-row blocks, up to 1024 entries per wavefront: twice the bytes in flight per wave (4 + 8 sixteen-byte loads), two rows
// per lane in the fold phase
template <int WALK, int ST>
__global__ __launch_bounds__(BLOCK) void k_wide2(WalkArgs wa, const Desc *__restrict__ desc, const int *__restrict__ row_ptr,
                                                 const int *__restrict__ col_idx, const double *__restrict__ val, const double *__restrict__ x,
                                                 double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part) {
    constexpr int CAP2 = 1024;
    __shared__ __attribute__((aligned(16))) double prod_all[NWAVE][CAP2 + 8]; __shared__ double red[NWAVE];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *prod = prod_all[wv]; double d0 = 0;
    Walker<WALK> w(wa, wv);
    for (int b = w.next(); b >= 0; b = w.next()) {
        Desc d = uniform_desc(desc, b);
        const bool eq = (d.rb >> 30) & 1;
        d.rb &= 0x3fffffff;
        const int shift = d.pa & 3, base = d.pa - shift, tot = d.nn + shift;
        int s[2], e[2]; double uu[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rcl = min(d.ra + lane + 64 * h, d.rb - 1);
            if (eq) { const int L = d.nn / (d.rb - d.ra); s[h] = (rcl - d.ra) * L; e[h] = s[h] + L; }
            else { s[h] = row_ptr[rcl] - d.pa; e[h] = row_ptr[rcl + 1] - d.pa; }
            uu[h] = u[rcl];
        }
        const int lastq = max(tot - 1, 0) >> 2;
        int4 c[4]; double2 v[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = min(lane + i * WAVE, lastq);
            c[i] = reinterpret_cast<const int4 *>(col_idx + base)[q];
            v[i][0] = reinterpret_cast<const double2 *>(val + base)[2 * q]; v[i][1] = reinterpret_cast<const double2 *>(val + base)[2 * q + 1];
        }
        double xg[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) { xg[4 * i] = x[c[i].x]; xg[4 * i + 1] = x[c[i].y]; xg[4 * i + 2] = x[c[i].z]; xg[4 * i + 3] = x[c[i].w]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * (lane + i * WAVE);
            double2 p0, p1;
            p0.x = xg[4 * i] * v[i][0].x; p0.y = xg[4 * i + 1] * v[i][0].y; p1.x = xg[4 * i + 2] * v[i][1].x; p1.y = xg[4 * i + 3] * v[i][1].y;
            if (k < tot) { *reinterpret_cast<double2 *>(prod + k) = p0; *reinterpret_cast<double2 *>(prod + k + 2) = p1; }
        }
        wave_lds_fence();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = d.ra + lane + 64 * h;
            if (r < d.rb) {
                double acc = 0; const int len = e[h] - s[h]; const double *pr = prod + shift;
                if (len <= 8) {
                    double pv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) pv[j] = pr[min(s[h] + j, CAP2 + 3)];
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j < len) acc = acc + pv[j];
                } else for (int k = s[h]; k < e[h]; ++k) acc = acc + pr[k];
                if constexpr (ST == 1) __builtin_nontemporal_store(acc, y + r); else y[r] = acc;
                d0 = d0 + uu[h] * acc;
            }
        }
        wave_lds_fence();
    }
    d0 = wave_sum(d0);
    if (lane == 0) red[wv] = d0;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

struct Variant { std::string name; std::function<void()> launch; };

int main(int argc, char **argv) {
    const char *filter = argc > 1 ? argv[1] : "";
    const int reps = argc > 2 ? atoi(argv[2]) : 12;
    const int nx = 500, ny = 500, nz = 200; const long n = (long)nx * ny * nz, P = (long)nx * ny;
    int *cnt, *rp, *ci; double *val, *x, *y, *yref, *u, *part, *t1, *t2;
    CK(hipMalloc(&cnt, n * 4)); CK(hipMalloc(&rp, (n + 1) * 4));
    gen_rowptr<<<4096, 256>>>(nx, ny, nz, cnt);
    std::vector<int> hc(n), hrp(n + 1);
    CK(hipMemcpy(hc.data(), cnt, n * 4, hipMemcpyDeviceToHost));
    hrp[0] = 0; for (long i = 0; i < n; ++i) hrp[i + 1] = hrp[i] + hc[i];
    const long nnz = hrp[n];
    CK(hipMemcpy(rp, hrp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ci, (nnz + 64) * 4)); CK(hipMalloc(&val, (nnz + 64) * 8)); CK(hipMemset(ci, 0, (nnz + 64) * 4)); CK(hipMemset(val, 0, (nnz + 64) * 8));
    for (double **p : {&x, &y, &yref, &u, &t1, &t2}) CK(hipMalloc(p, n * 8));
    CK(hipMalloc(&part, 8192 * 8));
    gen_fill<<<4096, 256>>>(nx, ny, nz, rp, ci, val, 1);
    fill_vec<<<4096, 256>>>(n, x, 1); fill_vec<<<4096, 256>>>(n, u, 2); fill_vec<<<4096, 256>>>(n, t1, 3); fill_vec<<<4096, 256>>>(n, t2, 4);
    ref_spmv<<<8192, 256>>>(n, rp, ci, val, x, yref);
    CK(hipDeviceSynchronize());
    const double bytes = nnz * 12.0 + (n + 1) * 4.0 + 2.0 * n * 8;
    printf("n %ld nnz %ld algorithmic bytes %.0f\n", n, nnz, bytes);

    {   // placement probe: is blockIdx & 7 the XCD?
        int *dx; CK(hipMalloc(&dx, 1024 * 4)); xcc_probe<<<1024, 64>>>(dx);
        std::vector<int> hx(1024); CK(hipMemcpy(hx.data(), dx, 4096, hipMemcpyDeviceToHost));
        int consistent = 1; for (int i = 8; i < 1024; ++i) consistent &= ((hx[i] & 15) == (hx[i & 7] & 15));
        printf("XCC_ID of blocks 0..7:"); for (int i = 0; i < 8; ++i) printf(" %d", hx[i] & 15); printf("  b%%8 consistent over 1024 blocks: %d\n", consistent);
    }

    // 64-row blocks
    std::vector<Desc> hd;
    for (long r = 0; r < n; r += 64) { long e = std::min<long>(r + 64, n); hd.push_back({(int)r, (int)e, hrp[r], hrp[e] - hrp[r]}); }
    const int nblk = (int)hd.size();
    Desc *dd; CK(hipMalloc(&dd, hd.size() * sizeof(Desc))); CK(hipMemcpy(dd, hd.data(), hd.size() * sizeof(Desc), hipMemcpyHostToDevice));
    // flagged copy: bit 30 of rb = all rows of the block have the same length
    std::vector<Desc> hde(hd);
    long neq = 0;
    for (auto &d : hde) {
        const int rows = d.rb - d.ra; bool eq = rows > 0 && d.nn % rows == 0;
        for (int r = d.ra; eq && r < d.rb; ++r) eq = hrp[r + 1] - hrp[r] == d.nn / rows;
        if (eq) { d.rb |= 1 << 30; ++neq; }
    }
    printf("equal-length blocks: %ld of %d\n", neq, nblk);
    // 128-row blocks for k_wide2 (flagged the same way)
    std::vector<Desc> hd2;
    for (long r = 0; r < n; r += 128) {
        long e = std::min<long>(r + 128, n); Desc d{(int)r, (int)e, hrp[r], hrp[e] - hrp[r]};
        const int rows = d.rb - d.ra; bool eq = d.nn % rows == 0;
        for (int q = d.ra; eq && q < d.rb; ++q) eq = hrp[q + 1] - hrp[q] == d.nn / rows;
        if (eq) d.rb |= 1 << 30;
        hd2.push_back(d);
    }
    const int nblk2 = (int)hd2.size();
    Desc *dd2; CK(hipMalloc(&dd2, hd2.size() * sizeof(Desc))); CK(hipMemcpy(dd2, hd2.data(), hd2.size() * sizeof(Desc), hipMemcpyHostToDevice));
    Desc *dde; CK(hipMalloc(&dde, hde.size() * sizeof(Desc))); CK(hipMemcpy(dde, hde.data(), hde.size() * sizeof(Desc), hipMemcpyHostToDevice));
    // chunk tables: chunk c = blocks whose first row is in [c*G, (c+1)*G)
    auto make_chunks = [&](double G, int **dcb, int *nchunk) {
        std::vector<int> cb; cb.push_back(0);
        long c = 1;
        for (int b = 0; b < nblk; ++b) while ((double)hd[b].ra >= c * G) { cb.push_back(b); ++c; }
        cb.push_back(nblk);
        *nchunk = (int)cb.size() - 1;
        CK(hipMalloc(dcb, cb.size() * 4)); CK(hipMemcpy(*dcb, cb.data(), cb.size() * 4, hipMemcpyHostToDevice));
    };
    int *cb8, *cb16, *cb4, *cbs, n8, n16, n4, ns, *cb64, *cb32, n64, n32, *cbs2, *cbs8, *cbs16, ns2, ns8, ns16, *cbs1, ns1, *cbs3, ns3, *cbs6, ns6;
    make_chunks(1024.0, &cbs1, &ns1); make_chunks(3072.0, &cbs3, &ns3); make_chunks(6144.0, &cbs6, &ns6);
    make_chunks(P / 64.0, &cb64, &n64); make_chunks(P / 32.0, &cb32, &n32); make_chunks(2048.0, &cbs2, &ns2); make_chunks(8192.0, &cbs8, &ns8); make_chunks(16384.0, &cbs16, &ns16);
    make_chunks(P / 8.0, &cb8, &n8); make_chunks(P / 16.0, &cb16, &n16); make_chunks(P / 4.0, &cb4, &n4); make_chunks(4096.0, &cbs, &ns);

    std::vector<Variant> vs;
    for (int grid : {1024, 768, 512}) {
        const std::string g = "/" + std::to_string(grid);
        WalkArgs xs1{nblk, ns1, cbs1}, xs3{nblk, ns3, cbs3}, xs6{nblk, ns6, cbs6}; WalkArgs rr{nblk, 0, nullptr}, x8{nblk, n8, cb8}, x16{nblk, n16, cb16}, x4{nblk, n4, cb4}, xs{nblk, ns, cbs}, x64{nblk, n64, cb64}, x32{nblk, n32, cb32}, xs2{nblk, ns2, cbs2}, xs8{nblk, ns8, cbs8}, xs16{nblk, ns16, cbs16};
#define V(NAME, K, WALKARGS) vs.push_back({std::string(NAME) + g, [=]() { K<<<grid, BLOCK>>>(WALKARGS, dd, rp, ci, val, x, y, u, part); }})
#define V2(NAME, K) vs.push_back({std::string(NAME) + g, [=]() { K<<<grid, BLOCK>>>(WalkArgs{nblk2, 0, nullptr}, dd2, rp, ci, val, x, y, u, part); }})
#define VE(NAME, K, WALKARGS) vs.push_back({std::string(NAME) + g, [=]() { K<<<grid, BLOCK>>>(WALKARGS, dde, rp, ci, val, x, y, u, part); }})
        VE("prod/rr", (k_prod<0>), rr);
        V2("wide2/rr", (k_wide2<0, 0>));
        V2("wide2-stnt/rr", (k_wide2<0, 1>));
        VE("wide/rr", (k_wide<0, 0, true, 0>), rr);
        V("wide-noeq/rr", (k_wide<0, 0, false, 0>), rr);
        VE("wide-nts/rr", (k_wide<0, 0, true, 1>), rr);
        VE("wide-stnt/rr", (k_wide<0, 1, true, 0>), rr);
        VE("wide/xc8", (k_wide<1, 0, true, 0>), x8);
        VE("wide/xc4096rows", (k_wide<1, 0, true, 0>), xs);
        VE("prod/xc4096rows", (k_prod<1>), xs);
        VE("wide/xc1024rows", (k_wide<1, 0, true, 0>), xs1);
        VE("wide/xc2048rows", (k_wide<1, 0, true, 0>), xs2);
        VE("wide/xc3072rows", (k_wide<1, 0, true, 0>), xs3);
        VE("wide/xc6144rows", (k_wide<1, 0, true, 0>), xs6);
        VE("wide/xc8192rows", (k_wide<1, 0, true, 0>), xs8);
        VE("wide/xc16384rows", (k_wide<1, 0, true, 0>), xs16);
        VE("wide/xc16", (k_wide<1, 0, true, 0>), x16);
        VE("wide-stnt/xc4096rows", (k_wide<1, 1, true, 0>), xs);
        VE("prod/xc2048rows", (k_prod<1>), xs2);
        VE("prod/xc8192rows", (k_prod<1>), xs8);
        if (grid != 1024) continue;
        V("base/rr", (k_base<0, false>), rr);
        V("pipe/rr", (k_pipe<0, false>), rr);
        V("pipe-stnt/rr", (k_pipe<0, false, 1>), rr);
        V("pipe/xc64", (k_pipe<1, false>), x64);
        V("pipe-stnt/xc64", (k_pipe<1, false, 1>), x64);
        V("pipe-stsc1/xc64", (k_pipe<1, false, 2>), x64);
        V("pipe/xc32", (k_pipe<1, false>), x32);
        V("pipe/xc8", (k_pipe<1, false>), x8);
        V("pipe-stnt/xc8", (k_pipe<1, false, 1>), x8);
        V("pipe-stsc1/xc8", (k_pipe<1, false, 2>), x8);
        V("pipe/xc2048rows", (k_pipe<1, false>), xs2);
        V("pipe/xc4096rows", (k_pipe<1, false>), xs);
        V("pipe/xc8192rows", (k_pipe<1, false>), xs8);
        V("pipe/xc16384rows", (k_pipe<1, false>), xs16);
#undef V
    }

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long *dbad; CK(hipMalloc(&dbad, 8));
    {   // ceilings on this box
        float ms;
        for (int w = 0; w < 2; ++w) read_only<<<2048, BLOCK>>>(nnz * 8 / 16, (const int4 *)val, (int *)part);
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) read_only<<<2048, BLOCK>>>(nnz * 8 / 16, (const int4 *)val, (int *)part); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("read-only 16B/lane stream of val[] (%.2f GB): %.0f GB/s\n", nnz * 8 / 1e9, nnz * 8.0 * 5 / ms / 1e6);
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) triad<<<2048, BLOCK>>>(n / 2, (const double2 *)t1, (double2 *)t2, 1e-9); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("triad 2R+1W on n doubles: %.0f GB/s\n", 3.0 * n * 8 * 5 / ms / 1e6);
    }
    printf("%-32s %10s %10s %8s %10s %8s %s\n", "variant", "b2b us", "GB/s", "frac", "altern us", "frac", "check");
    for (auto &v : vs) {
        if (strstr(v.name.c_str(), filter) == nullptr) continue;
        CK(hipMemset(y, 0xff, n * 8)); CK(hipMemset(dbad, 0, 8));
        v.launch();
        cmp_kernel<<<4096, 256>>>(n, y, yref, dbad);
        unsigned long long bad; CK(hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost));
        CK(hipDeviceSynchronize());
        if (hipGetLastError() != hipSuccess) { printf("%s: launch error\n", v.name.c_str()); return 1; }
        // back to back
        for (int w = 0; w < 2; ++w) v.launch();
        float ms, tot = 0;
        CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) v.launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); const double b2b = ms / reps;
        // alternating with a streaming kernel over two other vectors (per-launch events)
        for (int i = 0; i < reps; ++i) {
            triad<<<512, BLOCK>>>(n / 2, (const double2 *)t1, (double2 *)t2, 1e-9);
            CK(hipEventRecord(e0)); v.launch(); CK(hipEventRecord(e1));
            triad<<<512, BLOCK>>>(n / 2, (const double2 *)t2, (double2 *)t1, 1e-9);
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
        }
        const double alt = tot / reps;
        printf("%-32s %10.1f %10.0f %8.3f %10.1f %8.3f %s\n", v.name.c_str(), b2b * 1e3, bytes / b2b / 1e6, bytes / b2b / 1e6 / 8000.0,
               alt * 1e3, bytes / alt / 1e6 / 8000.0, bad ? (strstr(v.name.c_str(), "ABL") ? "(ablation)" : "MISMATCH") : "bit-exact");
        fflush(stdout);
    }
    return 0;
}
