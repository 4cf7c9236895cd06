// Round-3 micro-benchmark (not part of the product): what would an LDS window for the NEAR columns buy the
// two-rows-per-lane uniform path (spmv_pair2_kernel, full_uniform_block) on cfg 5's pattern?
//
// The matrix is the 7-diagonal operator y[r] = sum_j v_j x[r + o_j], o = (-P, -nx, -1, 0, 1, nx, P), rows [P, n - P)
// (the uniform path treats seam rows like the others, so the diagonals alone carry the memory behaviour).
//   gl5   : the product's shape — a wavefront owns 128-row blocks, 5 global 16-byte loads per lane (the +-1 columns
//           by wavefront shifts), next block's loads issued before this block's products, y by 16-byte stores
//   win   : a workgroup owns T consecutive rows, stages x[ts - W, ts + T + W) in LDS with 16-byte loads (W >= nx), takes
//           -nx / centre / +-1 / +nx from LDS and only the +-P windows from global memory:
//           per 128 rows (T + 2W)/T + 2 global loads instead of 5
// Both fold every row left to right like the reference; all variants are compared bit for bit with a row-per-thread fold.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/micro/stencil_window.hip -o /tmp/stencil_window && /tmp/stencil_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256, WAVE = 64;
struct alignas(16) D2 { double lo, hi; };

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
__global__ void ref_kernel(long r0, long r1, int nx, long P, Coef c, const double *x, double *y) {
    for (long r = r0 + blockIdx.x * (long)blockDim.x + threadIdx.x; r < r1; r += (long)gridDim.x * blockDim.x)
        y[r] = fold7(c, x[r - P], x[r - nx], x[r - 1], x[r], x[r + 1], x[r + nx], x[r + P]);
}
__global__ void cmp_kernel(long r0, long r1, const double *a, const double *b, unsigned long long *bad) {
    unsigned long long c = 0;
    for (long g = r0 + blockIdx.x * (long)blockDim.x + threadIdx.x; g < r1; g += (long)gridDim.x * blockDim.x)
        c += (__double_as_longlong(a[g]) != __double_as_longlong(b[g]));
    if (c) atomicAdd(bad, c);
}
__global__ void triad_kernel(long n, double a, const double *v, double *r) {        // the in-solve neighbour: r += a v, 16 B per lane
    const long n2 = n >> 1;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n2; g += (long)gridDim.x * blockDim.x) {
        D2 rv = reinterpret_cast<D2 *>(r)[g]; const D2 vv = reinterpret_cast<const D2 *>(v)[g];
        rv.lo = rv.lo + a * vv.lo; rv.hi = rv.hi + a * vv.hi;
        __builtin_nontemporal_store(rv.lo, &r[2 * g]); __builtin_nontemporal_store(rv.hi, &r[2 * g + 1]);
    }
}
__device__ __forceinline__ double shift_up(double prev0, double v) {        // lane l <- lane l - 1
    const long long o = __double_as_longlong(prev0), q = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)q, 0x138, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(q >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double shift_down(double next63, double v) {     // lane l <- lane l + 1
    const long long o = __double_as_longlong(next63), q = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)q, 0x130, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(q >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ D2 ldg2(const double *p) { return *reinterpret_cast<const D2 *>(p); }
__device__ __forceinline__ void stnt2(double *p, D2 v) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 w; __builtin_memcpy(&w, &v, 16);
    __builtin_nontemporal_store(w, reinterpret_cast<u4 *>(p));
}

// ---- gl5: the product's shape.  DOT: 0 none, 1 operand u (another vector), 2 operand = x
template <int DOT>
__global__ __launch_bounds__(BLOCK) void k_gl5(long r_begin, long r_end, int nx, long P, Coef c, const double *__restrict__ x,
                                               double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long nblk = (r_end - r_begin) >> 7;             // full 128-row blocks only (the driver makes the range a multiple)
    double d0 = 0.0;
    struct L { D2 m, a, cc, b, p, uu; double l, r; };
    auto issue = [&](long b, L &v) {
        const long r0 = r_begin + (b << 7) + 2 * lane;
        v.m = ldg2(x + r0 - P); v.a = ldg2(x + r0 - nx); v.cc = ldg2(x + r0); v.b = ldg2(x + r0 + nx); v.p = ldg2(x + r0 + P);
        v.l = x[r_begin + (b << 7) - 1]; v.r = x[r_begin + (b << 7) + 128];          // wave-uniform: scalar loads
        if (DOT == 1) v.uu = ldg2(u + r0);
    };
    long b = blockIdx.x * 4 + wv; const long step = (long)gridDim.x * 4;
    L cur, nxt;
    if (b < nblk) issue(b, cur);
    for (; b < nblk; b += step) {
        const bool more = b + step < nblk;
        if (more) issue(b + step, nxt);
        const long r0 = r_begin + (b << 7) + 2 * lane;
        const double xl = shift_up(cur.l, cur.cc.hi), xr = shift_down(cur.r, cur.cc.lo);
        D2 o;
        o.lo = fold7(c, cur.m.lo, cur.a.lo, xl, cur.cc.lo, cur.cc.hi, cur.b.lo, cur.p.lo);
        o.hi = fold7(c, cur.m.hi, cur.a.hi, cur.cc.lo, cur.cc.hi, xr, cur.b.hi, cur.p.hi);
        stnt2(y + r0, o);
        if (DOT == 1) { d0 = d0 + o.lo * cur.uu.lo; d0 = d0 + o.hi * cur.uu.hi; }
        if (DOT == 2) { d0 = d0 + o.lo * cur.cc.lo; d0 = d0 + o.hi * cur.cc.hi; }
        if (more) cur = nxt;
    }
    if (DOT) {
        for (int o = 32; o > 0; o >>= 1) d0 += __shfl_xor(d0, o, 64);
        if (lane == 0) part[blockIdx.x * 4 + wv] = d0;
    }
}

// ---- win: T rows per workgroup step, window half-width W (>= nx, even), both multiples of 512 / 2
template <int T, int W, int DOT, int PRE>
__global__ __launch_bounds__(BLOCK) void k_win(long r_begin, long r_end, int nx, long P, long n, Coef c, const double *__restrict__ x,
                                               double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part,
                                               const int *__restrict__ order, const int *__restrict__ xstart) {
    constexpr int NW = (T + 2 * W) / 2 / BLOCK;          // 16-byte window pieces per lane
    constexpr int NQ = T / 512;                          // 128-row blocks per wavefront and tile
    static_assert((T + 2 * W) % (2 * BLOCK) == 0 && T % 512 == 0, "shape");
    __shared__ __attribute__((aligned(16))) double win[T + 2 * W];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long ntile = (r_end - r_begin) / T;
    double d0 = 0.0;
    struct Far { D2 m[NQ], p[NQ], uu[NQ]; };
    typedef unsigned u4w __attribute__((ext_vector_type(4)));
    u4w wreg[NW];
    Far f;
    auto issue = [&](long t) {
        const long ts = r_begin + t * T;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            long g = ts - W + 2 * (long)(tid + i * BLOCK);
            g = g > n - 2 ? n - 2 : g;                                   // the last tile's window may pass the end of x
            wreg[i] = *reinterpret_cast<const u4w *>(x + g);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const long r0 = ts + ((q * 4 + wv) << 7) + 2 * lane;
            f.m[q] = ldg2(x + r0 - P); f.p[q] = ldg2(x + r0 + P);
            if (DOT == 1) f.uu[q] = ldg2(u + r0);
        }
    };
    // walk: round-robin over all tiles, or (order != nullptr) XCD b & 7 walks ITS list of tiles — the tiles whose phase within
    // the plane period falls into its eighth, in row order — so that a +-P window was some tile's centre window on the same L2
    long s = order ? xstart[blockIdx.x & 7] + (blockIdx.x >> 3) : blockIdx.x;
    const long send = order ? xstart[(blockIdx.x & 7) + 1] : ntile, sstep = order ? gridDim.x >> 3 : gridDim.x;
    if (PRE && s < send) issue(order ? order[s] : s);
    for (; s < send; s += sstep) {
        const long t = order ? order[s] : s;
        if (!PRE) issue(t);
        const long ts = r_begin + t * T;
        __syncthreads();                                                // the previous tile's LDS reads are done
#pragma unroll
        for (int i = 0; i < NW; ++i) *reinterpret_cast<u4w *>(&win[2 * (tid + i * BLOCK)]) = wreg[i];
        __syncthreads();
        Far g = f;
        if (PRE && s + sstep < send) issue(order ? order[s + sstep] : s + sstep);         // next tile's loads fly over this tile's products
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int li = W + ((q * 4 + wv) << 7) + 2 * lane;          // LDS index of x[r0]
            const D2 cc = *reinterpret_cast<const D2 *>(&win[li]);
            const D2 a = *reinterpret_cast<const D2 *>(&win[li - nx]);
            const D2 b = *reinterpret_cast<const D2 *>(&win[li + nx]);
            const double xl = win[li - 1], xr = win[li + 2];
            D2 o;
            o.lo = fold7(c, g.m[q].lo, a.lo, xl, cc.lo, cc.hi, b.lo, g.p[q].lo);
            o.hi = fold7(c, g.m[q].hi, a.hi, cc.lo, cc.hi, xr, b.hi, g.p[q].hi);
            stnt2(y + ts + ((q * 4 + wv) << 7) + 2 * lane, o);
            if (DOT == 1) { d0 = d0 + o.lo * g.uu[q].lo; d0 = d0 + o.hi * g.uu[q].hi; }
            if (DOT == 2) { d0 = d0 + o.lo * cc.lo; d0 = d0 + o.hi * cc.hi; }
        }
    }
    if (DOT) {
        for (int o = 32; o > 0; o >>= 1) d0 += __shfl_xor(d0, o, 64);
        if (lane == 0) part[blockIdx.x * 4 + wv] = d0;
    }
}

// ---- winv: the same with a value PER ENTRY (variable coefficients: 7 doubles per row streamed from val[], row-major):
// a wavefront loads the 896 values of its 128-row block with seven 16-byte loads, passes them through its LDS slice
// (stream order in, row order out) one block ahead of the fold
__global__ void fill_vals(long n, double *v) {
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n; g += (long)gridDim.x * blockDim.x) v[g] = hval(g * 5 + 11);
}
__global__ void refv_kernel(long r0, long r1, int nx, long P, const double *val, const double *x, double *y) {
    for (long r = r0 + blockIdx.x * (long)blockDim.x + threadIdx.x; r < r1; r += (long)gridDim.x * blockDim.x) {
        const double *v = val + (r - r0) * 7;
        double acc = 0.0;
        acc = acc + x[r - P] * v[0]; acc = acc + x[r - nx] * v[1]; acc = acc + x[r - 1] * v[2]; acc = acc + x[r] * v[3];
        acc = acc + x[r + 1] * v[4]; acc = acc + x[r + nx] * v[5]; acc = acc + x[r + P] * v[6];
        y[r] = acc;
    }
}
// gather version of the product's offset-code kernel shape: lane = row of a 64-row block, 7 8-byte gathers, values via LDS
template <int T, int W, int DOT>
__global__ __launch_bounds__(BLOCK) void k_winv(long r_begin, long r_end, int nx, long P, long n, const double *__restrict__ val, const double *__restrict__ x,
                                                double *__restrict__ y, const double *__restrict__ u, double *__restrict__ part,
                                                const int *__restrict__ order, const int *__restrict__ xstart) {
    constexpr int NW = (T + 2 * W) / 2 / BLOCK, NQ = T / 512;
    typedef unsigned u4w __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) double win[T + 2 * W];
    __shared__ __attribute__((aligned(16))) double vs[4][896 + 8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double d0 = 0.0;
    long s = order ? xstart[blockIdx.x & 7] + (blockIdx.x >> 3) : blockIdx.x;
    const long ntile = (r_end - r_begin) / T;
    const long send = order ? xstart[(blockIdx.x & 7) + 1] : ntile, sstep = order ? gridDim.x >> 3 : gridDim.x;
    for (; s < send; s += sstep) {
        const long t = order ? order[s] : s;
        const long ts = r_begin + t * T;
        u4w wreg[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) wreg[i] = *reinterpret_cast<const u4w *>(x + ts - W + 2 * (long)(tid + i * BLOCK));
        D2 fm[NQ], fp[NQ], uu[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const long r0 = ts + ((q * 4 + wv) << 7) + 2 * lane;
            fm[q] = ldg2(x + r0 - P); fp[q] = ldg2(x + r0 + P);
            if (DOT == 1) uu[q] = ldg2(u + r0);
        }
        u4w vreg[7];
        auto load_vals = [&](int q) {
            const double *vb = val + (ts - r_begin + ((q * 4 + wv) << 7)) * 7;            // 896 doubles, 16-byte aligned (block starts are even)
#pragma unroll
            for (int i = 0; i < 7; ++i) vreg[i] = *reinterpret_cast<const u4w *>(vb + 2 * (lane + i * 64));
        };
        load_vals(0);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NW; ++i) *reinterpret_cast<u4w *>(&win[2 * (tid + i * BLOCK)]) = wreg[i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int i = 0; i < 7; ++i) *reinterpret_cast<u4w *>(&vs[wv][2 * (lane + i * 64)]) = vreg[i];
            if (q + 1 < NQ) load_vals(q + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int li = W + ((q * 4 + wv) << 7) + 2 * lane;
            const double *v0 = &vs[wv][14 * lane], *v1 = v0 + 7;
            const D2 cc = *reinterpret_cast<const D2 *>(&win[li]);
            const D2 a = *reinterpret_cast<const D2 *>(&win[li - nx]);
            const D2 b = *reinterpret_cast<const D2 *>(&win[li + nx]);
            const double xl = win[li - 1], xr = win[li + 2];
            D2 o;
            double acc = 0.0;
            acc = acc + fm[q].lo * v0[0]; acc = acc + a.lo * v0[1]; acc = acc + xl * v0[2]; acc = acc + cc.lo * v0[3];
            acc = acc + cc.hi * v0[4]; acc = acc + b.lo * v0[5]; acc = acc + fp[q].lo * v0[6];
            o.lo = acc; acc = 0.0;
            acc = acc + fm[q].hi * v1[0]; acc = acc + a.hi * v1[1]; acc = acc + cc.lo * v1[2]; acc = acc + cc.hi * v1[3];
            acc = acc + xr * v1[4]; acc = acc + b.hi * v1[5]; acc = acc + fp[q].hi * v1[6];
            o.hi = acc;
            stnt2(y + ts + ((q * 4 + wv) << 7) + 2 * lane, o);
            if (DOT == 1) { d0 = d0 + o.lo * uu[q].lo; d0 = d0 + o.hi * uu[q].hi; }
            if (DOT == 2) { d0 = d0 + o.lo * cc.lo; d0 = d0 + o.hi * cc.hi; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if (DOT) {
        for (int o = 32; o > 0; o >>= 1) d0 += __shfl_xor(d0, o, 64);
        if (lane == 0) part[blockIdx.x * 4 + wv] = d0;
    }
}

int main(int argc, char **argv) {
    const std::string filt = argc > 1 ? argv[1] : "";
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nx = 500, ny = 500, nz = argc > 3 ? atoi(argv[3]) : 200;      // (nz = 25: the per-rank slab of N = 8)
    const long P = (long)nx * ny, n = P * nz;
    const long LCM = 8192;                                   // every variant's tile divides the timed range
    const long r_begin = P, r_end = r_begin + (n - 2 * P) / LCM * LCM;
    double *x, *y, *yr, *u, *w, *part; unsigned long long *bad;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&yr, n * 8)); CK(hipMalloc(&u, n * 8)); CK(hipMalloc(&w, n * 8)); CK(hipMemset(w, 0, n * 8)); CK(hipMalloc(&part, 1 << 20)); CK(hipMalloc(&bad, 8));
    fill_vec<<<2048, 256>>>(n, x, 1); fill_vec<<<2048, 256>>>(n, u, 7);
    CK(hipMemset(y, 0, n * 8)); CK(hipMemset(yr, 0, n * 8));
    Coef c{{-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0}};
    ref_kernel<<<4096, 256>>>(r_begin, r_end, nx, P, c, x, yr);
    CK(hipDeviceSynchronize());
    const double rows = (double)(r_end - r_begin);
    printf("rows timed %.0f of %ld; compulsory bytes per launch: DOT0/2 %.3f GB, DOT1 %.3f GB\n", rows, n, rows * 16 / 1e9, rows * 24 / 1e9);
    printf("%-28s %10s %10s %10s  %s\n", "variant", "b2b us", "altern us", "GB/s(b2b)", "check");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const std::string &name, int dot, std::function<void()> launch) {
        if (!filt.empty() && name.find(filt) == std::string::npos) return;
        CK(hipMemset(y, 0xff, n * 8));
        launch(); CK(hipDeviceSynchronize());
        CK(hipMemset(bad, 0, 8));
        cmp_kernel<<<2048, 256>>>(r_begin, r_end, y, yr, bad);
        unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double b2b = ms * 1e3 / reps;
        double alt = 0;                                       // alternating with the streaming neighbour: time the SpMV launches alone
        for (int i = 0; i < reps; ++i) {
            triad_kernel<<<512, 256>>>(n, 0.5, u, w);                   // 2R + 1W over 400 MB vectors, like K1 / K3 before an SpMV
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); alt += ms * 1e3;
        }
        alt /= reps;
        const double bytes = rows * (dot == 1 ? 24 : 16);
        printf("%-28s %10.1f %10.1f %10.0f  %s\n", name.c_str(), b2b, alt, bytes / b2b / 1e3, hb ? "MISMATCH" : "bit-exact");
        fflush(stdout);
    };
    (void)run;
    double *val = nullptr, *yrv = nullptr;
    const bool want_v = filt.empty() || filt.find("winv") != std::string::npos;
    auto make_order = [&](int T, int **d_order, int **d_xstart) {
        const long ntile = (r_end - r_begin) / T;
        std::vector<int> ord; std::vector<int> xs(9, 0);
        for (int xc = 0; xc < 8; ++xc) {
            xs[xc] = (int)ord.size();
            for (long t = 0; t < ntile; ++t) if ((int)(((t * T) % P) * 8 / P) == xc) ord.push_back((int)t);
        }
        xs[8] = (int)ord.size();
        CK(hipMalloc(d_order, ord.size() * 4)); CK(hipMemcpy(*d_order, ord.data(), ord.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(d_xstart, 36)); CK(hipMemcpy(*d_xstart, xs.data(), 36, hipMemcpyHostToDevice));
    };
    int *o1024, *x1024, *o2048, *x2048, *o4096, *x4096;
    make_order(1024, &o1024, &x1024); make_order(2048, &o2048, &x2048); make_order(4096, &o4096, &x4096);
    for (int grid : {256, 512, 768, 1024, 1536, 2048}) {
        const std::string g = "/" + std::to_string(grid);
        run("gl5/dot0" + g, 0, [&] { k_gl5<0><<<grid, BLOCK>>>(r_begin, r_end, nx, P, c, x, y, u, part); });
        run("gl5/dot1" + g, 1, [&] { k_gl5<1><<<grid, BLOCK>>>(r_begin, r_end, nx, P, c, x, y, u, part); });
        run("gl5/dot2" + g, 2, [&] { k_gl5<2><<<grid, BLOCK>>>(r_begin, r_end, nx, P, c, x, y, u, part); });
#define WIN(T, W, PRE, SFX) \
        run("win" #T "w" #W SFX "/dot0" + g, 0, [&] { k_win<T, W, 0, PRE><<<grid, BLOCK>>>(r_begin, r_end, nx, P, n, c, x, y, u, part, nullptr, nullptr); }); \
        run("win" #T "w" #W SFX "/dot1" + g, 1, [&] { k_win<T, W, 1, PRE><<<grid, BLOCK>>>(r_begin, r_end, nx, P, n, c, x, y, u, part, nullptr, nullptr); }); \
        run("win" #T "w" #W SFX "/dot2" + g, 2, [&] { k_win<T, W, 2, PRE><<<grid, BLOCK>>>(r_begin, r_end, nx, P, n, c, x, y, u, part, nullptr, nullptr); }); \
        run("win" #T "w" #W SFX "-period/dot0" + g, 0, [&] { k_win<T, W, 0, PRE><<<grid, BLOCK>>>(r_begin, r_end, nx, P, n, c, x, y, u, part, o##T, x##T); }); \
        run("win" #T "w" #W SFX "-period/dot1" + g, 1, [&] { k_win<T, W, 1, PRE><<<grid, BLOCK>>>(r_begin, r_end, nx, P, n, c, x, y, u, part, o##T, x##T); }); \
        run("win" #T "w" #W SFX "-period/dot2" + g, 2, [&] { k_win<T, W, 2, PRE><<<grid, BLOCK>>>(r_begin, r_end, nx, P, n, c, x, y, u, part, o##T, x##T); });
        WIN(1024, 512, 0, "") WIN(2048, 512, 0, "") WIN(4096, 512, 0, "")
        WIN(1024, 512, 1, "pre") WIN(2048, 512, 1, "pre")
    }
    if (want_v) {
        const long nv = (r_end - r_begin) * 7;
        CK(hipMalloc(&val, nv * 8)); CK(hipMalloc(&yrv, n * 8)); CK(hipMemset(yrv, 0, n * 8));
        fill_vals<<<4096, 256>>>(nv, val);
        refv_kernel<<<4096, 256>>>(r_begin, r_end, nx, P, val, x, yrv);
        CK(hipDeviceSynchronize());
        std::swap(yr, yrv);                                  // run() compares with yr
        const double vbytes = rows * 56;
        printf("variable coefficients: + %.3f GB of values per launch\n", vbytes / 1e9);
        for (int grid : {512, 768, 1024}) {
            const std::string g = "/" + std::to_string(grid);
            for (int per = 0; per < 2; ++per) {
                int *o = per ? o4096 : nullptr, *xs = per ? x4096 : nullptr;
                const std::string w = per ? "winv4096w512-period" : "winv4096w512";
                run(w + "/dot0" + g, 0, [&] { k_winv<4096, 512, 0><<<grid, BLOCK>>>(r_begin, r_end, nx, P, n, val, x, y, u, part, o, xs); });
                run(w + "/dot1" + g, 1, [&] { k_winv<4096, 512, 1><<<grid, BLOCK>>>(r_begin, r_end, nx, P, n, val, x, y, u, part, o, xs); });
                run(w + "/dot2" + g, 2, [&] { k_winv<4096, 512, 2><<<grid, BLOCK>>>(r_begin, r_end, nx, P, n, val, x, y, u, part, o, xs); });
            }
        }
        std::swap(yr, yrv);
    }
    return 0;
}
