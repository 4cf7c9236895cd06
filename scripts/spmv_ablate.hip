// Ablation probe for the SpMV stream path (not part of the product).  Builds the cfg-5 7-point
// matrix in HBM and times variants of the kernel body to see which phase bounds it:
//   0 full            1 no gather (x = 1)      2 no LDS/barriers (per-thread partial sums, wrong y)
//   3 stream only (loads of col/val, no gather, no LDS)       4 full, 512 rows / block (ITEMS 16)
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/spmv_ablate.hip -o /tmp/spmv_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256;
struct Desc { int ra, rb, pa, nn; };

__global__ void gen_rowptr(int nx, int ny, int nz, int *cnt) {
    long n = (long)nx * ny * nz;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n; g += (long)gridDim.x * blockDim.x) {
        int x = g % nx, y = (g / nx) % ny, z = g / ((long)nx * ny);
        cnt[g] = 1 + (x > 0) + (x < nx - 1) + (y > 0) + (y < ny - 1) + (z > 0) + (z < nz - 1);
    }
}
__global__ void gen_fill(int nx, int ny, int nz, const int *rp, int *ci, double *val) {
    long n = (long)nx * ny * nz, P = (long)nx * ny;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n; g += (long)gridDim.x * blockDim.x) {
        int x = g % nx, y = (g / nx) % ny, z = g / P;
        int p = rp[g];
        if (z > 0) { ci[p] = g - P; val[p++] = -1; }
        if (y > 0) { ci[p] = g - nx; val[p++] = -1; }
        if (x > 0) { ci[p] = g - 1; val[p++] = -1; }
        ci[p] = g; val[p++] = 6;
        if (x < nx - 1) { ci[p] = g + 1; val[p++] = -1; }
        if (y < ny - 1) { ci[p] = g + nx; val[p++] = -1; }
        if (z < nz - 1) { ci[p] = g + P; val[p++] = -1; }
    }
}

template <int MODE, int ROWS>
__global__ __launch_bounds__(BLOCK) void spmv(int nblk, const Desc *__restrict__ desc, const int *__restrict__ row_ptr,
                                              const int *__restrict__ col_idx, const double *__restrict__ val,
                                              const double *__restrict__ x, double *__restrict__ y) {
    constexpr int CAP = ROWS * 8;
    constexpr int ITEMS = CAP / BLOCK;
    __shared__ double prod[CAP];
    const int tid = threadIdx.x;
    double sink = 0;
    for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
        const Desc d = desc[b];
        const int pa = d.pa, nn = d.nn, last = nn - 1;
        int cidx[ITEMS]; double vv[ITEMS], xg[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { int k = min(tid + i * BLOCK, last); cidx[i] = col_idx[pa + k]; vv[i] = val[pa + k]; }
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) xg[i] = 1.0 + cidx[i] * 1e-30;
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) xg[i] = x[cidx[i]];
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) sink += xg[i] * vv[i];
            continue;
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { int k = tid + i * BLOCK; if (k < nn) prod[k] = xg[i] * vv[i]; }
        __syncthreads();
        for (int rr = tid; rr < d.rb - d.ra; rr += BLOCK) {
            const int r = d.ra + rr;
            const int s = row_ptr[r] - pa, len = row_ptr[r + 1] - pa - s;
            double pv[8], acc = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = prod[min(s + j, CAP - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < len) acc += pv[j];
            y[r] = acc;
        }
        __syncthreads();
    }
    if (MODE == 2 || MODE == 3) if (sink == 123.456) y[0] = sink;
}


__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// V: 0 = product structure with __syncthreads, 1 = lds_barrier, 2 = lds_barrier + next-block stream prefetch
template <int V>
__global__ __launch_bounds__(BLOCK) void spmv2(int nblk, const Desc *__restrict__ desc, const int *__restrict__ row_ptr,
                                               const int *__restrict__ col_idx, const double *__restrict__ val,
                                               const double *__restrict__ x, double *__restrict__ y) {
    constexpr int CAP = 2048, ITEMS = 8;
    __shared__ double prod[CAP];
    const int tid = threadIdx.x;
    int cN[ITEMS]; double vN[ITEMS];
    auto sload = [&](const Desc &d) {
        const int last = max(d.nn - 1, 0);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { int k = min(tid + i * BLOCK, last); cN[i] = col_idx[d.pa + k]; vN[i] = val[d.pa + k]; }
    };
    int b = blockIdx.x;
    bool have = b < nblk;
    Desc dn = {0, 0, 0, 0};
    if (have) { dn = desc[b]; if (V == 2) sload(dn); }
    while (have) {
        const Desc d = dn;
        b += gridDim.x; have = b < nblk;
        if (have) dn = desc[b];
        const int pa = d.pa, nn = d.nn;
        const int r = d.ra + tid; const bool has_row = r < d.rb; const int rcl = has_row ? r : d.rb - 1;
        const int s = row_ptr[rcl] - pa, e = row_ptr[rcl + 1] - pa;
        int cidx[ITEMS]; double vv[ITEMS], xg[ITEMS];
        if (V == 2) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) { cidx[i] = cN[i]; vv[i] = vN[i]; }
        } else {
            const int last = nn - 1;
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) { int k = min(tid + i * BLOCK, last); cidx[i] = col_idx[pa + k]; vv[i] = val[pa + k]; }
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) xg[i] = x[cidx[i]];
        if (V == 2 && have) sload(dn);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { int k = tid + i * BLOCK; if (k < nn) prod[k] = xg[i] * vv[i]; }
        if (V == 0) __syncthreads(); else lds_barrier();
        if (has_row) {
            const int len = e - s; double pv[8], acc = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = prod[min(s + j, CAP - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < len) acc += pv[j];
            y[r] = acc;
        }
        if (V == 0) __syncthreads(); else lds_barrier();
    }
}


// wave-independent variant: each wavefront owns 64 rows (<= 512 nnz) and a private LDS slice; no s_barrier
template <int WITEMS, int NG = 1>
__global__ __launch_bounds__(BLOCK) void spmv3(int ndesc, const Desc *__restrict__ desc, const int *__restrict__ row_ptr,
                                               const int *__restrict__ col_idx, const double *__restrict__ val,
                                               const double *__restrict__ x, double *__restrict__ y,
                                               const double *__restrict__ x2 = nullptr, const double *__restrict__ x3 = nullptr,
                                               double a = 0.5, double bb = 0.25, double *__restrict__ y2 = nullptr) {
    constexpr int WCAP = 64 * WITEMS;
    __shared__ double prod_all[4][WCAP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all[w];
    const int nw = gridDim.x * 4;
    for (int b = blockIdx.x * 4 + w; b < ndesc; b += nw) {
        const Desc d = desc[b];
        const int pa = d.pa, nn = d.nn, last = nn - 1;
        const int r = d.ra + lane; const bool has_row = r < d.rb; const int rcl = has_row ? r : d.rb - 1;
        const int s = row_ptr[rcl] - pa, e = row_ptr[rcl + 1] - pa;
        int cidx[WITEMS]; double vv[WITEMS], xg[WITEMS];
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) { int k = min(lane + i * 64, last); cidx[i] = col_idx[pa + k]; vv[i] = val[pa + k]; }
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) xg[i] = x[cidx[i]];
        if (NG >= 2) {
            double g2[WITEMS];
#pragma unroll
            for (int i = 0; i < WITEMS; ++i) g2[i] = x2[cidx[i]];
            if (NG >= 3) {
                double g3[WITEMS];
#pragma unroll
                for (int i = 0; i < WITEMS; ++i) g3[i] = x3[cidx[i]];
#pragma unroll
                for (int i = 0; i < WITEMS; ++i) xg[i] = (g2[i] * a + g3[i] * bb) + xg[i];
            } else {
#pragma unroll
                for (int i = 0; i < WITEMS; ++i) xg[i] = xg[i] + g2[i] * a;
            }
        }
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) { int k = lane + i * 64; if (k < nn) prod[k] = xg[i] * vv[i]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        if (has_row) {
            const int len = e - s; double pv[8], acc = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = prod[min(s + j, WCAP - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < len) acc += pv[j];
            y[r] = acc;
            if (NG >= 2) y2[r] = x[r] + x2[r] * a + (NG >= 3 ? x3[r] * bb : 0.0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
}


// spmv4: stage (col, val) through LDS, then lane = row: the x gathers of 64 consecutive rows at the same
// in-row position are (for stencil / banded matrices) 64 consecutive addresses -> coalesced in the TA
template <int WITEMS>
__global__ __launch_bounds__(BLOCK) void spmv4(int ndesc, const Desc *__restrict__ desc, const int *__restrict__ row_ptr,
                                               const int *__restrict__ col_idx, const double *__restrict__ val,
                                               const double *__restrict__ x, double *__restrict__ y) {
    constexpr int WCAP = 64 * WITEMS;
    __shared__ double lv_all[4][WCAP];
    __shared__ int lc_all[4][WCAP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *lv = lv_all[w]; int *lc = lc_all[w];
    const int nw = gridDim.x * 4;
    for (int b = blockIdx.x * 4 + w; b < ndesc; b += nw) {
        const Desc d = desc[b];
        const int pa = d.pa, nn = d.nn, last = nn - 1;
        const int r = d.ra + lane; const bool has_row = r < d.rb; const int rcl = has_row ? r : d.rb - 1;
        const int s = row_ptr[rcl] - pa, e = row_ptr[rcl + 1] - pa;
        int cidx[WITEMS]; double vv[WITEMS];
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) { int k = min(lane + i * 64, last); cidx[i] = col_idx[pa + k]; vv[i] = val[pa + k]; }
#pragma unroll
        for (int i = 0; i < WITEMS; ++i) { int k = lane + i * 64; if (k < nn) { lc[k] = cidx[i]; lv[k] = vv[i]; } }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        const int len = e - s;
        if (has_row && len > 0) {
            int cc[8]; double v8[8], x8[8]; double acc = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { int idx = min(s + j, e - 1); cc[j] = lc[idx]; v8[j] = lv[idx]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) x8[j] = x[cc[j]];
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < len) acc += x8[j] * v8[j];
            y[r] = acc;
        } else if (has_row) y[r] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
}

int main(int argc, char **argv) {
    const int nx = 500, ny = 500, nz = 200; const long n = (long)nx * ny * nz;
    int *cnt, *rp, *ci; double *val, *x, *y;
    CK(hipMalloc(&cnt, n * 4)); CK(hipMalloc(&rp, (n + 1) * 4));
    gen_rowptr<<<4096, 256>>>(nx, ny, nz, cnt);
    std::vector<int> hc(n), hrp(n + 1);
    CK(hipMemcpy(hc.data(), cnt, n * 4, hipMemcpyDeviceToHost));
    hrp[0] = 0; for (long i = 0; i < n; ++i) hrp[i + 1] = hrp[i] + hc[i];
    const long nnz = hrp[n];
    CK(hipMemcpy(rp, hrp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ci, nnz * 4)); CK(hipMalloc(&val, nnz * 8)); CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8));
    gen_fill<<<4096, 256>>>(nx, ny, nz, rp, ci, val);
    CK(hipMemset(x, 0, n * 8));
    const double bytes = nnz * 12.0 + (n + 1) * 4.0 + 2.0 * n * 8;
    for (int rows : {256, 512}) {
        std::vector<Desc> hd;
        for (long r = 0; r < n; r += rows) { long e = std::min<long>(r + rows, n); hd.push_back({(int)r, (int)e, hrp[r], hrp[e] - hrp[r]}); }
        Desc *dd; CK(hipMalloc(&dd, hd.size() * sizeof(Desc))); CK(hipMemcpy(dd, hd.data(), hd.size() * sizeof(Desc), hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int grid : {1024, 2048}) for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&]() {
#define L(M, R) spmv<M, R><<<grid, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y)
                if (rows == 256) { if (mode == 0) L(0, 256); else if (mode == 1) L(1, 256); else if (mode == 2) L(2, 256); else L(3, 256); }
                else { if (mode == 0) L(0, 512); else if (mode == 1) L(1, 512); else if (mode == 2) L(2, 512); else L(3, 512); }
            };
            for (int w = 0; w < 3; ++w) launch();
            CK(hipEventRecord(e0)); for (int it = 0; it < 20; ++it) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
            printf("rows/blk %d grid %d mode %d : %8.1f us  %7.1f GB/s (algorithmic)\n", rows, grid, mode, ms * 1e3, bytes / ms / 1e6);
        }
        if (rows == 256) {
            for (int rnd = 0; rnd < 2; ++rnd) for (int grid : {1024, 1280, 2048}) for (int v = 0; v < 3; ++v) {
                auto launch = [&]() {
                    if (v == 0) spmv2<0><<<grid, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y);
                    else if (v == 1) spmv2<1><<<grid, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y);
                    else spmv2<2><<<grid, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y);
                };
                for (int w = 0; w < 3; ++w) launch();
                CK(hipEventRecord(e0)); for (int it = 0; it < 20; ++it) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
                printf("spmv2 variant %d grid %d : %8.1f us  %7.1f GB/s\n", v, grid, ms * 1e3, bytes / ms / 1e6);
            }
        }
        CK(hipFree(dd));
    }
    {   // wave-level descriptors: 64 rows each
        std::vector<Desc> hd;
        for (long r = 0; r < n; r += 64) { long e = std::min<long>(r + 64, n); hd.push_back({(int)r, (int)e, hrp[r], hrp[e] - hrp[r]}); }
        Desc *dd; CK(hipMalloc(&dd, hd.size() * sizeof(Desc))); CK(hipMemcpy(dd, hd.data(), hd.size() * sizeof(Desc), hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rnd = 0; rnd < 2; ++rnd) for (int grid : {1024, 1536, 2048}) {
            auto launch = [&]() { spmv3<8><<<grid, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y); };
            for (int w = 0; w < 3; ++w) launch();
            CK(hipEventRecord(e0)); for (int it = 0; it < 20; ++it) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
            printf("spmv3 (wave-independent) grid %d : %8.1f us  %7.1f GB/s\n", grid, ms * 1e3, bytes / ms / 1e6);
        }
        {
            double *xb, *xc, *yb; CK(hipMalloc(&xb, n * 8)); CK(hipMalloc(&xc, n * 8)); CK(hipMalloc(&yb, n * 8));
            CK(hipMemset(xb, 0, n * 8)); CK(hipMemset(xc, 0, n * 8));
            for (int rnd = 0; rnd < 2; ++rnd) for (int ng = 1; ng <= 3; ++ng) {
                auto launch = [&]() {
                    if (ng == 1) spmv3<8, 1><<<1024, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y);
                    else if (ng == 2) spmv3<8, 2><<<1024, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y, xb, xc, 0.5, 0.25, yb);
                    else spmv3<8, 3><<<1024, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y, xb, xc, 0.5, 0.25, yb);
                };
                for (int w = 0; w < 3; ++w) launch();
                CK(hipEventRecord(e0)); for (int it = 0; it < 20; ++it) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
                printf("spmv3 with %d gathered vectors (+1 extra output for >1): %8.1f us\n", ng, ms * 1e3);
            }
        }
        for (int rnd = 0; rnd < 2; ++rnd) for (int grid : {1024, 2048}) for (int var = 3; var <= 4; ++var) {
            auto launch = [&]() {
                if (var == 3) spmv3<8><<<grid, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y);
                else spmv4<8><<<grid, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y);
            };
            for (int w = 0; w < 3; ++w) launch();
            CK(hipEventRecord(e0)); for (int it = 0; it < 20; ++it) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
            printf("A/B spmv%d grid %d : %8.1f us  %7.1f GB/s\n", var, grid, ms * 1e3, bytes / ms / 1e6);
        }
        // correctness vs spmv2<1>
        std::vector<double> hx(n); for (long i = 0; i < n; ++i) hx[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
        CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice));
        double *y2; CK(hipMalloc(&y2, n * 8));
        spmv3<8><<<1024, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y2);
        double *y4; CK(hipMalloc(&y4, n * 8));
        spmv4<8><<<1024, BLOCK>>>((int)hd.size(), dd, rp, ci, val, x, y4);
        std::vector<Desc> hb;
        for (long r = 0; r < n; r += 256) { long e = std::min<long>(r + 256, n); hb.push_back({(int)r, (int)e, hrp[r], hrp[e] - hrp[r]}); }
        Desc *db; CK(hipMalloc(&db, hb.size() * sizeof(Desc))); CK(hipMemcpy(db, hb.data(), hb.size() * sizeof(Desc), hipMemcpyHostToDevice));
        spmv2<1><<<1024, BLOCK>>>((int)hb.size(), db, rp, ci, val, x, y);
        std::vector<double> a(n), b(n);
        CK(hipMemcpy(a.data(), y, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), y2, n * 8, hipMemcpyDeviceToHost));
        long bad = 0; for (long i = 0; i < n; ++i) bad += (a[i] != b[i]);
        printf("spmv3 vs spmv2 mismatches: %ld\n", bad);
        CK(hipMemcpy(b.data(), y4, n * 8, hipMemcpyDeviceToHost));
        bad = 0; for (long i = 0; i < n; ++i) bad += (a[i] != b[i]);
        printf("spmv4 vs spmv2 mismatches: %ld\n", bad);
    }
    return 0;
}
