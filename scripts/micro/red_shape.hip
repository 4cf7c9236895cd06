// Which launch shape streams a read-only reduction (nrm2: one vector; dot: two) fastest at HBM size?  n = 50 M doubles
// (cfg 5).  Not part of the product; the winner's shape goes into csrc/blas1.hip (nrm2sq_kernel / dot_kernel).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/micro/red_shape.hip -o scripts/micro/red_shape
// Variants: U = 16-byte packs in flight per lane per trip (grid-stride between them, as in the product: the additions keep
// the order i, i+st, i+2st, ...), NT = non-temporal loads, grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ double2 ld(const double2 *p) {
    if constexpr (NT) { const u4 q = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(p)); double2 r; __builtin_memcpy(&r, &q, 16); return r; }
    else return *p;
}
__device__ __forceinline__ double block_sum(double v, double *sm) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}
template <int U, bool NT>
__global__ __launch_bounds__(BLOCK) void nrm2(long np, const double2 *__restrict__ x, double *part) {
    __shared__ double sm[4];
    double acc = 0;
    const long st = (long)gridDim.x * BLOCK;
    long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    for (; i + (U - 1) * st < np; i += U * st) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(x + i + u * st);
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += v[u].x * v[u].x; acc += v[u].y * v[u].y; }
    }
    for (; i < np; i += st) { const double2 v = ld<NT>(x + i); acc += v.x * v.x; acc += v.y * v.y; }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
template <int U, bool NT>
__global__ __launch_bounds__(BLOCK) void dot(long np, const double2 *__restrict__ x, const double2 *__restrict__ y, double *part) {
    __shared__ double sm[4];
    double acc = 0;
    const long st = (long)gridDim.x * BLOCK;
    long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    for (; i + (U - 1) * st < np; i += U * st) {
        double2 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = ld<NT>(x + i + u * st); b[u] = ld<NT>(y + i + u * st); }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += a[u].x * b[u].x; acc += a[u].y * b[u].y; }
    }
    for (; i < np; i += st) { const double2 a = ld<NT>(x + i), b = ld<NT>(y + i); acc += a.x * b.x; acc += a.y * b.y; }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 50000000, np = n / 2;
    double *x, *y, *z, *part;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&z, n * 8)); CK(hipMalloc(&part, 8192 * 8));
    CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8)); CK(hipMemset(z, 0, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // between two timed launches another 0.4 GB vector is read, so that nothing of x / y survives in the Infinity Cache
    auto timeit = [&](const char *name, int grid, auto launch, double bytes) {
        float tot = 0;
        for (int i = 0; i < 7; ++i) {
            nrm2<4, false><<<1024, BLOCK>>>(np, (const double2 *)z, part);
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (i >= 2) tot += ms;
        }
        tot /= 5;
        printf("%-28s grid %5d : %7.1f us %6.0f GB/s  %.3f of 8 TB/s\n", name, grid, tot * 1e3, bytes / tot / 1e6, bytes / tot / 1e6 / 8000); fflush(stdout);
    };
#define N2(U, NT, G) timeit("nrm2 U=" #U " nt=" #NT, G, [&]() { nrm2<U, NT><<<G, BLOCK>>>(np, (const double2 *)x, part); }, 1.0 * n * 8)
#define DT(U, NT, G) timeit("dot  U=" #U " nt=" #NT, G, [&]() { dot<U, NT><<<G, BLOCK>>>(np, (const double2 *)x, (const double2 *)y, part); }, 2.0 * n * 8)
    for (int G : {512, 1024, 2048, 4096}) {
        N2(4, false, G); N2(4, true, G); N2(8, false, G); N2(8, true, G); N2(16, true, G);
        DT(2, false, G); DT(2, true, G); DT(4, false, G); DT(4, true, G); DT(8, true, G);
    }
    return 0;
}
