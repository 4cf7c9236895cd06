// Which launch shape moves a K5-like pass (5 read streams, 2 of them written back in place: x, r) and a K1-like pass
// (3 reads, 1 written back) fastest?  n = 50 M doubles per vector (cfg 5).  (not part of the product)
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/micro/blas1_shape.hip -o scripts/micro/blas1_shape
// Variants: U = packs (16 B) per lane per trip; CONTIG = a workgroup's U packs are adjacent 4 KiB pieces (one 4U KiB
// run per stream) instead of grid-strided; NTL = non-temporal loads of the read-only operands; grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256;
template <bool NT> __device__ __forceinline__ double2 ld(const double2 *p) {
    if constexpr (NT) { double2 r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); return r; }
    else return *p;
}
template <int U, bool CONTIG, bool NTL>
__global__ __launch_bounds__(BLOCK) void k5(long np, double2 *__restrict__ x, const double2 *__restrict__ y, double2 *__restrict__ r,
                                            const double2 *__restrict__ t, const double2 *__restrict__ r0, double na, double nw, double *part) {
    double accN = 0, accR = 0;
    const long tile = (long)BLOCK * U, ntile = np / tile;          // np is a multiple of tile here
    for (long tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        long idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) idx[u] = CONTIG ? tl * tile + u * BLOCK + threadIdx.x : (tl + (long)u * 0) * tile + u * BLOCK + threadIdx.x;
        double2 xv[U], yv[U], rv[U], tv[U], qv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { xv[u] = x[idx[u]]; yv[u] = ld<NTL>(y + idx[u]); rv[u] = r[idx[u]]; tv[u] = ld<NTL>(t + idx[u]); qv[u] = ld<NTL>(r0 + idx[u]); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double2 xx, rr;
            xx.x = (xv[u].x + yv[u].x * na) + rv[u].x * nw; xx.y = (xv[u].y + yv[u].y * na) + rv[u].y * nw;
            rr.x = rv[u].x + tv[u].x * nw; rr.y = rv[u].y + tv[u].y * nw;
            accN += rr.x * rr.x + rr.y * rr.y; accR += qv[u].x * rr.x + qv[u].y * rr.y;
            x[idx[u]] = xx; r[idx[u]] = rr;
        }
    }
    if (accN == 1.2345e300) part[blockIdx.x] = accN + accR;
}
template <int U, bool NTL>
__global__ __launch_bounds__(BLOCK) void k1(long np, const double2 *__restrict__ v, double2 *__restrict__ p, const double2 *__restrict__ r, double a, double b) {
    const long tile = (long)BLOCK * U, ntile = np / tile;
    for (long tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        double2 vv[U], pv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = tl * tile + u * BLOCK + threadIdx.x; vv[u] = ld<NTL>(v + i); pv[u] = p[i]; rv[u] = ld<NTL>(r + i); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = tl * tile + u * BLOCK + threadIdx.x;
            double2 o; o.x = (vv[u].x * a + pv[u].x * b) + rv[u].x; o.y = (vv[u].y * a + pv[u].y * b) + rv[u].y;
            p[i] = o;
        }
    }
}
int main() {
    const long n = 50000000 / 2048 * 2048, np = n / 2;
    double *v[7], *part;
    for (auto &q : v) { CK(hipMalloc(&q, n * 8 + 4096)); CK(hipMemset(q, 0, n * 8)); }
    CK(hipMalloc(&part, 8192 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, int grid, auto launch, double bytes) {
        launch(); launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("%-40s grid %5d : %7.1f us %6.0f GB/s\n", name, grid, ms * 1e3, bytes / ms / 1e6); fflush(stdout);
    };
    auto D = [&](int i) { return (double2 *)v[i]; };
#define K5(U, NTL, G) timeit("K5 (5R+2W) U=" #U " nt=" #NTL, G, [&]() { k5<U, true, NTL><<<G, BLOCK>>>(np, D(0), D(1), D(2), D(3), D(4), -0.3, -0.2, part); }, 7.0 * n * 8)
#define K1(U, NTL, G) timeit("K1 (3R+1W) U=" #U " nt=" #NTL, G, [&]() { k1<U, NTL><<<G, BLOCK>>>(np, D(1), D(5), D(2), 0.3, 0.2); }, 4.0 * n * 8)
    for (int rep = 0; rep < 2; ++rep) for (int G : {256, 512, 768, 1024, 2048}) {
        K5(1, false, G); K5(2, false, G); K5(4, false, G); K5(1, true, G); K5(2, true, G);
        K1(1, false, G); K1(2, false, G); K1(4, false, G); K1(1, true, G); K1(2, true, G);
    }
    return 0;
}
