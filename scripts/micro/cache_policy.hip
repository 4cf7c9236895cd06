// Which cache-policy bits should a pure streaming pass carry on gfx950?  (not part of the product)
// K5-shaped pass (5 reads + 2 writes of 16 B per lane) over 0.4 GB vectors; loads / stores issued with the given
// modifier strings through inline assembly (one s_waitcnt vmcnt(0) before the arithmetic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256;
typedef double d2 __attribute__((ext_vector_type(2)));
struct Ptrs { d2 *r[5]; d2 *w[2]; };
#define LD(MOD) template <> __device__ __forceinline__ d2 ld<__COUNTER__ - CB>(const d2 *p) { d2 v; asm volatile("global_load_dwordx4 %0, %1, off " MOD : "=v"(v) : "v"(p) : "memory"); return v; }
#define ST(MOD) template <> __device__ __forceinline__ void st<__COUNTER__ - CS>(d2 *p, d2 v) { asm volatile("global_store_dwordx4 %0, %1, off " MOD :: "v"(p), "v"(v) : "memory"); }
template <int V> __device__ __forceinline__ d2 ld(const d2 *p);
template <int V> __device__ __forceinline__ void st(d2 *p, d2 v);
constexpr int CB = __COUNTER__ + 1;
LD("") LD("nt") LD("sc0") LD("sc1") LD("sc0 sc1") LD("sc0 sc1 nt") LD("sc1 nt") LD("sc0 nt")
constexpr int NLD = 8;
constexpr int CS = __COUNTER__ + 1;
ST("") ST("nt") ST("sc0 sc1") ST("sc0 sc1 nt") ST("sc1 nt") ST("sc0 nt")
constexpr int NST = 6;
const char *LDN[NLD] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc1 nt", "sc0 nt"};
const char *STN[NST] = {"plain", "nt", "sc0 sc1", "sc0 sc1 nt", "sc1 nt", "sc0 nt"};

template <int LV, int SV>
__global__ __launch_bounds__(BLOCK) void pass(Ptrs p, double s, long ntile) {
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long i = t * BLOCK + threadIdx.x;
        d2 x[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) x[j] = ld<LV>(p.r[j] + i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        d2 acc = x[0];
#pragma unroll
        for (int j = 1; j < 5; ++j) acc += s * x[j];
        st<SV>(p.w[0] + i, acc);
        st<SV>(p.w[1] + i, acc * s);
    }
}
template <int LV, int SV>
double run(d2 **v, long n, int grid) {
    const long ntile = n / 2 / BLOCK;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 24;
    for (int r = -4; r < reps; ++r) {
        if (r == 0) CK(hipEventRecord(e0));
        const int k = r + 8;
        Ptrs p{};
        p.w[0] = v[k % 8]; p.w[1] = v[(k + 4) % 8];
        int got = 0;
        for (int back = 1; got < 5; ++back) {
            const int idx = ((k - back) % 8 + 8) % 8;
            if (idx == k % 8 || idx == (k + 4) % 8) continue;
            p.r[got++] = v[idx];
        }
        hipLaunchKernelGGL((pass<LV, SV>), dim3(grid), dim3(BLOCK), 0, 0, p, 0.5, ntile);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}
template <int LV, int SV> void one(d2 **v, long n, int grid) {
    const double us = run<LV, SV>(v, n, grid);
    printf("loads %-12s stores %-12s %7.1f us  %6.0f GB/s\n", LDN[LV], STN[SV], us, 7.0 * n * 8 / us / 1e3);
}
int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 50000000L;
    const int grid = argc > 2 ? atoi(argv[2]) : 512;
    d2 *v[8];
    for (int j = 0; j < 8; ++j) { CK(hipMalloc(&v[j], n * 8)); CK(hipMemset(v[j], 0, n * 8)); }
    one<0, 0>(v, n, grid); one<1, 0>(v, n, grid); one<1, 1>(v, n, grid); one<0, 1>(v, n, grid);
    one<2, 1>(v, n, grid); one<3, 1>(v, n, grid); one<4, 1>(v, n, grid); one<5, 1>(v, n, grid); one<6, 1>(v, n, grid); one<7, 1>(v, n, grid);
    one<1, 2>(v, n, grid); one<1, 3>(v, n, grid); one<1, 4>(v, n, grid); one<1, 5>(v, n, grid);
    one<5, 3>(v, n, grid); one<6, 4>(v, n, grid);
    one<0, 0>(v, n, grid); one<1, 1>(v, n, grid);
    return 0;
}
