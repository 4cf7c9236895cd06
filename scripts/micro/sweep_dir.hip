// Does the direction of a streaming pass matter to the pass that follows it?  (not part of the product)
// The 256 MiB Infinity Cache is memory-side: after a front-to-back pass over vectors far larger than it, the TAIL of
// what was touched is resident.  A next pass that also runs front-to-back evicts that tail before reaching it; one that
// runs back-to-front starts on it.  Chain of dependent passes over a pool of 8 vectors of n doubles, as a Krylov
// iteration has them: pass k writes v[k % 8] (and v[(k + 4) % 8] when NW = 2) and reads the NR vectors written most
// recently before it.  Timed per pass: all passes front-to-back vs alternating directions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256;
struct Ptrs { double2 *r[5]; double2 *w[2]; };
template <int NR, int NW>
__global__ __launch_bounds__(BLOCK) void pass(Ptrs p, double s, long ntile, int backwards) {
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long tt = backwards ? ntile - 1 - t : t;
        const long i = tt * BLOCK + threadIdx.x;
        double2 x[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) x[j] = p.r[j][i];
        double2 acc = x[0];
#pragma unroll
        for (int j = 1; j < NR; ++j) { acc.x += s * x[j].x; acc.y += s * x[j].y; }
        p.w[0][i] = acc;
        if (NW == 2) p.w[1][i] = double2{acc.y, acc.x};
    }
}
template <int NR, int NW>
void run(double2 **v, long n, int grid) {
    const long ntile = n / 2 / BLOCK;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 48;
    double us[3];
    for (int mode = 0; mode < 3; ++mode) {          // 0: all forward, 1: alternate, 2: all forward again (drift check)
        for (int r = -8; r < reps; ++r) {
            if (r == 0) CK(hipEventRecord(e0));
            const int k = r + 8;
            Ptrs p{};
            p.w[0] = v[k % 8]; p.w[1] = v[(k + 4) % 8];
            int got = 0;
            for (int back = 1; got < NR; ++back) {                  // the most recently written vectors not being written now
                const int idx = ((k - back) % 8 + 8) % 8;
                if (idx == k % 8 || (NW == 2 && idx == (k + 4) % 8)) continue;
                p.r[got++] = v[idx];
            }
            hipLaunchKernelGGL((pass<NR, NW>), dim3(grid), dim3(BLOCK), 0, 0, p, 0.5, ntile, mode == 1 ? (k & 1) : 0);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us[mode] = ms * 1e3 / reps;
    }
    printf("n=%ld grid=%4d %dR+%dW  same %7.1f / %7.1f us  alternating %7.1f us  (%+5.1f %%)  %6.0f -> %6.0f GB/s\n", n, grid, NR, NW,
           us[0], us[2], us[1], 100.0 * (us[1] / (0.5 * (us[0] + us[2])) - 1.0), (NR + NW) * n * 8.0 / (0.5 * (us[0] + us[2])) / 1e3,
           (NR + NW) * n * 8.0 / us[1] / 1e3);
}
int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 50000000L;
    double2 *v[8];
    for (int j = 0; j < 8; ++j) { CK(hipMalloc(&v[j], n * 8)); CK(hipMemset(v[j], 0, n * 8)); }
    for (int grid : {512, 1024, 2048}) {
        run<2, 1>(v, n, grid);
        run<3, 1>(v, n, grid);
        run<5, 2>(v, n, grid);
    }
    return 0;
}
