// Round-3 micro-benchmark: does the speed of a multi-stream vector pass depend on WHERE its arrays start relative to each
// other?  k arrays of n doubles are carved from one allocation at base + i * (n * 8 rounded up to 2 MiB) + i * shift, and a
// pass reads R of them and writes W with 16-byte non-temporal accesses at the same index (the fused recurrence kernels' shape:
// K3 = 2R + 1W, K1 = 3R + 1W, K5 = 5R + 2W).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/stagger.hip -o /tmp/stagger && /tmp/stagger
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
struct alignas(16) D2 { double a, b; };
struct Ptrs { const double *r[5]; double *w[2]; };

template <int R, int W>
__global__ __launch_bounds__(256) void pass(long n2, Ptrs p, double c) {
    for (long g = blockIdx.x * 256L + threadIdx.x; g < n2; g += (long)gridDim.x * 256L) {
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const u4 q = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(p.r[i]) + g);
            D2 d; __builtin_memcpy(&d, &q, 16);
            s0 += c * d.a; s1 += c * d.b;
        }
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const D2 d{s0 + i, s1 + i};
            u4 q; __builtin_memcpy(&q, &d, 16);
            __builtin_nontemporal_store(q, reinterpret_cast<u4 *>(p.w[i]) + g);
        }
    }
}

int main() {
    const long n = 50000000, bytes = n * 8, slot = (bytes + (2L << 20) - 1) / (2L << 20) * (2L << 20);
    const long maxshift = 8L << 20;
    char *base; CK(hipMalloc(&base, 7 * (slot + maxshift) + (64L << 20)));
    CK(hipMemset(base, 0, 7 * (slot + maxshift)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long shifts[] = {0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1L << 20, (1L << 20) + 4096, 400000000L - slot};
    printf("%-12s %12s %12s %12s   (us per pass; GB/s)\n", "shift", "2R+1W", "3R+1W", "5R+2W");
    for (long sh : shifts) {
        Ptrs p;
        for (int i = 0; i < 5; ++i) p.r[i] = reinterpret_cast<const double *>(base + (32L << 20) + i * slot + i * sh);
        for (int i = 0; i < 2; ++i) p.w[i] = reinterpret_cast<double *>(base + (32L << 20) + (5 + i) * slot + (5 + i) * sh);
        double t[3];
        for (int k = 0; k < 3; ++k) {
            auto launch = [&]() {
                if (k == 0) pass<2, 1><<<512, 256>>>(n / 2, p, 0.5);
                if (k == 1) pass<3, 1><<<512, 256>>>(n / 2, p, 0.5);
                if (k == 2) pass<5, 2><<<512, 256>>>(n / 2, p, 0.5);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t[k] = ms * 1e3 / 20;
        }
        printf("%-12ld %7.1f %5.0f %7.1f %5.0f %7.1f %5.0f\n", sh, t[0], 3 * bytes / t[0] / 1e3, t[1], 4 * bytes / t[1] / 1e3, t[2], 7 * bytes / t[2] / 1e3);
        fflush(stdout);
    }
    return 0;
}
