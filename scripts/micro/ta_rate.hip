// Micro-benchmark (not part of the product): how many cycles does a CU's vector-memory path need per 64-lane load
// instruction, by access width, when every access hits the L1?  Run: hipcc -O3 --offload-arch=gfx950 ta_rate.hip && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)

struct f2 { float x, y; };  struct d2 { double x, y; };
__device__ inline f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ inline d2 operator+(d2 a, d2 b) { return {a.x + b.x, a.y + b.y}; }


template <class V, int MODE>   // MODE 0: consecutive lanes consecutive elements; 1: all lanes the same element; 2: misaligned by one element; 3: lane stride 2 elements; 4: lane stride 2, odd elements
__global__ __launch_bounds__(256) void k(const V *__restrict__ buf, V *__restrict__ out, int iters, int mask_elems) {
    const int lane = threadIdx.x & 63;
    int base = (threadIdx.x >> 6) * 64 + (MODE == 2 ? 1 : 0);
    V acc{};
    for (int it = 0; it < iters; ++it) {
        V v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = (base + j * 512 + (MODE == 1 ? 0 : (MODE >= 3 ? 2 * lane + (MODE == 4) : lane))) & (mask_elems - 1);
            v[j] = buf[idx];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = acc + v[j];
        base += 64;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class V, int BYTES>
__global__ __launch_bounds__(256) void kmis(const char *__restrict__ buf, V *__restrict__ out, int iters, int mask_bytes) {
    const int lane = threadIdx.x & 63;
    int base = (threadIdx.x >> 6) * 64 * (int)sizeof(V) + BYTES;
    V acc{};
    for (int it = 0; it < iters; ++it) {
        V v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int off = (base + j * 4096 + lane * (int)sizeof(V)) & (mask_bytes - 1);
            v[j] = *reinterpret_cast<const V *>(buf + off);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = acc + v[j];
        base += 64 * (int)sizeof(V);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <class V, int BYTES>
void runmis(const char *name, int wg_per_cu) {
    const int bytes = 16384;
    char *buf; V *out;
    CK(hipMalloc(&buf, bytes + 4096)); CK(hipMemset(buf, 0, bytes + 4096));
    const int grid = 256 * wg_per_cu, iters = 2000;
    CK(hipMalloc(&out, (size_t)grid * 256 * sizeof(V)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((kmis<V, BYTES>), dim3(grid), dim3(256), 0, 0, buf, out, 50, bytes);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((kmis<V, BYTES>), dim3(grid), dim3(256), 0, 0, buf, out, iters, bytes);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double instr_per_cu = (double)wg_per_cu * 4 * iters * 8;
    printf("%-28s %d WG/CU: %.3f ms  -> %.1f clk @2.1GHz per wave-load per CU, %.1f GB/s per CU\n", name, wg_per_cu, ms,
           ms * 1e6 / instr_per_cu * 2.1, instr_per_cu * 64 * sizeof(V) / (ms * 1e6));
    CK(hipFree(buf)); CK(hipFree(out));
}

template <class V, int MODE>
void run(const char *name, int wg_per_cu) {
    const int elems = 8192 / sizeof(V) * 2;   // 16 KiB window: L1-resident
    V *buf, *out;
    CK(hipMalloc(&buf, elems * sizeof(V) + 4096)); CK(hipMemset(buf, 0, elems * sizeof(V) + 4096));
    const int grid = 256 * wg_per_cu, iters = 2000;
    CK(hipMalloc(&out, (size_t)grid * 256 * sizeof(V)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<V, MODE>), dim3(grid), dim3(256), 0, 0, buf, out, 50, elems);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<V, MODE>), dim3(grid), dim3(256), 0, 0, buf, out, iters, elems);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double instr_per_cu = (double)wg_per_cu * 4 * iters * 8;          // wave-level load instructions per CU
    printf("%-28s %d WG/CU: %.3f ms  -> %.1f ns per wave-load per CU = %.1f clk @2.1GHz, %.1f GB/s per CU\n", name, wg_per_cu, ms,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1, instr_per_cu * 64 * sizeof(V) / (ms * 1e6));
    CK(hipFree(buf)); CK(hipFree(out));
}

int main() {
    for (int w : {4, 8}) {
        run<float, 0>("dword  consecutive", w);
        run<double, 0>("dwordx2 consecutive", w);
        run<d2, 0>("dwordx4 consecutive", w);
        run<double, 1>("dwordx2 same address", w);
        run<double, 2>("dwordx2 misaligned +8B", w);
        run<d2, 2>("dwordx4 misaligned +16B", w);
        run<float, 3>("dword  lane stride 8B (lo)", w);
        run<float, 4>("dword  lane stride 8B (hi)", w);
        run<f2, 0>("dwordx2 as float2", w);
        runmis<d2, 0>("dwordx4 byte offset 0", w);
        runmis<d2, 8>("dwordx4 byte offset 8", w);
        runmis<double, 4>("dwordx2 byte offset 4", w);
    }
    return 0;
}
