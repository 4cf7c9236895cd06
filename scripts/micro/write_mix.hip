// How should a kernel that reads R bytes and writes R/16 bytes issue its writes?  (not part of the product)
// All variants read a 4 GiB buffer with non-temporal 16 B/lane loads (8 in flight per lane) and write 0.25 GiB.
//   V1 store after every tile (2 KiB per workgroup per tile)           — what the SpMV does
//   V2 buffer M tiles' results in LDS, then write M*2 KiB contiguous
//   V3 1/16 of the workgroups only write (streaming), the others only read
//   V4 every workgroup reads its whole share first (results to a small LDS ring, last ones kept), then writes it all
//   V5 pure read and pure write back to back (two launches) for reference
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256, U = 8;
__device__ __forceinline__ int4 ldnt(const int4 *p) {
    int4 r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); r.z = __builtin_nontemporal_load(&p->z); r.w = __builtin_nontemporal_load(&p->w); return r;
}
__device__ __forceinline__ int2 read_tile(const int4 *a, long t) {
    const int4 *p = a + t * (BLOCK * U) + threadIdx.x;
    int4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ldnt(p + u * BLOCK);
    int2 s{0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) { s.x ^= v[u].x ^ v[u].z; s.y ^= v[u].y ^ v[u].w; }
    return s;
}
__global__ __launch_bounds__(BLOCK) void v1(const int4 *__restrict__ a, int2 *__restrict__ b, long ntile) {
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) b[t * BLOCK + threadIdx.x] = read_tile(a, t);
}
// V7: as V1, but the store of tile t is issued AFTER the loads of tile t+1: vmcnt counts loads and stores in one
// in-order queue on gfx9, so in V1 the wait for tile t+1's loads also waits for the write acknowledgement of tile t
__global__ __launch_bounds__(BLOCK) void v7(const int4 *__restrict__ a, int2 *__restrict__ b, long ntile) {
    long t = blockIdx.x;
    if (t >= ntile) return;
    const int4 *p = a + t * (BLOCK * U) + threadIdx.x;
    int4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ldnt(p + u * BLOCK);
    while (true) {
        int2 s{0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) { s.x ^= v[u].x ^ v[u].z; s.y ^= v[u].y ^ v[u].w; }
        const long tn = t + gridDim.x;
        const long tl = tn < ntile ? tn : t;          // past the end: re-read the last tile (unconditional loads)
        const int4 *q = a + tl * (BLOCK * U) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ldnt(q + u * BLOCK);
        __builtin_amdgcn_sched_barrier(0);
        b[t * BLOCK + threadIdx.x] = s;
        if (tn >= ntile) break;
        t = tn;
    }
}
template <int M>
__global__ __launch_bounds__(BLOCK) void v2(const int4 *__restrict__ a, int2 *__restrict__ b, long ntile) {
    __shared__ int2 buf[M][BLOCK];
    const long ngroup = ntile / M;   // ntile is a multiple of M
    for (long g = blockIdx.x; g < ngroup; g += gridDim.x) {
#pragma unroll 1
        for (int m = 0; m < M; ++m) buf[m][threadIdx.x] = read_tile(a, g * M + m);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; ++m) b[(g * M + m) * BLOCK + threadIdx.x] = buf[m][threadIdx.x];
        __syncthreads();
    }
}
__global__ __launch_bounds__(BLOCK) void v3(const int4 *__restrict__ a, int2 *__restrict__ b, long ntile, int *sink) {
    if ((blockIdx.x & 15) == 15) {   // writer: streams the whole of b with the other writers
        const int w = blockIdx.x >> 4, nw = gridDim.x >> 4;
        for (long t = w; t < ntile; t += nw) b[t * BLOCK + threadIdx.x] = int2{(int)t, w};
    } else {
        const int r = blockIdx.x - (blockIdx.x >> 4), nr = gridDim.x - (gridDim.x >> 4);
        int keep = 0;
        for (long t = r; t < ntile; t += nr) { int2 s = read_tile(a, t); keep ^= s.x ^ s.y; }
        if (keep == 0x12345678) *sink = keep;
    }
}
__global__ __launch_bounds__(BLOCK) void v4(const int4 *__restrict__ a, int2 *__restrict__ b, long ntile) {
    int2 keep{0, 0};
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) { int2 s = read_tile(a, t); keep.x ^= s.x; keep.y ^= s.y; }
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) b[t * BLOCK + threadIdx.x] = keep;
}
// V6: results buffered in LDS (up to M tiles) and flushed when the chip-wide real-time counter (100 MHz, the same
// value on every CU) enters a new epoch of `period` ticks: every workgroup writes at the same wall-clock time,
// without any communication, so HBM sees long pure-read intervals separated by short write bursts
template <int M>
__global__ __launch_bounds__(BLOCK) void v6(const int4 *__restrict__ a, int2 *__restrict__ b, long ntile, long period) {
    __shared__ int2 buf[M][BLOCK];
    __shared__ long slot[M];
    __shared__ int flush;
    long epoch = (long)wall_clock64() / period;
    int m = 0;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        buf[m][threadIdx.x] = read_tile(a, t);
        if (threadIdx.x == 0) {
            slot[m] = t;
            const long e = (long)wall_clock64() / period;
            flush = (e != epoch) || (m == M - 1) || (t + gridDim.x >= ntile);
            epoch = e;
        }
        ++m;
        __syncthreads();
        if (flush) {
            for (int k = 0; k < m; ++k) b[slot[k] * BLOCK + threadIdx.x] = buf[k][threadIdx.x];
            m = 0;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(BLOCK) void rd_only(const int4 *__restrict__ a, long ntile, int *sink) {
    int keep = 0;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) { int2 s = read_tile(a, t); keep ^= s.x ^ s.y; }
    if (keep == 0x12345678) *sink = keep;
}
__global__ __launch_bounds__(BLOCK) void wr_only(int2 *__restrict__ b, long ntile) {
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) b[t * BLOCK + threadIdx.x] = int2{(int)t, 1};
}
int main() {
    const long bytes = 4L << 30, ntile = bytes / (16 * BLOCK * U);
    char *a, *b; int *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes / 16 + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, int grid, auto launch) {
        launch(); launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < 6; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 6;
        printf("%-52s grid %5d : %8.1f us\n", name, grid, ms * 1e3); fflush(stdout);
    };
    for (int G : {1024}) {
        timeit("read only (4 GiB, nt)", G, [&]() { rd_only<<<G, BLOCK>>>((const int4 *)a, ntile, sink); });
        timeit("write only (0.25 GiB, 8 B/lane)", G, [&]() { wr_only<<<G, BLOCK>>>((int2 *)b, ntile); });
        timeit("V5 read launch + write launch", G, [&]() { rd_only<<<G, BLOCK>>>((const int4 *)a, ntile, sink); wr_only<<<G, BLOCK>>>((int2 *)b, ntile); });
        timeit("V1 store after every tile", G, [&]() { v1<<<G, BLOCK>>>((const int4 *)a, (int2 *)b, ntile); });
        timeit("V7 store of tile t after the loads of tile t+1", G, [&]() { v7<<<G, BLOCK>>>((const int4 *)a, (int2 *)b, ntile); });
        timeit("V2 LDS-buffered, 4 tiles (8 KiB bursts)", G, [&]() { v2<4><<<G, BLOCK>>>((const int4 *)a, (int2 *)b, ntile); });
        timeit("V2 LDS-buffered, 16 tiles (32 KiB bursts)", G, [&]() { v2<16><<<G, BLOCK>>>((const int4 *)a, (int2 *)b, ntile); });
        timeit("V3 dedicated writer workgroups (1 in 16)", G, [&]() { v3<<<G, BLOCK>>>((const int4 *)a, (int2 *)b, ntile, sink); });
        timeit("V6 clock-synchronised flush, 40 us epochs, M=16", G, [&]() { v6<16><<<G, BLOCK>>>((const int4 *)a, (int2 *)b, ntile, 4000); });
        timeit("V6 clock-synchronised flush, 80 us epochs, M=16", G, [&]() { v6<16><<<G, BLOCK>>>((const int4 *)a, (int2 *)b, ntile, 8000); });
        timeit("V6 clock-synchronised flush, 20 us epochs, M=16", G, [&]() { v6<16><<<G, BLOCK>>>((const int4 *)a, (int2 *)b, ntile, 2000); });
        timeit("V6 never by clock (period huge), M=16", G, [&]() { v6<16><<<G, BLOCK>>>((const int4 *)a, (int2 *)b, ntile, 1L << 40); });
        timeit("V4 per-workgroup read phase then write phase", G, [&]() { v4<<<G, BLOCK>>>((const int4 *)a, (int2 *)b, ntile); });
    }
    return 0;
}
