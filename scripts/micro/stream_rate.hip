// What read rate does this box's HBM give, and to which access shape?  (not part of the product)
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/stream_rate.hip -o scripts/micro/stream_rate
// Reads a 4 GiB buffer (far beyond the 256 MiB Infinity Cache) with W-byte loads per lane, U independent loads in
// flight per lane, grid G, either grid-strided (wave tiles interleaved over the whole buffer) or blocked (every
// workgroup owns one contiguous piece), plain or non-temporal; then the same with a concurrent write stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)
constexpr int BLOCK = 256;
template <class V> __device__ __forceinline__ int fold(V v);
template <> __device__ __forceinline__ int fold<int>(int v) { return v; }
template <> __device__ __forceinline__ int fold<int2>(int2 v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ int fold<int4>(int4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <bool NT, class V> __device__ __forceinline__ V ld(const V *p) {
    if constexpr (NT) {
        if constexpr (sizeof(V) == 4) return __builtin_nontemporal_load(p);
        else if constexpr (sizeof(V) == 8) { V r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); return r; }
        else { V r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); r.z = __builtin_nontemporal_load(&p->z); r.w = __builtin_nontemporal_load(&p->w); return r; }
    } else return *p;
}
// tile = BLOCK * U elements of V, contiguous; tiles dealt grid-strided or blocked
template <class V, int U, bool NT, bool BLOCKED>
__global__ __launch_bounds__(BLOCK) void rd(const V *__restrict__ a, long ntile, int *sink) {
    int acc = 0;
    long t0, t1, ts;
    if (BLOCKED) { const long per = (ntile + gridDim.x - 1) / gridDim.x; t0 = blockIdx.x * per; t1 = min(ntile, t0 + per); ts = 1; }
    else { t0 = blockIdx.x; t1 = ntile; ts = gridDim.x; }
    for (long t = t0; t < t1; t += ts) {
        const V *p = a + t * (BLOCK * U) + threadIdx.x;
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(p + u * BLOCK);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= fold<V>(v[u]);
    }
    if (acc == 0x12345678) *sink = acc;
}
// read a (16 B/lane, U in flight per lane, NT or plain) and write one W-byte store per lane every K-th tile
// (write volume = read volume * W / (16 * U * K)); SNT: non-temporal stores
template <int U, int K, bool NT, bool SNT, class W>
__global__ __launch_bounds__(BLOCK) void rdwr(const int4 *__restrict__ a, W *__restrict__ b, long ntile, int *sink) {
    int keep = 0;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int4 *p = a + t * (BLOCK * U) + threadIdx.x;
        int4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(p + u * BLOCK);
        int4 s = v[0];
#pragma unroll
        for (int u = 1; u < U; ++u) { s.x ^= v[u].x; s.y ^= v[u].y; s.z ^= v[u].z; s.w ^= v[u].w; }
        keep ^= s.x ^ s.y ^ s.z ^ s.w;
        if (t % K == 0) {
            W *q = b + (t / K) * BLOCK + threadIdx.x;
            if constexpr (sizeof(W) == 16) { if (SNT) { __builtin_nontemporal_store(s.x, &q->x); __builtin_nontemporal_store(s.y, &q->y); __builtin_nontemporal_store(s.z, &q->z); __builtin_nontemporal_store(s.w, &q->w); } else *q = s; }
            else { int2 w2{s.x, s.y}; if (SNT) { __builtin_nontemporal_store(w2.x, &q->x); __builtin_nontemporal_store(w2.y, &q->y); } else *q = w2; }
        }
    }
    if (keep == 0x12345678) *sink = keep;
}
template <class W, bool SNT>
__global__ __launch_bounds__(BLOCK) void wr(W *__restrict__ b, long nel) {
    for (long g = blockIdx.x * (long)BLOCK + threadIdx.x; g < nel; g += (long)gridDim.x * BLOCK) {
        W *q = b + g;
        if constexpr (sizeof(W) == 16) { if (SNT) { __builtin_nontemporal_store(1, &q->x); __builtin_nontemporal_store(2, &q->y); __builtin_nontemporal_store(3, &q->z); __builtin_nontemporal_store(4, &q->w); } else *q = int4{1, 2, 3, 4}; }
        else { if (SNT) { __builtin_nontemporal_store(1, &q->x); __builtin_nontemporal_store(2, &q->y); } else *q = int2{1, 2}; }
    }
}
int main() {
    const long bytes = 4L << 30;
    char *a, *b; int *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, int grid, auto launch, double moved) {
        launch(); launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < 6; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 6;
        printf("%-44s grid %5d : %8.1f us %7.0f GB/s\n", name, grid, ms * 1e3, moved / ms / 1e6); fflush(stdout);
    };
#define RD(V, U, NT, BL, G) timeit("rd " #V " U=" #U " nt=" #NT " blocked=" #BL, G, [&]() { rd<V, U, NT, BL><<<G, BLOCK>>>((const V *)a, bytes / (sizeof(V) * BLOCK * U), sink); }, (double)bytes)
    for (int G : {1024, 2048}) {
        RD(int4, 1, false, false, G); RD(int4, 2, false, false, G); RD(int4, 4, false, false, G); RD(int4, 8, false, false, G);
    }
    for (int G : {1024, 2048}) {
        RD(int4, 4, true, false, G); RD(int4, 8, true, false, G);
        RD(int4, 4, false, true, G); RD(int4, 4, true, true, G);
        RD(int2, 4, false, false, G); RD(int2, 8, false, false, G); RD(int2, 16, false, false, G);
        RD(int, 8, false, false, G); RD(int, 16, false, false, G);
    }
#define RW(U, K, NT, SNT, W, G) timeit("rdwr U=" #U " K=" #K " ldnt=" #NT " stnt=" #SNT " store " #W, G, [&]() { rdwr<U, K, NT, SNT, W><<<G, BLOCK>>>((const int4 *)a, (W *)b, bytes / (16 * BLOCK * U), sink); }, (double)bytes * (1.0 + sizeof(W) / (16.0 * U * K)))
    for (int G : {1024, 2048}) {
        RW(8, 1, false, false, int4, G); RW(8, 1, true, false, int4, G); RW(8, 1, true, true, int4, G); RW(8, 1, false, true, int4, G);
        RW(8, 1, false, false, int2, G); RW(8, 1, true, false, int2, G); RW(8, 1, true, true, int2, G);
        RW(8, 2, true, false, int4, G); RW(8, 2, true, true, int4, G); RW(8, 2, true, false, int2, G); RW(8, 2, true, true, int2, G);
        RW(4, 1, true, true, int4, G); RW(2, 1, true, true, int4, G); RW(1, 1, true, true, int4, G); RW(1, 1, false, false, int4, G);
    }
#define WR(W, SNT, G) timeit("wr " #W " stnt=" #SNT, G, [&]() { wr<W, SNT><<<G, BLOCK>>>((W *)b, bytes / sizeof(W)); }, (double)bytes)
    for (int G : {1024, 2048}) { WR(int4, false, G); WR(int4, true, G); WR(int2, false, G); WR(int2, true, G); }
    return 0;
}
