// Device-side building blocks: 64-lane wavefront reductions, the fixed-order partial
// reduction every consumer kernel runs in its prologue, and 16-byte packed vector access.
#pragma once
#include "internal.hpp"

namespace sprs {

constexpr int WAVE = 64;
constexpr int NWAVE = BLOCK / WAVE;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v = v + __shfl_down(v, off, WAVE);
    return v;
}
__device__ __forceinline__ cplx wave_sum(cplx v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        v.re = v.re + __shfl_down(v.re, off, WAVE);
        v.im = v.im + __shfl_down(v.im, off, WAVE);
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v = v + __shfl_down(v, off, WAVE);
    return v;
}
__device__ __forceinline__ cplxf wave_sum(cplxf v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        v.re = v.re + __shfl_down(v.re, off, WAVE);
        v.im = v.im + __shfl_down(v.im, off, WAVE);
    }
    return v;
}

// Sum over the workgroup; every thread returns the same value.  Fixed order: butterfly inside
// each wavefront, then wave 0..3 left to right.  `smem` needs NWAVE elements; safe to call
// back-to-back with the same buffer.
template <class T>
__device__ __forceinline__ T block_sum(T v, T *smem) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    __syncthreads();  // previous users of smem are done
    if (lane == 0) smem[wv] = v;
    __syncthreads();
    T acc = smem[0];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w) acc = sadd(acc, smem[w]);
    return acc;
}

// Every workgroup reduces the same P partials in the same order => bit-identical scalars in
// every workgroup (and in every later kernel that re-reduces them).
template <class T>
__device__ __forceinline__ T reduce_partials(const T *__restrict__ part, int P, T *smem) {
    T acc = szero<T>();
    for (int i = threadIdx.x; i < P; i += BLOCK) acc = sadd(acc, part[i]);
    return block_sum(acc, smem);
}

// 16-byte packs: 2 doubles or 1 complex per lane per access (global_load_dwordx4).
template <class T, int PK>
struct alignas(sizeof(T) * PK) Pack {
    T v[PK];
};
template <class T, int PK>
__device__ __forceinline__ Pack<T, PK> ldp(const T *p, int64_t i) {
    return *reinterpret_cast<const Pack<T, PK> *>(p + i * PK);
}
template <class T, int PK>
__device__ __forceinline__ void stp(T *p, int64_t i, const Pack<T, PK> &v) {
    *reinterpret_cast<Pack<T, PK> *>(p + i * PK) = v;
}

template <class T> struct pack_width { static constexpr int value = 16 / sizeof(T); };

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Element-wise driver: `body(i)` for every element index, visiting 16-byte packs when PK > 1.
// Grid-stride; the (n mod PK) tail elements are done by the first threads of block 0.
#define SPRS_FOREACH_PACK(n, PK, i)                                                         \
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x, np__ = (int64_t)(n) / (PK), \
                 st__ = (int64_t)gridDim.x * BLOCK;                                         \
         i < np__; i += st__)

}  // namespace sprs
