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

// Two partial arrays at once: both sets of loads are issued before either is consumed and the two block sums share
// their barriers (a consumer prologue is a chain of dependent round trips; this removes one).  Same per-thread
// addition order and same wave / block order as two reduce_partials calls => bit-identical values.
template <class TA, class TB>
__device__ __forceinline__ void reduce_partials2(const TA *__restrict__ pa, const TB *__restrict__ pb, int P, TA *smA, TB *smB,
                                                 TA &ra, TB &rb) {
    TA a = szero<TA>(); TB b = szero<TB>();
    for (int i = threadIdx.x; i < P; i += BLOCK) { const TA va = pa[i]; const TB vb = pb[i]; a = sadd(a, va); b = sadd(b, vb); }
    a = wave_sum(a); b = wave_sum(b);
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { smA[wv] = a; smB[wv] = b; }
    __syncthreads();
    ra = smA[0]; rb = smB[0];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w) { ra = sadd(ra, smA[w]); rb = sadd(rb, smB[w]); }
}

// ---- last-arriving-workgroup finalize of a fused reduction (struct Fin, internal.hpp)
// Inter-workgroup visibility on gfx950 (per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed by
// another CU's stores): every handed-off byte is stored AND loaded with agent-scope (`sc1`) accesses, each storing lane
// drains its store (`s_waitcnt vmcnt(0)`) before it counts its arrival with an agent-scope atomic add, and the workgroup
// whose add returned gridDim.x - 1 loads only after that add has returned (its other waves behind a workgroup barrier) —
// MI355X_MICROARCH.md, "Valid forms", first row of the sc1 hand-off table.  No cache-wide release / acquire is issued.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "st_through / ld_through / finalize_last_block rely on gfx942 / gfx950 semantics: sc1 accesses write through to / read from memory, and stores count in vmcnt (gfx10+ counts them in vscnt). Port the hand-off (release fetch_add + acquire fence) before building for another target."
#endif
__device__ __forceinline__ void st_through(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_through(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_through(cplx *p, cplx v) { st_through(&p->re, v.re); st_through(&p->im, v.im); }
__device__ __forceinline__ void st_through(cplxf *p, cplxf v) { st_through(&p->re, v.re); st_through(&p->im, v.im); }
__device__ __forceinline__ double ld_through(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_through(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ cplx ld_through(const cplx *p) { return cplx{ld_through(&p->re), ld_through(&p->im)}; }
__device__ __forceinline__ cplxf ld_through(const cplxf *p) { return cplxf{ld_through(&p->re), ld_through(&p->im)}; }

// thread 0's store of its workgroup's partial: through to memory when the launch finalizes, plain otherwise
template <class T>
__device__ __forceinline__ void st_partial(const Fin &fin, T *p, T v) {
    if (fin.counter) st_through(p, v); else *p = v;
}

// A producer that returns at its first instruction because the solve's status word is no longer "running" leaves its
// hand-off cells at ZERO: the distributed consumers' in-place all-reduce would otherwise multiply a stale value by the
// world size once per no-op hand-off (up to 3 * poll of them; f32 would reach inf).  Consumers test the status word
// before they use a handed-off value, so the zero is never read as data.
__device__ __forceinline__ void fin_idle(const Fin &fin, bool two) {
    if (fin.counter == nullptr || blockIdx.x != 0 || threadIdx.x != 0) return;
    double *a = reinterpret_cast<double *>(fin.out0);
    a[0] = 0.0; a[1] = 0.0;                                   // a 16-byte cell whatever the scalar type
    if (two) { double *b = reinterpret_cast<double *>(fin.out1); b[0] = 0.0; b[1] = 0.0; }
}

// Called by ALL threads of every workgroup after thread 0 has st_partial'ed its partial(s).  Same per-thread addition
// order and same wave / block order as reduce_partials => the value is bit-identical to what a consumer's prologue (or
// a finalize launch) computes from the same partials.  `two`: the reduction has a second array (base1 / out1, type TB).
template <class TA, class TB>
__device__ __forceinline__ void finalize_last_block(const Fin &fin, bool two, TA *smA, TB *smB) {
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's partials have left the CU ...
        const unsigned int t = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before its arrival counts
        s_last = t == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    const TA *pa = reinterpret_cast<const TA *>(fin.base0);
    TA a = szero<TA>();
    for (int i = threadIdx.x; i < fin.P; i += BLOCK) a = sadd(a, ld_through(pa + i));
    a = block_sum(a, smA);
    if (threadIdx.x == 0) *reinterpret_cast<TA *>(fin.out0) = a;
    TB b = szero<TB>();
    if (two) {
        const TB *pb = reinterpret_cast<const TB *>(fin.base1);
        for (int i = threadIdx.x; i < fin.P; i += BLOCK) b = sadd(b, ld_through(pb + i));
        b = block_sum(b, smB);
        if (threadIdx.x == 0) *reinterpret_cast<TB *>(fin.out1) = b;
    }
    if (fin.tag != 0) mbox_post(fin, a, b);               // peer-to-peer hand-off: every rank's mailbox gets this rank's values
    if (threadIdx.x == 0) __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}

// ---- peer-to-peer hand-off (struct P2pBox, internal.hpp): posted by finalize_last_block, summed by the consumers' prologues
// One 16-byte cell per reduced value (a real scalar: the value and a zero pad; a complex one: re, im — the layout of `red`).
template <class TV> __device__ __forceinline__ void to_cell(TV v, unsigned int *w) {          // 4 words
    w[0] = w[1] = w[2] = w[3] = 0u;
    __builtin_memcpy(w, &v, sizeof(TV));
}
// All threads of the posting workgroup hold the same (a, b): thread t writes granule t % 8 of the entry of THIS rank in rank
// (t / 8)'s mailbox — one naturally aligned 8-byte system-scope store {data word, tag}: whole or absent on the other side.
template <class TA, class TB>
__device__ __forceinline__ void mbox_post(const Fin &fin, TA a, TB b) {
    const P2pBox *bx = fin.box;
    const int world = bx->world;
    if ((int)threadIdx.x >= world * MB_GRAN) return;
    unsigned int w[8];
    to_cell(a, w); to_cell(b, w + 4);
    const int dst = threadIdx.x / MB_GRAN, g = threadIdx.x % MB_GRAN;
    unsigned long long *p = reinterpret_cast<unsigned long long *>(bx->peer[dst] + fin.mb_off + (size_t)bx->rank * MB_ENTRY) + g;
    unsigned int word = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) word = g == k ? w[k] : word;
    __hip_atomic_store(p, (unsigned long long)word | ((unsigned long long)fin.tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// What a consumer reads instead of partials when its producer posted to the mailboxes: `entries` = this rank's mailbox at the
// hand-off's [slot][parity] (one MB_ENTRY per source rank).  Every workgroup of every rank polls the `world` entries until each
// granule carries the hand-off's tag (bounded: `timeout` ticks of the 100 MHz wall clock), then sums cell `cell` of the entries
// in rank order from zero — the same bits in every workgroup of every rank, i.e. the same branch decisions everywhere.
struct MboxSrc { const unsigned long long *entries; int world; unsigned int tag; unsigned long long timeout; };
template <class TA, class TB>
__device__ __forceinline__ bool mbox_sum2(const MboxSrc &m, TA &ra, TB &rb) {
    __shared__ unsigned int s_mb[MB_RANKS][MB_GRAN];
    __shared__ int s_fail;
    if (threadIdx.x == 0) s_fail = 0;
    __syncthreads();
    if ((int)threadIdx.x < m.world * MB_GRAN) {
        const unsigned long long *p = m.entries + threadIdx.x;          // entries are contiguous: [src][granule]
        const unsigned long long t0 = wall_clock64();
        unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while ((unsigned int)(v >> 32) != m.tag) {
            if (wall_clock64() - t0 > m.timeout) { s_fail = 1; break; }
            __builtin_amdgcn_s_sleep(2);
            v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        s_mb[threadIdx.x / MB_GRAN][threadIdx.x % MB_GRAN] = (unsigned int)v;
    }
    __syncthreads();
    if (s_fail) return false;
    ra = szero<TA>(); rb = szero<TB>();
    for (int src = 0; src < m.world; ++src) {
        TA a; TB b;
        __builtin_memcpy(&a, &s_mb[src][0], sizeof(TA));
        __builtin_memcpy(&b, &s_mb[src][4], sizeof(TB));
        ra = sadd(ra, a); rb = sadd(rb, b);
    }
    __syncthreads();                                                    // s_mb may be reused by a later call
    return true;
}
template <class TA>
__device__ __forceinline__ bool mbox_sum1(const MboxSrc &m, TA &ra) { TA dummy; return mbox_sum2<TA, TA>(m, ra, dummy); }

// 16-byte packs: 2 doubles or 1 complex per lane per access (global_load_dwordx4).
template <class T, int PK>
struct alignas(sizeof(T) * PK) Pack {
    T v[PK];
};
template <class T, int PK, bool NT = false>
__device__ __forceinline__ Pack<T, PK> ldp(const T *p, int64_t i) {
    if constexpr (NT) {
        constexpr int W = (int)(sizeof(T) * PK / 4);
        typedef unsigned int uvec __attribute__((ext_vector_type(W)));
        const uvec q = __builtin_nontemporal_load(reinterpret_cast<const uvec *>(p + i * PK));
        Pack<T, PK> r;
        __builtin_memcpy(&r, &q, sizeof(r));
        return r;
    } else {
        return *reinterpret_cast<const Pack<T, PK> *>(p + i * PK);
    }
}
template <class T, int PK, bool NT = false>
__device__ __forceinline__ void stp(T *p, int64_t i, const Pack<T, PK> &v) {
    if constexpr (NT) {
        constexpr int W = (int)(sizeof(T) * PK / 4);
        typedef unsigned int uvec __attribute__((ext_vector_type(W)));
        uvec q;
        __builtin_memcpy(&q, &v, sizeof(q));
        __builtin_nontemporal_store(q, reinterpret_cast<uvec *>(p + i * PK));
    } else {
        *reinterpret_cast<Pack<T, PK> *>(p + i * PK) = v;
    }
}

template <class T> struct pack_width { static constexpr int value = 16 / sizeof(T); };

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Element-wise driver: `body(i)` for every element index, visiting 16-byte packs when PK > 1.
// Grid-stride; the (n mod PK) tail elements are done by the first threads of block 0.
#define SPRS_FOREACH_PACK(n, PK, i)                                                         \
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x, np__ = (int64_t)(n) / (PK), \
                 st__ = (int64_t)gridDim.x * BLOCK;                                         \
         i < np__; i += st__)

constexpr int ROWS_CAP = WAVE;    // rows per stream block: one wavefront owns a block, one lane per row in the reduce phase
constexpr int LONG_ROW = 96;      // rows longer than this go to the wavefront-per-row path
constexpr uint32_t VEC_FLAG = 0x80000000u;


// nnz per stream block = per wavefront (its private LDS slice; x4 wavefronts per workgroup)
template <class T> struct nnz_cap { static constexpr int value = 512; };        // f64: 4 KiB per wavefront
template <> struct nnz_cap<cplx> { static constexpr int value = 320; };         // 5 KiB
template <> struct nnz_cap<float> { static constexpr int value = 512; };        // 2 KiB
template <> struct nnz_cap<cplxf> { static constexpr int value = 512; };        // 4 KiB
static inline int nnz_cap_of(int dtype) { return dtype == DT_Z ? nnz_cap<cplx>::value : 512; }


// Row-block descriptor, precomputed at handle creation so that one 16-byte load tells a workgroup
// everything about its next block (no dependent rowblk -> row_ptr -> row_ptr chain per block).
struct alignas(16) BlkDesc {
    int32_t ra;      // first row
    int32_t rb;      // one past the last row; bit 31 = vector (wavefront-per-row) block
    int32_t pa;      // first nnz
    int32_t nn;      // nnz in the block
};


// Ordering point between a wavefront's LDS writes and its own later LDS reads (and vice versa).  LDS
// operations of one wavefront execute in issue order, so no s_barrier and no wait is needed — this only
// stops the compiler from moving LDS accesses across it.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}


}  // namespace sprs
