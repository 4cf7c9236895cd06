// Dictionary-compressed CSR SpMV — the same arithmetic as spmv.hip on fewer HBM bytes and fewer instructions.
//
// The SpMV streams (col_idx, val): 12 of its ~15 bytes per nnz (f64).  Matrices that come from grids and
// bands (every BASELINE config; the reference's own tests and benches) repeat themselves:
//   * the column OFFSET col - row takes a handful of values (7 for the 7-point stencil, 9 for the band of
//     cfg 3, the halo blocks of a slab partition add a few more);
//   * constant-coefficient operators also repeat their VALUES (two distinct ones in cfg 2 and cfg 5), so
//     the (offset, value) PAIRS are few as well (7 in cfg 5).
// At handle creation the distinct offsets — and, for real scalars, the distinct value bit patterns and the
// distinct (offset, value) pairs — are collected on the device.  With <= 256 of them each nnz is re-encoded
// as ONE BYTE (CSR-DU / CSR-VI style, Kourtis et al.):
//   mode 1 "offset codes":  1 B offset code + the original value           12 -> 9 B/nnz (f64)
//   mode 2 "pair codes":    1 B code of the (offset, value) pair           12 -> 1 B/nnz
// Values are matched by BIT PATTERN, the tables hold the original values and every product x[col]*val is
// formed and added in the original order, so y is BIT-IDENTICAL to the plain kernel's (and to the
// reference's fold, mat.rs:100-105).  Matrices that do not qualify (too many offsets, rows longer than
// LONG_ROW) keep the plain stream; nothing is approximated.
//
// Kernel shape: as in spmv.hip a wavefront owns a row block (<= 64 rows, <= CAP nnz).  The code bytes are
// staged to LDS with aligned dword loads; then lane r walks row ra + r: one LDS byte read gives the code,
// one LDS table read gives (offset*sizeof(T), value), and x is gathered with base-in-SGPR + 32-bit lane
// offset addressing — consecutive rows of a stencil gather consecutive x, so the gather is itself
// coalesced.  Products are added left to right from zero.  Once the stream is this small the kernel is
// bound by the CU's vector-ALU and vector-memory issue, not by HBM: the loop is written to keep the
// per-nnz instruction count down (immediate-offset LDS reads, no 64-bit address arithmetic).
// For HBM-sized stencil-like matrices the LDS x-window TILE kernels take over (one staged window of x per 4096 rows
// for all near columns): those run at the memory system's rate again.
//
// In this file, in order:
//   dict_collect_kernel / dict_encode_kernel / dict_pair_*      the dictionaries and the code streams (creation)
//   dict_walk, spmv_dict_kernel                                 64-row blocks, lane per row: offset codes (all types), pair codes (c64 / f32 / c32)
//   mark_uniform_kernel                                         uniform and seam blocks (creation)
//   full_uniform_block, pair2_walk, spmv_pair2_kernel           128-row blocks, two rows per lane: f64 pair codes
//   tile_mark_kernel / tile_flag_kernel                         runs of one pattern (creation)
//   spmv_tile_kernel, spmv_tile_off_kernel                      LDS x-window tiles: f64 pair codes / f64 offset codes + values
//   xcd_period_order, build_tile_plan, build_dict_t             schedules, tile plans, the creation driver
//   launch_spmv_dict                                            which kernel a launch takes
#include <algorithm>
#include <cstring>
#include <map>

#include "device.hpp"

namespace sprs {

namespace {

constexpr int TAB = 256;          // dictionary entries (one byte per code)
constexpr int HSLOTS = 1024;      // open-addressing table used while collecting (4x the dictionary)
constexpr uint32_t EMPTY32 = 0x80000000u;            // INT32_MIN: never a valid col - row (|.| < 2^31 - 1)
constexpr uint64_t EMPTY64 = 0xFFFFFFFFFFFFFFFFull;  // a NaN pattern; a matrix holding it is not compressed

__device__ __forceinline__ uint32_t hash32(uint32_t k) { k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16; return k; }
__device__ __forceinline__ uint32_t hash64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; return (uint32_t)k; }

// value keys: the bit pattern, so -0.0 / NaN payloads survive the round trip
__device__ __forceinline__ uint64_t key_of(double v) { return (uint64_t)__double_as_longlong(v); }
__device__ __forceinline__ uint64_t key_of(float v) { return (uint64_t)__float_as_uint(v); }

// insert-or-find; returns the slot or -1 when the table is (being) abandoned
__device__ __forceinline__ int probe32(uint32_t *tab, int *count, uint32_t key, bool insert) {
    uint32_t h = hash32(key) & (HSLOTS - 1);
    for (int t = 0; t < HSLOTS; ++t, h = (h + 1) & (HSLOTS - 1)) {
        uint32_t cur = __hip_atomic_load(tab + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return (int)h;
        if (cur == EMPTY32) {
            if (!insert) return -1;
            if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > TAB) return -1;
            cur = atomicCAS(tab + h, EMPTY32, key);
            if (cur == EMPTY32) { atomicAdd(count, 1); return (int)h; }
            if (cur == key) return (int)h;
        }
    }
    return -1;
}
__device__ __forceinline__ int probe64(unsigned long long *tab, int *count, uint64_t key, bool insert) {
    uint32_t h = hash64(key) & (HSLOTS - 1);
    for (int t = 0; t < HSLOTS; ++t, h = (h + 1) & (HSLOTS - 1)) {
        uint64_t cur = __hip_atomic_load(tab + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return (int)h;
        if (cur == EMPTY64) {
            if (!insert) return -1;
            if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > TAB) return -1;
            cur = atomicCAS(tab + h, (unsigned long long)EMPTY64, (unsigned long long)key);
            if (cur == EMPTY64) { atomicAdd(count, 1); return (int)h; }
            if (cur == key) return (int)h;
        }
    }
    return -1;
}

// The same insert-or-find on a workgroup's LDS table (no agent-scope traffic: the hot path of the collection pass)
__device__ __forceinline__ int lds_probe32(uint32_t *tab, int *count, uint32_t key) {
    uint32_t h = hash32(key) & (HSLOTS - 1);
    for (int t = 0; t < HSLOTS; ++t, h = (h + 1) & (HSLOTS - 1)) {
        uint32_t cur = __hip_atomic_load(tab + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == key) return (int)h;
        if (cur == EMPTY32) {
            if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > TAB) return -1;
            cur = atomicCAS(tab + h, EMPTY32, key);
            if (cur == EMPTY32) { atomicAdd(count, 1); return (int)h; }
            if (cur == key) return (int)h;
        }
    }
    return -1;
}
__device__ __forceinline__ int lds_probe64(unsigned long long *tab, int *count, uint64_t key) {
    uint32_t h = hash64(key) & (HSLOTS - 1);
    for (int t = 0; t < HSLOTS; ++t, h = (h + 1) & (HSLOTS - 1)) {
        uint64_t cur = __hip_atomic_load(tab + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == key) return (int)h;
        if (cur == EMPTY64) {
            if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > TAB) return -1;
            cur = atomicCAS(tab + h, (unsigned long long)EMPTY64, (unsigned long long)key);
            if (cur == EMPTY64) { atomicAdd(count, 1); return (int)h; }
            if (cur == key) return (int)h;
        }
    }
    return -1;
}

// counts[0] = distinct offsets, counts[1] = distinct values, counts[2] = value dictionary impossible
//
// Two levels (the analogue of mkl_sparse_optimize must stay cheap next to the solve it prepares, mkl_mat.rs:81-148): a
// workgroup collects the distinct keys of ITS rows in LDS tables — no agent-scope access per entry — and merges them into
// the global tables once, at its end: <= 257 inserts per workgroup and dictionary instead of one atomic probe chain per
// entry (cfg 3: 14.6 ms -> well under 1 ms, profiles/r03_tuning.md).  A table that takes its 257th key is dead, in LDS as
// in HBM: offsets dead => the workgroup stops (no compressed stream at all); values dead => it stops looking at values.
// The global tables end up holding the same key SETS as a flat insertion; the codes are assigned on the host in sorted
// key order, so the dictionaries are identical.
template <class T, bool VALS>
__global__ __launch_bounds__(BLOCK) void dict_collect_kernel(int n, const int32_t *__restrict__ row_ptr,
                                                             const int32_t *__restrict__ col_idx, const T *__restrict__ val,
                                                             uint32_t *off_h, unsigned long long *val_h, int *counts) {
    __shared__ uint32_t s_off[HSLOTS];
    __shared__ unsigned long long s_val[VALS ? HSLOTS : 1];
    __shared__ int s_cnt[4];           // [0] offsets, [1] values held by the LDS tables, [2] a value equal to EMPTY64 was seen
    for (int i = threadIdx.x; i < HSLOTS; i += BLOCK) { s_off[i] = EMPTY32; if (VALS) s_val[i] = EMPTY64; }
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t last_off = EMPTY32;
    uint64_t last_val = EMPTY64;
    bool vals_dead = !VALS;
    int trip = 0;
    for (int row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK, ++trip) {
        // the overflow flags are re-read once per row: this workgroup's (LDS) every time, the chip's every 16th trip
        if (__hip_atomic_load(s_cnt + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > TAB) break;
        if ((trip & 15) == 0 && __hip_atomic_load(counts + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > TAB) break;
        if constexpr (VALS) {
            if (!vals_dead) {
                vals_dead = __hip_atomic_load(s_cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > TAB ||
                            __hip_atomic_load(s_cnt + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
                if (!vals_dead && (trip & 15) == 0)
                    vals_dead = __hip_atomic_load(counts + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > TAB ||
                                __hip_atomic_load(counts + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            }
        }
        const int ks = row_ptr[row], ke = row_ptr[row + 1];
        for (int k = ks; k < ke; ++k) {
            const uint32_t d = (uint32_t)(col_idx[k] - row);
            if (d != last_off) { (void)lds_probe32(s_off, s_cnt + 0, d); last_off = d; }
            if constexpr (VALS) {
                if (vals_dead) continue;
                const uint64_t kv = key_of(val[k]);
                if (kv == EMPTY64) { s_cnt[2] = 1; continue; }
                if (kv != last_val) { (void)lds_probe64(s_val, s_cnt + 1, kv); last_val = kv; }
            }
        }
    }
    __syncthreads();
    // ---- merge: every key this workgroup saw goes into the global table once
    if (s_cnt[0] > TAB) { if (threadIdx.x == 0) atomicMax(counts + 0, TAB + 1); }
    else
        for (int i = threadIdx.x; i < HSLOTS; i += BLOCK)
            if (s_off[i] != EMPTY32) (void)probe32(off_h, counts + 0, s_off[i], true);
    if constexpr (VALS) {
        if (s_cnt[2] != 0) { if (threadIdx.x == 0) counts[2] = 1; }
        else if (s_cnt[1] > TAB) { if (threadIdx.x == 0) atomicMax(counts + 1, TAB + 1); }
        else
            for (int i = threadIdx.x; i < HSLOTS; i += BLOCK)
                if (s_val[i] != EMPTY64) (void)probe64(val_h, counts + 1, s_val[i], true);
    }
}

// Encoding pass: the (now read-only) hash tables and the code of every slot are staged in LDS once per workgroup; the
// per-entry look-ups never leave the CU.
template <class T, bool VALS>
__global__ __launch_bounds__(BLOCK) void dict_encode_kernel(int n, const int32_t *__restrict__ row_ptr,
                                                            const int32_t *__restrict__ col_idx, const T *__restrict__ val,
                                                            const uint32_t *__restrict__ off_h, const uint8_t *__restrict__ off_code_of_slot,
                                                            const unsigned long long *__restrict__ val_h, const uint8_t *__restrict__ val_code_of_slot,
                                                            uint8_t *__restrict__ idx_code, uint8_t *__restrict__ val_code,
                                                            int *__restrict__ bad) {
    __shared__ uint32_t s_off[HSLOTS];
    __shared__ uint8_t s_offc[HSLOTS];
    __shared__ unsigned long long s_val[VALS ? HSLOTS : 1];
    __shared__ uint8_t s_valc[VALS ? HSLOTS : 1];
    for (int i = threadIdx.x; i < HSLOTS; i += BLOCK) {
        s_off[i] = off_h[i]; s_offc[i] = off_code_of_slot[i];
        if constexpr (VALS) { s_val[i] = val_h[i]; s_valc[i] = val_code_of_slot[i]; }
    }
    __syncthreads();
    auto find32 = [&](uint32_t key) -> int {
        uint32_t h = hash32(key) & (HSLOTS - 1);
        for (int t = 0; t < HSLOTS; ++t, h = (h + 1) & (HSLOTS - 1)) {
            const uint32_t cur = s_off[h];
            if (cur == key) return (int)h;
            if (cur == EMPTY32) return -1;
        }
        return -1;
    };
    [[maybe_unused]] auto find64 = [&](uint64_t key) -> int {
        uint32_t h = hash64(key) & (HSLOTS - 1);
        for (int t = 0; t < HSLOTS; ++t, h = (h + 1) & (HSLOTS - 1)) {
            const uint64_t cur = s_val[h];
            if (cur == key) return (int)h;
            if (cur == EMPTY64) return -1;
        }
        return -1;
    };
    int lbad = 0;
    for (int row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK) {
        for (int k = row_ptr[row]; k < row_ptr[row + 1]; ++k) {
            const int so = find32((uint32_t)(col_idx[k] - row));
            if (so < 0) { lbad = 1; continue; }
            idx_code[k] = s_offc[so];
            if constexpr (VALS) {
                const int sv = find64(key_of(val[k]));
                if (sv < 0) { lbad = 1; continue; }
                val_code[k] = s_valc[sv];
            }
        }
    }
    if (lbad) *bad = 1;
}

// mark the (offset code, value code) pairs that occur / translate them to pair codes
__global__ __launch_bounds__(BLOCK) void dict_pair_mark_kernel(int64_t nnz, const uint8_t *__restrict__ idx_code,
                                                               const uint8_t *__restrict__ val_code, uint8_t *__restrict__ seen) {
    for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * BLOCK) {
        const int pr = idx_code[k] | (val_code[k] << 8);
        if (seen[pr] == 0) seen[pr] = 1;       // benign race: every writer stores 1
    }
}
__global__ __launch_bounds__(BLOCK) void dict_pair_encode_kernel(int64_t nnz, const uint8_t *__restrict__ idx_code,
                                                                 const uint8_t *__restrict__ val_code,
                                                                 const uint8_t *__restrict__ pair_of, uint8_t *__restrict__ pair_code) {
    for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * BLOCK)
        pair_code[k] = pair_of[idx_code[k] | (val_code[k] << 8)];
}

// ---------------------------------------------------------------------------------------------------------
constexpr int CPAD = 16;    // readable bytes behind a block's codes: the row phase reads up to 7 + 3 bytes past them

// LDS table entry of the pair stream: byte offset of the column relative to the row, and the value
template <class T> struct alignas(sizeof(T) >= 8 ? 16 : 8) PairEnt { int32_t off8; T val; };

// What a wavefront loads for one row block before it can work on it.  The loads of block i+1 are issued
// before block i is processed (and the descriptor of block i+2 before that), so a block costs one exposed
// memory round trip — its x gather — instead of three dependent ones (descriptor -> codes/row_ptr -> x).
template <class T, bool PAIR, int ITEMS>
struct BlkLoads {
    int ra, rb, pa, nn;      // descriptor (nn = entries of the block, also for uniform blocks)
    int ulen;                // > 0: uniform block — every row repeats the first row's ulen (<= UNI_OFF_MAXLEN) codes; no row_ptr, 1-9 code dwords
    int s;                   // row_ptr[row] of this lane's row
    T uu;                    // dot operand of this lane's row
    uint32_t wc[2];          // code dwords
    int di[2];               // ... and the LDS slots they go to
    T vv[PAIR ? 1 : ITEMS];  // values (offset-code stream only)
};

// WV (f64 offset codes only): the block's values are read with 16 bytes per lane over its 16-byte-aligned window (entries
// 2l, 2l + 1 of [pa - (pa & 1), ..) per load: 4 loads per 512 entries instead of 8) and staged to LDS with 16-byte stores;
// the last 2-entry group of val, which may reach one entry past the array, comes from the handle's zero-padded tail copy.
struct alignas(16) V2d { double a, b; };
// The walk of the 64-row-block kernels of the compressed streams over `n_rowblk` blocks (positions of `order`, or natural
// order), shared by spmv_dict_kernel (the whole matrix) and spmv_tile_kernel's offset-code flavour (the blocks outside its
// tiles).  s_pair / s_off8: the staged tables; s_c: NWAVE zeroed code slices of CW dwords; s_v: NWAVE zeroed value slices of
// s_v_stride (>= CAP + 16) entries, 16-byte aligned (offset-code stream).  d0 / d1: the lane's running dot partials.
template <class T, int DOT, bool CONJX, bool PAIR, bool WV>
__device__ __forceinline__ void dict_walk(int n_rowblk, int xcd_chunk, const BlkDesc *__restrict__ desc, const int32_t *__restrict__ order,
                                          const int32_t *__restrict__ row_ptr, const uint8_t *__restrict__ code,
                                          const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y, const T *__restrict__ u,
                                          const V2d *__restrict__ tail2, int g2_last,
                                          const PairEnt<T> *s_pair, const int32_t *s_off8, uint32_t (*s_c)[(nnz_cap<T>::value + 3 + CPAD + 3) / 4],
                                          T *s_v, int s_v_stride, T &d0, T &d1) {
    constexpr int CAP = nnz_cap<T>::value;          // nnz per row block (per wavefront)
    constexpr int ITEMS = CAP / WAVE;
    using Loads = BlkLoads<T, PAIR, ITEMS>;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint8_t *cb = reinterpret_cast<const uint8_t *>(s_c[wv]);
    [[maybe_unused]] T *vs = s_v + (PAIR ? 0 : wv) * (size_t)s_v_stride;
    const char *xbytes = reinterpret_cast<const char *>(x);

    int b, bstep, bend;                             // the persistent walk of spmv.hip
    if (xcd_chunk) {
        const int chunk = (n_rowblk + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        b = xcd * chunk + (blockIdx.x >> 3) * NWAVE + wv;
        bstep = (gridDim.x >> 3) * NWAVE;
        bend = min(n_rowblk, (xcd + 1) * chunk);
    } else {
        b = blockIdx.x * NWAVE + wv; bstep = gridDim.x * NWAVE; bend = n_rowblk;
    }
    if (b >= bend) b = bend;                        // falls through to the partial sums below

    // Descriptors (and schedule entries) are fetched with VECTOR loads on a wave-uniform address: scalar loads
    // return out of order, so one in flight would turn every later LDS wait into a full lgkmcnt(0) stall.
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef const v4i __attribute__((address_space(1))) *gv4i_p;
    typedef const int32_t __attribute__((address_space(1))) *gi32_p;
    uintptr_t desc_a = reinterpret_cast<uintptr_t>(desc), order_a = reinterpret_cast<uintptr_t>(order);
    asm volatile("" : "+v"(desc_a));                // hide the uniformity: keeps the loads on the vector path
    asm volatile("" : "+v"(order_a));
    const gv4i_p desc_v = reinterpret_cast<gv4i_p>(desc_a);
    const gi32_p order_v = reinterpret_cast<gi32_p>(order_a);
    auto load_desc = [&](int bi) -> BlkDesc { const v4i q = desc_v[bi]; return BlkDesc{q.x, q.y, q.z, q.w}; };
    auto block_index = [&](int bi) -> int { return order ? order_v[bi] : bi; };
    // ... and turned back into scalars where they are consumed, so that everything derived from a descriptor
    // (block bounds, code alignment, branch conditions) is scalar-ALU work instead of 64-lane vector work
    auto uniform = [&](const BlkDesc &d) -> BlkDesc {
        return BlkDesc{__builtin_amdgcn_readfirstlane(d.ra), __builtin_amdgcn_readfirstlane(d.rb),
                       __builtin_amdgcn_readfirstlane(d.pa), __builtin_amdgcn_readfirstlane(d.nn)};
    };
    // Phase 1 of a block: issue its loads (unconditional, clamped addresses: they go out back to back).
    // Nothing here uses a loaded value, so the wavefront does not wait.
    auto issue = [&](const BlkDesc &d, Loads &L) {
        // dictionary matrices have no vector blocks (bit 31); bit 30 = uniform block (descriptors of the offset-code
        // stream only, mark_uniform_kernel): its nn field holds the common row length, not the block's entry count
        L.ra = d.ra; L.rb = d.rb & 0x3fffffff; L.pa = d.pa;
        const bool uni = ((uint32_t)d.rb & UNI2) != 0;                     // scalar
        L.ulen = uni ? d.nn : 0;
        L.nn = uni ? (L.rb - L.ra) * d.nn : d.nn;
        const int r = L.ra + lane;
        const int rcl = r < L.rb ? r : L.rb - 1;
        // uniform base + 32-bit lane offset everywhere (launch checks the sizes): no 64-bit address arithmetic
        if (uni) L.s = L.pa + (rcl - L.ra) * L.ulen;                       // every row has ulen entries: row_ptr is not read
        else L.s = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(row_ptr) + (uint32_t)rcl * 4u);   // row_ptr[row + 1] comes from the next lane (adopt)
        if (DOT != 0) L.uu = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(u) + (uint32_t)rcl * (uint32_t)sizeof(T));
        const int shift = L.pa & 3;
        // dwords covering [pa, pa + nn) (<= CAP/4 + 1) — of a uniform block only the first row's codes: 1-9 dwords
        const int nd = max((shift + (uni ? L.ulen : L.nn) + 3) >> 2, 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            L.di[i] = min(lane + i * WAVE, nd - 1);
            L.wc[i] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(code) + (uint32_t)(L.pa - shift + 4 * L.di[i]));
        }
        if constexpr (!PAIR) {
            [[maybe_unused]] const int last = max(L.nn - 1, 0);
            if constexpr (WV) {
                const int vsh = L.pa & 1, g0 = (L.pa - vsh) >> 1, lastq = max(L.nn + vsh - 1, 0) >> 1;
                const V2d *val2 = reinterpret_cast<const V2d *>(val);
#pragma unroll
                for (int i = 0; i < ITEMS / 2; ++i) {
                    const int G = g0 + min(lane + i * WAVE, lastq);
                    const V2d q = *(G == g2_last ? tail2 : val2 + G);
                    L.vv[2 * i] = q.a; L.vv[2 * i + 1] = q.b;
                }
            } else {
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) L.vv[i] = val[L.pa + min(lane + i * WAVE, last)];
            }
        }
    };
    // Phase 2: the loads have landed — put the code bytes (and values) into this wavefront's LDS slice.
    // Called when the slice is free: before the first block and at the bottom of the loop, after the row
    // phase of the previous block has consumed it.
    auto stage = [&](const Loads &L) {
        const int shift = L.pa & 3;
        if (L.nn > 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) s_c[wv][L.di[i]] = L.wc[i];   // clamped duplicates store the same dword to the same slot
            if constexpr (CAP / 4 + 1 > 2 * WAVE) {
                if (L.ulen == 0 && ((shift + L.nn + 3) >> 2) > 2 * WAVE && lane == 0)     // the 129th dword exists only when shift + nn > 512
                    s_c[wv][2 * WAVE] = reinterpret_cast<const uint32_t *>(code + (L.pa - shift))[2 * WAVE];
            }
            if constexpr (!PAIR) {
                if constexpr (WV) {
                    const int tot = L.nn + (L.pa & 1);                  // window order: vs[k] = val[pa - (pa & 1) + k]
#pragma unroll
                    for (int i = 0; i < ITEMS / 2; ++i) {
                        const int k = 2 * (lane + i * WAVE);
                        if (k < tot) *reinterpret_cast<V2d *>(vs + k) = V2d{L.vv[2 * i], L.vv[2 * i + 1]};
                    }
                } else {
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) {
                    const int k = lane + i * WAVE;
                    if (k < L.nn) vs[k] = L.vv[i];
                }
                }
            }
        }
    };

    // loop-carried state of the block being processed: plain values, no load in flight behind them
    int c_ra = 0, c_rb = 0, c_s = 0, c_len = 0, c_shift = 0;
    [[maybe_unused]] int c_vsh = 0;       // WV: the staged values start this many entries into the wavefront's slice
    bool c_uni = false;      // scalar: every lane reads the FIRST row's codes
    T c_uu = szero<T>();
    auto adopt = [&](const Loads &L) {
        const int r = L.ra + lane;
        c_ra = L.ra; c_rb = L.rb; c_shift = L.pa & 3; c_uni = L.ulen > 0;
        if constexpr (WV) c_vsh = L.pa & 1;
        c_s = L.s - L.pa;
        int e = __shfl_down(L.s, 1, WAVE);          // next row's start; the block's last row ends at pa + nn
        if (r == L.rb - 1) e = L.pa + L.nn;
        c_len = r < L.rb ? e - L.s : 0;
        if (DOT != 0) c_uu = L.uu;
    };
    // Software pipeline, everything consumed one iteration after it was requested:
    //   top of iteration i:    issue loads of block i+1 (descriptor dn), descriptor of block i+2 (index o2),
    //                          schedule entry of block i+3
    //   middle:                row phase of block i — its x gather is the only exposed memory round trip
    //   bottom:                stage block i+1 into LDS, rotate dn <- dn2, o2 <- o3
    BlkDesc dn{0, 1, 0, 0};
    int o2 = 0;
    if (b < bend) {
        Loads first;
        issue(uniform(load_desc(__builtin_amdgcn_readfirstlane(block_index(b)))), first);
        if (b + bstep < bend) dn = uniform(load_desc(__builtin_amdgcn_readfirstlane(block_index(b + bstep))));
        if (b + 2 * bstep < bend) o2 = __builtin_amdgcn_readfirstlane(block_index(b + 2 * bstep));
        stage(first);
        adopt(first);
    }
    for (; b < bend; b += bstep) {
        const bool more = b + bstep < bend;
        Loads nxt;
        BlkDesc dn2{0, 1, 0, 0};
        int o3 = 0;
        if (b + 2 * bstep < bend) dn2 = load_desc(o2);
        if (b + 3 * bstep < bend) o3 = block_index(b + 3 * bstep);
        if (more) issue(dn, nxt);
        wave_lds_fence();
        // ---- one lane per row: mat.rs:100-105, fold(T::zero(), |acc, (col, val)| acc + x[col] * val)
        const int r = c_ra + lane;
        const uint32_t r8 = (uint32_t)r * (uint32_t)sizeof(T);     // byte offset of x[row]; launch checks ncols*sizeof(T) < 4 GiB
        const int s = c_s, len = c_len;
        T acc = szero<T>();
        for (int j0 = 0; __builtin_amdgcn_ballot_w64(j0 < len) != 0; j0 += 8) {
            // slots past a row's end read the (in-bounds, stale or zero) bytes behind it and are dropped below;
            // one clamp per chunk keeps the whole chunk inside the wavefront's slice
            const int kb = min(s + j0, CAP);
            const uint8_t *cp = cb + c_shift + (c_uni ? j0 : kb);    // uniform block: the first row's codes, at one address for all lanes
            T xg[8], av[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { xg[t] = szero<T>(); av[t] = szero<T>(); }
            // slots 4..7 are skipped (scalar branch) when no row of the block is that long: a 5-point row wastes
            // none of the LDS look-ups and gathers of slots 5..7, a 7-point row none of slot 7
            const uint64_t m4 = __builtin_amdgcn_ballot_w64(j0 + 4 < len), m5 = __builtin_amdgcn_ballot_w64(j0 + 5 < len),
                           m6 = __builtin_amdgcn_ballot_w64(j0 + 6 < len), m7 = __builtin_amdgcn_ballot_w64(j0 + 7 < len);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t == 4 && m4 == 0) break;
                if (t == 5 && m5 == 0) break;
                if (t == 6 && m6 == 0) break;
                if (t == 7 && m7 == 0) break;
                const bool valid = j0 + t < len;
                const int cd = cp[t];
                int off8;
                if constexpr (PAIR) { const PairEnt<T> e = s_pair[cd]; off8 = e.off8; av[t] = e.val; }
                else { off8 = s_off8[cd]; av[t] = vs[(WV ? c_vsh : 0) + kb + t]; }
                const uint32_t vo = valid ? r8 + (uint32_t)off8 : 0u;     // lanes past their row gather x[0] and drop it
                xg[t] = *reinterpret_cast<const T *>(xbytes + vo);
            }
            // all 8 gathers go out before the first product is formed (the scheduler otherwise hoists the
            // first multiply between them and with it a wait for the first gather: two round trips per block)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (j0 + t < len) acc = sadd(acc, smul(CONJX ? sconj(xg[t]) : xg[t], av[t]));
        }
        if (r < c_rb) {
            *reinterpret_cast<T *>(reinterpret_cast<char *>(y) + r8) = acc;
            if (DOT == 1) d0 = sadd(d0, smul(sconj(c_uu), acc));
            if (DOT == 2) { d0 = sadd(d0, smul(sconj(acc), acc)); d1 = sadd(d1, smul(sconj(acc), c_uu)); }
        }
        wave_lds_fence();   // the row phase is done with the LDS slice: refill it for the next block
        if (more) { stage(nxt); adopt(nxt); }
        dn = uniform(dn2); o2 = __builtin_amdgcn_readfirstlane(o3);
    }
}

template <class T, int DOT, bool CONJX, bool PAIR, bool WV = false>
__global__ __launch_bounds__(BLOCK) void spmv_dict_kernel(int n_rowblk, int xcd_chunk, const BlkDesc *__restrict__ desc,
                                                          const int32_t *__restrict__ order,
                                                          const int32_t *__restrict__ row_ptr,
                                                          const uint8_t *__restrict__ code,       // offset or pair codes
                                                          const int32_t *__restrict__ off_tab,    // per code: col - row
                                                          const T *__restrict__ val_tab,          // per pair code: value
                                                          const T *__restrict__ val, const T *__restrict__ x,
                                                          T *__restrict__ y, const T *__restrict__ u, T *__restrict__ part0,
                                                          T *__restrict__ part1, const int *__restrict__ status, const Fin fin,
                                                          const V2d *__restrict__ tail2, int g2_last) {
    static_assert(!WV || (sizeof(T) == 8 && !PAIR), "wide value loads: f64 offset-code stream");
    constexpr int CAP = nnz_cap<T>::value;          // nnz per row block (per wavefront)
    constexpr int CW = (CAP + 3 + CPAD + 3) / 4;    // dwords: CAP code bytes at any 4-byte phase + the readable pad
    __shared__ PairEnt<T> s_pair[PAIR ? TAB : 1];
    __shared__ int32_t s_off8[PAIR ? 1 : TAB];
    __shared__ uint32_t s_c[NWAVE][CW];
    __shared__ __attribute__((aligned(16))) T s_v[PAIR ? 1 : NWAVE][PAIR ? 1 : CAP + 16];
    __shared__ T red[NWAVE];
    // the solve's status word is requested first and looked at after the table loads: one memory round trip, not two
    // (a kernel of a finished solve must not store anything; it may load)
    const int run_state = status != nullptr ? *status : (int)ST_RUNNING;

    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    // the wavefront index as a SCALAR: the block walk (b, loop branches) then lives in SGPRs
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (PAIR) s_pair[tid] = PairEnt<T>{off_tab[tid] * (int32_t)sizeof(T), val_tab[tid]};    // BLOCK == TAB
    else s_off8[tid] = off_tab[tid] * (int32_t)sizeof(T);
    for (int i = lane; i < CW; i += WAVE) s_c[wv][i] = 0;       // the pad is read (and ignored) before it is ever written
    if constexpr (!PAIR) for (int i = lane; i < CAP + 16; i += WAVE) s_v[wv][i] = szero<T>();
    __syncthreads();                                // the only workgroup barrier: tables are read-only afterwards
    if (run_state != ST_RUNNING) return;

    T d0 = szero<T>(), d1 = szero<T>();
    dict_walk<T, DOT, CONJX, PAIR, WV>(n_rowblk, xcd_chunk, desc, order, row_ptr, code, val, x, y, u, tail2, g2_last, s_pair, s_off8, s_c,
                                       &s_v[0][0], PAIR ? 0 : CAP + 16, d0, d1);
    if (DOT >= 1) {
        d0 = block_sum(d0, red);
        if (tid == 0) st_partial(fin, part0 + blockIdx.x, d0);
    }
    if (DOT == 2) {
        d1 = block_sum(d1, red);
        if (tid == 0) st_partial(fin, part1 + blockIdx.x, d1);
    }
    if (DOT >= 1 && fin.counter) finalize_last_block<T, T>(fin, DOT == 2, red, red);
}

// ---------------------------------------------------------------------------------------------------------
// Two rows per lane (f64, pair codes).  A 64-lane 8-byte gather costs the CU's vector-memory pipe ~14.4 cycles
// whatever it touches — the same as a 16-byte one (scripts/micro/ta_rate.hip) — and the x gathers are most of the
// pipe's work in the kernel above.  Here lane l owns rows ra + 2l and ra + 2l + 1 of a 128-row block; where both
// rows have the same column offset in a slot (every interior row of a stencil) ONE 16-byte load returns x for
// both, as does one 16-byte load for u and one 16-byte store for y.  Rows whose slots disagree (grid boundaries,
// irregular rows) take an extra 8-byte gather for the second row, issued only if some lane of the wavefront needs
// it.  Same fold per row (left to right from zero): y stays bit-identical; the fused dot partials group the rows
// differently, so those reductions differ from the 64-row kernel's in summation order only.
struct alignas(16) D2 { double lo, hi; };
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
constexpr int CAP2 = 2 * nnz_cap<double>::value;          // 1024 code bytes per wide block
constexpr int CW2 = (CAP2 + 3 + CPAD + 15) / 16 * 4;      // dwords of a wavefront's slice (multiple of 4: b128 stores)

constexpr int UNI_OFF_MAXLEN = 32;          // offset-code stream: the pattern is read from the staged first row, chunk by chunk
constexpr int UNI2_MAXLEN = 8;              // ... of at most this many codes; the descriptor's nn then holds that length

// One wavefront per 128-row block: are all its rows copies of the first one (same length, same codes)?  Interior
// rows of a constant-coefficient stencil are; such a block needs neither its 1 KiB of codes nor row_ptr — the
// pattern is read once per block from the first row (spmv_pair2_kernel, uniform path).
//
// off_tab != nullptr (pair codes) and `triple`: the descriptor also names the CENTRE of a column triple (c - 1, c, c + 1) where the
// pattern has one — three consecutive slots whose offsets differ by one, as the i-neighbours of a stencil are in a
// row with sorted columns.  The kernel reads x for all three from the centre slot's 16-byte loads (nn bits 8..15 =
// the centre's slot index, 0 = none; bits 0..7 = the row length).
__global__ __launch_bounds__(BLOCK) void mark_uniform_kernel(int n_wide, BlkDesc *__restrict__ desc,
                                                             const int32_t *__restrict__ row_ptr,
                                                             const uint8_t *__restrict__ code, int maxlen,
                                                             const int32_t *__restrict__ off_tab, int triple, int seam,
                                                             int ncols) {
    const int lane = threadIdx.x & (WAVE - 1);
    for (int b = blockIdx.x * NWAVE + (threadIdx.x >> 6); b < n_wide; b += gridDim.x * NWAVE) {
        const BlkDesc d = desc[b];
        const int nr = d.rb - d.ra;
        const int L0 = row_ptr[d.ra + 1] - row_ptr[d.ra];
        bool ok = nr >= 1 && L0 >= 1 && L0 <= maxlen && d.nn == nr * L0;
        if (ok) {
            for (int r = d.ra + lane; r < d.rb; r += WAVE) {
                const int s = row_ptr[r];
                ok = ok && s == d.pa + (r - d.ra) * L0;             // with nn == nr * L0: every row has L0 entries
                for (int j = 0; ok && j < L0; ++j) ok = code[s + j] == code[d.pa + j];
            }
        }
        const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
        if (all_ok && lane == 0) {
            int centre = 0;
            if (off_tab != nullptr && triple != 0)
                for (int j = 1; j + 1 < L0 && centre == 0; ++j) {
                    const int o = off_tab[code[d.pa + j]];
                    if (off_tab[code[d.pa + j - 1]] == o - 1 && off_tab[code[d.pa + j + 1]] == o + 1) centre = j;
                }
            desc[b] = BlkDesc{d.ra, (int32_t)((uint32_t)d.rb | UNI2), d.pa, L0 | (centre << 8)};
        }
        if (all_ok || off_tab == nullptr || seam == 0 || nr != 2 * WAVE) continue;
        // ---- SEAM blocks (pair codes, full 128-row blocks): every row repeats one pattern A (the longest row's) except
        // one row, or two adjacent ones, which hold
        //   * a subsequence of A's codes (the x = nx - 1 | x = 0 seam of a truncated stencil: one slot missing), or
        //   * a single entry on one of A's offsets with a value of its own (a Dirichlet / identity row inside a stencil).
        // Such a block runs the uniform path: all rows gather with A's offsets (checked in bounds for every row here, since
        // that path does not clamp), the exceptional rows fold only the slots they have, with their own value where
        // they have one.  Same entries, same order per row.  Encoding (rb is redundant for a full block: ra + 128):
        //   nn  bits 0..7 length of A, 8..15 centre slot, 16..22 first exceptional row k, 23..30 its slot mask
        //   rb  bit 30 UNI2, bit 29 SEAM, 0..7 slot mask of row k + 1 (0xff..: regular), 8..15 / 16..23 the value-bearing
        //       pair code of row k / k + 1, bit 24 / 25: that row takes its value from that code
        int len[2], start[2];
        int Lmax = 0;
        for (int h = 0; h < 2; ++h) {
            const int r = d.ra + lane + h * WAVE;
            start[h] = row_ptr[r]; len[h] = row_ptr[r + 1] - start[h];
            Lmax = max(Lmax, len[h]);
        }
        for (int o = 32; o > 0; o >>= 1) Lmax = max(Lmax, __shfl_xor(Lmax, o, WAVE));
        if (Lmax < 2 || Lmax > 8) continue;                                       // (wave-uniform)
        const uint64_t f0 = __builtin_amdgcn_ballot_w64(len[0] == Lmax), f1 = __builtin_amdgcn_ballot_w64(len[1] == Lmax);
        const int prow = f0 ? (int)__builtin_ctzll(f0) : WAVE + (int)__builtin_ctzll(f1);     // the first row of full length
        const int pA = row_ptr[d.ra + prow];
        uint8_t A[8];
        int omin = 0x7fffffff, omax = -0x7fffffff;
        for (int j = 0; j < 8; ++j) A[j] = j < Lmax ? code[pA + j] : (uint8_t)0;
        for (int j = 0; j < Lmax; ++j) { const int o = off_tab[A[j]]; omin = min(omin, o); omax = max(omax, o); }
        const int full = (1 << Lmax) - 1;
        int mask[2], ovc[2];           // slots the row has; its own value-bearing code or -1
        bool bad = false;
        for (int h = 0; h < 2; ++h) {
            mask[h] = 0; ovc[h] = -1;
            int j = 0;
            bool sub = true;
            for (int i = 0; i < len[h] && sub; ++i) {
                const uint8_t cd = code[start[h] + i];
                while (j < Lmax && A[j] != cd) ++j;
                if (j == Lmax) sub = false; else { mask[h] |= 1 << j; ++j; }
            }
            if (sub) continue;
            mask[h] = 0;
            if (len[h] == 1) {
                const int cd = code[start[h]], o = off_tab[cd];
                for (int q = 0; q < Lmax && mask[h] == 0; ++q) if (off_tab[A[q]] == o) mask[h] = 1 << q;
                if (mask[h] != 0) ovc[h] = cd; else bad = true;
            } else bad = true;
        }
        if (__builtin_amdgcn_ballot_w64(bad) != 0) continue;
        if ((int64_t)d.ra + omin < 0 || (int64_t)d.rb - 1 + omax > (int64_t)ncols - 1) continue;
        const uint64_t e0 = __builtin_amdgcn_ballot_w64(mask[0] != full), e1 = __builtin_amdgcn_ballot_w64(mask[1] != full);
        const int ne = __builtin_popcountll(e0) + __builtin_popcountll(e1);
        if (ne < 1 || ne > 2) continue;
        const int k = e0 ? (int)__builtin_ctzll(e0) : WAVE + (int)__builtin_ctzll(e1);     // first exceptional row (local index)
        const int k1 = k + 1;
        if (ne == 2 && k1 >= 2 * WAVE) continue;
        // masks and codes of rows k and k + 1, fetched from the lanes that hold them
        const int mA = __shfl(k < WAVE ? mask[0] : mask[1], k & (WAVE - 1), WAVE);
        const int cA = __shfl(k < WAVE ? ovc[0] : ovc[1], k & (WAVE - 1), WAVE);
        int mB = full, cB = -1;
        if (k1 < 2 * WAVE) {
            mB = __shfl(k1 < WAVE ? mask[0] : mask[1], k1 & (WAVE - 1), WAVE);
            cB = __shfl(k1 < WAVE ? ovc[0] : ovc[1], k1 & (WAVE - 1), WAVE);
        }
        if (ne == 2 && mB == full) continue;                                      // two exceptional rows that are not adjacent
        if (lane == 0) {
            int centre = 0;
            for (int j = 1; triple != 0 && j + 1 < Lmax && centre == 0; ++j) {
                const int o = off_tab[A[j]];
                if (off_tab[A[j - 1]] == o - 1 && off_tab[A[j + 1]] == o + 1) centre = j;
            }
            // rows that are regular in every slot A has keep the mask 0xff (slots >= Lmax are never folded)
            const uint32_t mB8 = mB == full ? 0xffu : (uint32_t)mB;
            const uint32_t rbw = UNI2 | SEAM2 | mB8 | ((uint32_t)(cA < 0 ? 0 : cA) << 8) | ((uint32_t)(cB < 0 ? 0 : cB) << 16) |
                                 (cA >= 0 ? 1u << 24 : 0u) | (cB >= 0 ? 1u << 25 : 0u);
            desc[b] = BlkDesc{d.ra, (int32_t)rbw, pA, Lmax | (centre << 8) | (k << 16) | (mA << 23)};
        }
    }
}

// FULL uniform block of the two-rows-per-lane kernels (every lane has both rows — the interior of a stencil): no row
// masks, and no clamp either: the second row's column r0 + 1 + off is valid, so the 16-byte load at r0 + off stays
// inside x.  Offset and value of a slot are wave-uniform: made scalars, the gather is SGPR base + lane offset and the
// products take the value from SGPRs — 4 vector ALU instructions per slot instead of ~25 (the kernel ran at 43 % VALU
// utilisation, profiles/r02_tuning.md §9).
//
// UL, SC > 0 (compile time): the pattern has UL slots and a column triple (c - 1, c, c + 1) in slots SC - 1, SC, SC + 1
// (mark_uniform_kernel).  The outer two are not loaded: lane l's x[r0 - 1 + o] is lane l - 1's second half of the
// centre pair, x[r0 + 2 + o] lane l + 1's first half (wavefront shifts); the two ends of the block come from two
// scalar loads.  UL - 2 vector loads instead of UL, all issued in straight-line code.
// UL == 0: run-time length `ulen`, every slot loaded.  after_loads() runs between the last load and the first product.
__device__ __forceinline__ double wave_shift_up(double prev_for_lane0, double v) {       // lane l <- lane l - 1
    const long long o = __double_as_longlong(prev_for_lane0), q = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)q, 0x138, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32), (int)(uint32_t)(q >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double wave_shift_down(double next_for_last_lane, double v) { // lane l <- lane l + 1
    const long long o = __double_as_longlong(next_for_last_lane), q = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)q, 0x130, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32), (int)(uint32_t)(q >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
template <int UL, int SC, class AfterLoads>
__device__ __forceinline__ void full_uniform_block(const PairEnt<double> *s_pair, uint64_t pat, int ulen, bool seam, int seam1, int seam2, const char *xbytes,
                                                   uint32_t r8, uint32_t ra8, int lane, AfterLoads &&after_loads,
                                                   double &acc0, double &acc1) {
    using T = double;
    T pl[8], ph[8], av[8];
    int off8c = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) { pl[t] = 0.0; ph[t] = 0.0; av[t] = 0.0; }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (UL == 0 ? t >= ulen : t >= UL) break;
        const PairEnt<T> e = s_pair[(int)((pat >> (8 * t)) & 255u)];
        const int off8 = __builtin_amdgcn_readfirstlane(e.off8);
        const uint32_t vlo = __builtin_amdgcn_readfirstlane((int)(uint32_t)__double_as_longlong(e.val));
        const uint32_t vhi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(__double_as_longlong(e.val) >> 32));
        av[t] = __longlong_as_double((long long)(((uint64_t)vhi << 32) | vlo));
        if (SC > 0 && (t == SC - 1 || t == SC + 1)) continue;
        const D2 px = *reinterpret_cast<const D2 *>(xbytes + (int64_t)off8 + r8);
        pl[t] = px.lo; ph[t] = px.hi;
        if (SC > 0 && t == SC) off8c = off8;
    }
    // the two ends of the block, x[ra - 1 + o] and x[ra + 128 + o]: wave-uniform addresses, read through the SCALAR
    // cache after the last LDS read of the block (scalar loads return out of order and share the LDS counter) — a
    // 64-lane load of them would cost the vector-memory pipe as much as a gather
    T e_lo = 0.0, e_hi = 0.0;
    if (SC > 0) {
        const T *xe = reinterpret_cast<const T *>(xbytes + (int64_t)off8c + ra8);
        e_lo = xe[-1]; e_hi = xe[2 * WAVE];
    }
    after_loads();
    __builtin_amdgcn_sched_barrier(0);
    if (SC > 0) {
        constexpr int C = SC > 0 ? SC : 1;
        const T left = wave_shift_up(e_lo, ph[C]), right = wave_shift_down(e_hi, pl[C]);
        pl[C - 1] = left; ph[C - 1] = pl[C];
        pl[C + 1] = ph[C]; ph[C + 1] = right;
    }
    if (!seam) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (UL == 0 ? t >= ulen : t >= UL) break;
            acc0 = acc0 + pl[t] * av[t];
            acc1 = acc1 + ph[t] * av[t];
        }
    } else {
        // seam block: local rows k and k + 1 fold only the slots of their masks, with their own value where they carry one
        // (a row with a value of its own has exactly one slot)
        const int k = seam1 & 127, maskA = (seam1 >> 7) & 255, maskB = seam2 & 255;
        const T valA = s_pair[(seam2 >> 8) & 255].val, valB = s_pair[(seam2 >> 16) & 255].val;
        const bool ovA = ((seam2 >> 24) & 1) != 0, ovB = ((seam2 >> 25) & 1) != 0;
        const bool a0 = 2 * lane == k, a1 = 2 * lane + 1 == k, b0 = 2 * lane == k + 1, b1 = 2 * lane + 1 == k + 1;
        const int pm0 = a0 ? maskA : (b0 ? maskB : 255), pm1 = a1 ? maskA : (b1 ? maskB : 255);
        const bool o0 = (a0 && ovA) || (b0 && ovB), o1 = (a1 && ovA) || (b1 && ovB);
        const T v0 = a0 ? valA : valB, v1 = a1 ? valA : valB;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (UL == 0 ? t >= ulen : t >= UL) break;
            const T n0 = acc0 + pl[t] * (o0 ? v0 : av[t]), n1 = acc1 + ph[t] * (o1 ? v1 : av[t]);
            acc0 = ((pm0 >> t) & 1) ? n0 : acc0;
            acc1 = ((pm1 >> t) & 1) ? n1 : acc1;
        }
    }
}
template <class AfterLoads>
__device__ __forceinline__ void full_uniform_dispatch(const PairEnt<double> *s_pair, uint64_t pat, int ulen, int sc, bool seam, int seam1, int seam2, const char *xbytes,
                                                      uint32_t r8, uint32_t ra8, int lane, AfterLoads &&after_loads,
                                                      double &acc0, double &acc1) {
    // (scalar branches) the stencils: 7-point 3-D, 5-point 2-D, 3-point 1-D with sorted columns; anything else generic
    if (ulen == 7 && sc == 3) full_uniform_block<7, 3>(s_pair, pat, ulen, seam, seam1, seam2, xbytes, r8, ra8, lane, after_loads, acc0, acc1);
    else if (ulen == 5 && sc == 2) full_uniform_block<5, 2>(s_pair, pat, ulen, seam, seam1, seam2, xbytes, r8, ra8, lane, after_loads, acc0, acc1);
    else if (ulen == 3 && sc == 1) full_uniform_block<3, 1>(s_pair, pat, ulen, seam, seam1, seam2, xbytes, r8, ra8, lane, after_loads, acc0, acc1);
    else full_uniform_block<0, 0>(s_pair, pat, ulen, seam, seam1, seam2, xbytes, r8, ra8, lane, after_loads, acc0, acc1);
}


struct Blk2Loads {
    int ra, rb, pa, nn;      // descriptor of the 128-row block
    bool uni; int ulen;      // uniform block (every row = the first row's ulen codes)
    int tri;                 // ... and the slot of its column triple's centre (0: none)
    bool is_seam; int seam1, seam2;   // ... or uniform but for one or two rows (mark_uniform_kernel's encoding: nn >> 16, rb's low bits)
    int a, b;                // row_ptr[i0], row_ptr[i0 + 1], i0 = min(ra + 2 lane, rb - 1)
    double u0, u1;           // dot operands of the lane's two rows
    u4v wc;                  // 16 code bytes
    int di;                  // ... and the b128 slot they go to
};

// The walk of the two-rows-per-lane kernels over `n_wide` 128-row blocks (positions of `order`, or natural order): a
// wavefront takes every (gridDim.x * NWAVE)-th position, the next block's loads are issued before this block's products.
// Shared by spmv_pair2_kernel (the whole matrix) and spmv_tile_kernel (the blocks outside its tiles).  d0 / d1: the
// lane's running dot partials (DOT as in launch_spmv).  s_pair: the staged (byte offset, value) table; s_c: NWAVE
// zero-initialised code slices.
// YNT: y is written with non-temporal stores (HBM-sized vectors: the result is not read again before it has been evicted)
template <int DOT, bool YNT>
__device__ __forceinline__ void pair2_walk(int n_wide, int xcd_chunk, const BlkDesc *__restrict__ desc,
                                           const int32_t *__restrict__ order, const int32_t *__restrict__ row_ptr,
                                           const uint8_t *__restrict__ code, const double *__restrict__ x,
                                           double *__restrict__ y, const double *__restrict__ u, int nrows, int ncols,
                                           const PairEnt<double> *s_pair, uint32_t (*s_c)[CW2], double &d0, double &d1) {
    using T = double;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint8_t *cb = reinterpret_cast<const uint8_t *>(s_c[wv]);
    const char *xbytes = reinterpret_cast<const char *>(x);
    const uint32_t xlast_pair = (uint32_t)(ncols - 2) * 8u;                    // last byte offset a 16-byte x load may start at

    int b, bstep, bend;
    if (xcd_chunk) {
        const int chunk = (n_wide + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        b = xcd * chunk + (blockIdx.x >> 3) * NWAVE + wv;
        bstep = (gridDim.x >> 3) * NWAVE;
        bend = min(n_wide, (xcd + 1) * chunk);
    } else {
        b = blockIdx.x * NWAVE + wv; bstep = gridDim.x * NWAVE; bend = n_wide;
    }
    if (b >= bend) b = bend;

    typedef const int32_t __attribute__((address_space(1))) *gi32_p;
    uintptr_t order_a = reinterpret_cast<uintptr_t>(order);
    asm volatile("" : "+v"(order_a));               // vector (in-order, 4 bytes per lane) loads of the walk order
    const gi32_p order_v = reinterpret_cast<gi32_p>(order_a);
    // descriptors: wave-uniform index, read through the scalar cache (`desc` must stay un-captured for that: the
    // compiler only uses scalar loads on memory it can prove nothing in the kernel writes)
    auto load_desc = [&](int bi) -> BlkDesc { return desc[bi]; };
    auto block_index = [&](int bi) -> int { return order ? order_v[bi] : bi; };
    auto uniform = [&](const BlkDesc &d) -> BlkDesc {
        return BlkDesc{__builtin_amdgcn_readfirstlane(d.ra), __builtin_amdgcn_readfirstlane(d.rb),
                       __builtin_amdgcn_readfirstlane(d.pa), __builtin_amdgcn_readfirstlane(d.nn)};
    };
    auto issue = [&](const BlkDesc &d, Blk2Loads &L) {
        L.uni = ((uint32_t)d.rb & UNI2) != 0;                                   // scalar: all rows share one code sequence of d.nn codes
        L.is_seam = L.uni && ((uint32_t)d.rb & SEAM2) != 0;                     // ... but for one or two of them
        L.ra = d.ra; L.rb = L.is_seam ? d.ra + 2 * WAVE : (int)((uint32_t)d.rb & ~UNI2); L.pa = d.pa;
        L.ulen = L.uni ? (d.nn & 0xff) : 0;
        L.tri = L.uni ? ((d.nn >> 8) & 0xff) : 0;
        L.seam1 = L.is_seam ? (d.nn >> 16) : 0;
        L.seam2 = L.is_seam ? (int)((uint32_t)d.rb & 0x3ffffffu) : 0;
        L.nn = L.uni ? L.ulen * (L.rb - L.ra) : d.nn;
        const int r0 = L.ra + 2 * lane;
        L.a = 0; L.b = 0;
        if (!L.uni) {                                                           // a uniform block needs no row_ptr
            const int i0 = min(r0, L.rb - 1);                                   // row_ptr[i0 + 1] exists: i0 + 1 <= rb <= nrows
            const int2 ab = *reinterpret_cast<const int2 *>(reinterpret_cast<const char *>(row_ptr) + (uint32_t)i0 * 4u);
            L.a = ab.x; L.b = ab.y;
        }
        if (DOT != 0) {
            const int p0 = min(r0, nrows - 2);                                  // the pair (u[p0], u[p0 + 1]) is inside u
            const D2 uu = *reinterpret_cast<const D2 *>(reinterpret_cast<const char *>(u) + (uint32_t)p0 * 8u);
            L.u0 = r0 == p0 ? uu.lo : uu.hi;                                    // r0 == nrows - 1: its operand is the pair's second half
            L.u1 = uu.hi;
        }
        const int shift = L.pa & 3;
        const int nq = L.uni ? 1 : max((shift + L.nn + 15) >> 4, 1);            // 16-byte pieces covering the codes, <= 65 (uniform: the first row's only)
        L.di = min(lane, nq - 1);
        L.wc = *reinterpret_cast<const u4v *>(reinterpret_cast<const char *>(code) + (uint32_t)(L.pa - shift + 16 * L.di));
    };
    int c_ra = 0, c_rb = 0, c_shift = 0, c_s0 = 0, c_s1 = 0, c_len0 = 0, c_len1 = 0;
    bool c_uni = false;
    uint64_t c_pat = 0;      // uniform block: its (at most 8) codes, first code in the low byte
    int c_tri = 0;           // ... and the centre slot of its column triple
    bool c_seam = false;     // ... or uniform but for one or two rows:
    int c_seam1 = 0, c_seam2 = 0;
    T c_u0 = 0.0, c_u1 = 0.0;
    auto stage = [&](const Blk2Loads &L) {
        const int shift = L.pa & 3;
        c_uni = L.uni;
        if (L.uni) {
            // every lane holds the same 16 bytes [pa - shift, pa - shift + 16): the pattern starts `shift` bytes in
            const uint64_t lo = (uint64_t)__builtin_amdgcn_readfirstlane(L.wc.x) | ((uint64_t)__builtin_amdgcn_readfirstlane(L.wc.y) << 32);
            const uint64_t hi = (uint64_t)__builtin_amdgcn_readfirstlane(L.wc.z);
            c_pat = shift ? (lo >> (8 * shift)) | (hi << (64 - 8 * shift)) : lo;
            const int r0 = L.ra + 2 * lane;
            c_ra = L.ra; c_rb = L.rb; c_shift = shift; c_tri = L.tri; c_seam = L.is_seam; c_seam1 = L.seam1; c_seam2 = L.seam2;
            c_s0 = 0; c_s1 = 0;
            c_len0 = r0 < L.rb ? L.ulen : 0;
            c_len1 = r0 + 1 < L.rb ? L.ulen : 0;
            if (DOT != 0) { c_u0 = L.u0; c_u1 = L.u1; }
            return;
        }
        if (L.nn > 0) {
            *reinterpret_cast<u4v *>(&s_c[wv][4 * L.di]) = L.wc;               // clamped duplicates store the same 16 bytes
            if (shift + L.nn > CAP2 && lane == 0)                               // the 65th piece exists only then: one dword is enough
                s_c[wv][CAP2 / 4] = *reinterpret_cast<const uint32_t *>(code + (L.pa - shift) + CAP2);
        }
        const int r0 = L.ra + 2 * lane;
        int c = __shfl_down(L.a, 1, WAVE);                                      // row_ptr[r0 + 2] sits in the next lane
        if (r0 + 2 >= L.rb) c = L.pa + L.nn;                                    // ... unless the block ends there
        c_ra = L.ra; c_rb = L.rb; c_shift = shift;
        c_s0 = L.a - L.pa; c_s1 = L.b - L.pa;
        c_len0 = r0 < L.rb ? L.b - L.a : 0;
        c_len1 = r0 + 1 < L.rb ? c - L.b : 0;
        if (DOT != 0) { c_u0 = L.u0; c_u1 = L.u1; }
    };

    BlkDesc dn{0, 1, 0, 0};
    int o2 = 0;
    if (b < bend) {
        Blk2Loads first;
        issue(uniform(load_desc(__builtin_amdgcn_readfirstlane(block_index(b)))), first);
        if (b + bstep < bend) dn = uniform(load_desc(__builtin_amdgcn_readfirstlane(block_index(b + bstep))));
        if (b + 2 * bstep < bend) o2 = __builtin_amdgcn_readfirstlane(block_index(b + 2 * bstep));
        stage(first);
    }
    for (; b < bend; b += bstep) {
        const bool more = b + bstep < bend;
        Blk2Loads nxt;
        BlkDesc dn2{0, 1, 0, 0};
        int o3 = 0;
        if (b + 3 * bstep < bend) o3 = block_index(b + 3 * bstep);
        if (more) issue(dn, nxt);
        // The descriptor of the block after next comes through the SCALAR cache (its index is wave-uniform): a 64-lane
        // 16-byte load of it costs the vector-memory pipe as much as a gather.  Scalar loads return out of order and
        // share their counter with the LDS, so it is requested after the block's gathers (and the LDS reads that
        // address them) are out, and looked at when the block is done.
        const bool want_dn2 = b + 2 * bstep < bend;
        auto after_gathers = [&]() { if (want_dn2) dn2 = load_desc(o2); };
        wave_lds_fence();
        const int r0 = c_ra + 2 * lane;
        const uint32_t r8 = (uint32_t)r0 * 8u;
        const int len0 = c_len0, len1 = c_len1, lenm = max(len0, len1);
        T acc0 = 0.0, acc1 = 0.0;
        if (c_uni) {
            // ---- uniform block: the pattern is scalar.  Per slot one LDS read of {offset, value} at a wave-uniform
            // address, one 16-byte gather for the lane's two rows, two multiply-adds; no codes, no row_ptr.
            const int ulen = __builtin_amdgcn_readfirstlane(lenm);             // == the block's row length (lane 0 always has a row)
            T pl[8], ph[8], av[8];
            uint32_t hi_bits = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) { pl[t] = 0.0; ph[t] = 0.0; av[t] = 0.0; }
            if (c_rb - c_ra == 2 * WAVE) {
                full_uniform_dispatch(s_pair, c_pat, ulen, c_tri, c_seam, c_seam1, c_seam2, xbytes, r8, (uint32_t)c_ra * 8u, lane, after_gathers, acc0, acc1);
            } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t >= ulen) break;
                const PairEnt<T> e = s_pair[(int)((c_pat >> (8 * t)) & 255u)];
                av[t] = e.val;
                const uint32_t vo0 = len0 > 0 ? r8 + (uint32_t)e.off8 : 0u;
                const uint32_t vp = min(vo0, xlast_pair);                       // only a single-row lane at the matrix end is ever clamped
                hi_bits |= (vo0 != vp ? 1u : 0u) << t;
                const D2 px = *reinterpret_cast<const D2 *>(xbytes + vp);
                pl[t] = px.lo; ph[t] = px.hi;
            }
            after_gathers();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t >= ulen) break;
                if (len0 > 0) acc0 = acc0 + (((hi_bits >> t) & 1u) ? ph[t] : pl[t]) * av[t];
                if (len1 > 0) acc1 = acc1 + ph[t] * av[t];
            }
            }
        } else {
        if (__builtin_amdgcn_ballot_w64(0 < lenm) == 0) after_gathers();        // (a block of empty rows)
        for (int j0 = 0; __builtin_amdgcn_ballot_w64(j0 < lenm) != 0; j0 += 8) {
            const uint8_t *cp0 = cb + c_shift + min(c_s0 + j0, CAP2);
            const uint8_t *cp1 = cb + c_shift + min(c_s1 + j0, CAP2);
            T pl[8], ph[8];           // the 16-byte gather of the slot: x[col0], x[col0 + 1]
            T xs[8];                  // row 1's own gather where its column is not row 0's + 1
            uint32_t same_bits = 0, hi_bits = 0;      // per slot: row 1 shares the gather / row 0's x is the pair's second half
#pragma unroll
            for (int t = 0; t < 8; ++t) { pl[t] = 0.0; ph[t] = 0.0; xs[t] = 0.0; }
            const uint64_t m4 = __builtin_amdgcn_ballot_w64(j0 + 4 < lenm), m5 = __builtin_amdgcn_ballot_w64(j0 + 5 < lenm),
                           m6 = __builtin_amdgcn_ballot_w64(j0 + 6 < lenm), m7 = __builtin_amdgcn_ballot_w64(j0 + 7 < lenm);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t == 4 && m4 == 0) break;
                if (t == 5 && m5 == 0) break;
                if (t == 6 && m6 == 0) break;
                if (t == 7 && m7 == 0) break;
                const bool v0 = j0 + t < len0, v1 = j0 + t < len1;
                const int off0 = s_pair[cp0[t]].off8, off1 = s_pair[cp1[t]].off8;
                const bool same = v0 && v1 && off0 == off1;                     // column of row 1 == column of row 0 + 1
                const uint32_t vo0 = v0 ? r8 + (uint32_t)off0 : 0u;             // byte offset of x[col0]; unused rows read x[0]
                const uint32_t vp = min(vo0, xlast_pair);                       // a 16-byte load must start at or before x[ncols - 2]
                same_bits |= (same ? 1u : 0u) << t;
                hi_bits |= (vo0 != vp ? 1u : 0u) << t;                          // col0 == ncols - 1: it is the pair's second half
                const D2 px = *reinterpret_cast<const D2 *>(xbytes + vp);
                pl[t] = px.lo; ph[t] = px.hi;
                const bool need1 = v1 && !same;
                if (__builtin_amdgcn_ballot_w64(need1) != 0)                    // scalar branch: interior stencil blocks skip it
                    xs[t] = *reinterpret_cast<const T *>(xbytes + (need1 ? r8 + 8u + (uint32_t)off1 : 0u));
            }
            if (j0 == 0) after_gathers();
            __builtin_amdgcn_sched_barrier(0);                                  // every gather out before the first product
            asm volatile("" ::: "memory");      // the values are looked up again below rather than held in 32 registers across the wait
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (j0 + t < len0) acc0 = acc0 + (((hi_bits >> t) & 1u) ? ph[t] : pl[t]) * s_pair[cp0[t]].val;
                if (j0 + t < len1) acc1 = acc1 + (((same_bits >> t) & 1u) ? ph[t] : xs[t]) * s_pair[cp1[t]].val;
            }
        }
        }
        if (r0 + 1 < c_rb) {
            if constexpr (YNT) {
                const D2 yy{acc0, acc1};
                u4v q;
                __builtin_memcpy(&q, &yy, 16);
                __builtin_nontemporal_store(q, reinterpret_cast<u4v *>(reinterpret_cast<char *>(y) + r8));
            } else {
                *reinterpret_cast<D2 *>(reinterpret_cast<char *>(y) + r8) = D2{acc0, acc1};
            }
            if (DOT == 1) { d0 = d0 + c_u0 * acc0; d0 = d0 + c_u1 * acc1; }
            if (DOT == 2) { d0 = d0 + acc0 * acc0; d1 = d1 + acc0 * c_u0; d0 = d0 + acc1 * acc1; d1 = d1 + acc1 * c_u1; }
        } else if (r0 < c_rb) {
            *reinterpret_cast<T *>(reinterpret_cast<char *>(y) + r8) = acc0;
            if (DOT == 1) d0 = d0 + c_u0 * acc0;
            if (DOT == 2) { d0 = d0 + acc0 * acc0; d1 = d1 + acc0 * c_u0; }
        }
        wave_lds_fence();
        if (more) stage(nxt);
        dn = uniform(dn2); o2 = __builtin_amdgcn_readfirstlane(o3);
    }
}

template <int DOT, bool YNT>
__global__ __launch_bounds__(BLOCK) void spmv_pair2_kernel(int n_wide, int xcd_chunk, const BlkDesc *__restrict__ desc,
                                                           const int32_t *__restrict__ order,
                                                           const int32_t *__restrict__ row_ptr,
                                                           const uint8_t *__restrict__ code,
                                                           const int32_t *__restrict__ off_tab,
                                                           const double *__restrict__ val_tab, const double *__restrict__ x,
                                                           double *__restrict__ y, const double *__restrict__ u,
                                                           double *__restrict__ part0, double *__restrict__ part1,
                                                           const int *__restrict__ status, int nrows, int ncols, const Fin fin) {
    using T = double;
    __shared__ PairEnt<T> s_pair[TAB];
    __shared__ __attribute__((aligned(16))) uint32_t s_c[NWAVE][CW2];
    __shared__ T red[NWAVE];
    const int run_state = status != nullptr ? *status : (int)ST_RUNNING;       // looked at after the table loads (one round trip)

    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    s_pair[tid] = PairEnt<T>{off_tab[tid] * 8, val_tab[tid]};                  // BLOCK == TAB
    for (int i = lane; i < CW2; i += WAVE) s_c[wv][i] = 0;                     // the pad is read (and ignored) before it is written
    __syncthreads();
    if (run_state != ST_RUNNING) return;

    T d0 = 0.0, d1 = 0.0;
    pair2_walk<DOT, YNT>(n_wide, xcd_chunk, desc, order, row_ptr, code, x, y, u, nrows, ncols, s_pair, s_c, d0, d1);
    if (DOT >= 1) {
        d0 = block_sum(d0, red);
        if (tid == 0) st_partial(fin, part0 + blockIdx.x, d0);
    }
    if (DOT == 2) {
        d1 = block_sum(d1, red);
        if (tid == 0) st_partial(fin, part1 + blockIdx.x, d1);
    }
    if (DOT >= 1 && fin.counter) finalize_last_block<T, T>(fin, DOT == 2, red, red);
}

// ---- LDS x-window tiles (knob "spmv_tile"; profiles/r03_tuning.md §8) -------------------------------------------------
// The per-block walk above pulls every column window of a 128-row block through the vector L1 on its own: five 16-byte
// loads per lane for a 7-point stencil, and the texture-address unit / L1 miss path is what the kernel waits for
// (TA busy 78 %, 0.45 of the HBM rate).  Consecutive row blocks of a stencil overlap in all their NEAR windows: a
// workgroup that owns TILE_ROWS consecutive rows needs x[ts - W, ts + TILE_ROWS + W) ONCE for every column within W of the
// diagonal — (T + 2W) / T = 1.25 loads per lane and 128 rows instead of one per near window — and only the FAR windows
// (the +-plane neighbours) one by one.  So: a tile = TILE_B consecutive FULL uniform 128-row blocks (plain or seam,
// mark_uniform_kernel) that share one pattern whose slots are, in row order, FL far slots, UL - FL - FH near slots, FH
// far slots (sorted columns give exactly that).  The workgroup stages the window in LDS with 16-byte loads, issues the far
// pair loads of all its rows, and every lane folds its two rows' slots left to right — the same products in the same order
// as full_uniform_block, x taken from LDS for the near slots: y bit-identical.  Tiles are dealt to the XCDs by their
// phase within the far period (tile_plan below) so that a far window was some tile's near window on the same L2.  The
// 128-row blocks outside the tiles (boundary planes, the tiles a boundary line cuts, the matrix ends) are walked by the
// same launch afterwards (pair2_walk), so the launch writes all of y and one partial per workgroup.
constexpr int TILE_ROWS = 4096, TILE_W = 512, TILE_W_WIDE = 1536, TILE_B = TILE_ROWS / (2 * WAVE);
struct TilePat { int32_t off[8]; double val[8]; };

// one thread per candidate tile (blocks [t TILE_B, (t + 1) TILE_B)): its pattern (up to 8 codes, low byte first) and
// length, or length 0 when the blocks are not TILE_B consecutive full uniform blocks of one pattern
__global__ __launch_bounds__(BLOCK) void tile_mark_kernel(int n_cand, const BlkDesc *__restrict__ desc, const uint8_t *__restrict__ code,
                                                          unsigned long long *__restrict__ pat_out, int *__restrict__ len_out) {
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_cand) return;
    unsigned long long pat0 = 0;
    int L0 = 0, ra0 = 0;
    bool ok = true;
    for (int i = 0; i < TILE_B && ok; ++i) {
        const BlkDesc d = desc[t * TILE_B + i];
        const uint32_t rb = (uint32_t)d.rb;
        ok = (rb & UNI2) != 0;
        if (!ok) break;
        const bool seam = (rb & SEAM2) != 0;
        const int L = d.nn & 0xff;
        const int nr = seam ? 2 * WAVE : (int)(rb & ~UNI2) - d.ra;
        unsigned long long pat = 0;
        for (int j = 0; j < L && j < 8; ++j) pat |= (unsigned long long)code[d.pa + j] << (8 * j);
        if (i == 0) { pat0 = pat; L0 = L; ra0 = d.ra; }
        ok = nr == 2 * WAVE && L == L0 && L >= 1 && L <= 8 && pat == pat0 && d.ra == ra0 + i * 2 * WAVE;
    }
    pat_out[t] = ok ? pat0 : 0ull;
    len_out[t] = ok ? L0 : 0;
}

// one thread per 128-row block: 1 = a full uniform block (plain or seam) of exactly the pattern (pat, L)
__global__ __launch_bounds__(BLOCK) void tile_flag_kernel(int n_wide, const BlkDesc *__restrict__ desc, const uint8_t *__restrict__ code,
                                                          unsigned long long pat, int L, uint8_t *__restrict__ flag) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n_wide) return;
    const BlkDesc d = desc[j];
    const uint32_t rb = (uint32_t)d.rb;
    bool ok = (rb & UNI2) != 0 && (d.nn & 0xff) == L;
    if (ok) {
        const bool seam = (rb & SEAM2) != 0;
        ok = seam || (int)(rb & ~UNI2) - d.ra == 2 * WAVE;
        unsigned long long p = 0;
        for (int t = 0; t < L && t < 8; ++t) p |= (unsigned long long)code[d.pa + t] << (8 * t);
        ok = ok && p == pat;
    }
    flag[j] = ok ? 1 : 0;
}

// UX: the dot operand is the input vector itself (mul_vec_dot, MINRES' v.Av, K4 without a preconditioner): taken from the window
template <int DOT, bool UX, int UL, int FL, int FH, int W>
__global__ __launch_bounds__(BLOCK) void spmv_tile_kernel(const int2 *__restrict__ tile_list, const int32_t *__restrict__ xstart,
                                                          const BlkDesc *__restrict__ desc, const TilePat pat,
                                                          int n_left, const int32_t *__restrict__ left_order,
                                                          const int32_t *__restrict__ row_ptr, const uint8_t *__restrict__ code,
                                                          const int32_t *__restrict__ off_tab, const double *__restrict__ val_tab,
                                                          const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ u,
                                                          double *__restrict__ part0, double *__restrict__ part1,
                                                          const int *__restrict__ status, int nrows, int ncols, const Fin fin) {
    using T = double;
    constexpr int TR = TILE_ROWS;                       // W: half-width of the window (TILE_W, or TILE_W_WIDE for line bands up to 1534)
    constexpr int NW = (TR + 2 * W) / 2 / BLOCK;        // 16-byte window pieces per lane
    constexpr int NQ = TILE_B / NWAVE;                  // 128-row blocks per wavefront and tile
    constexpr int NN = UL - FL - FH;                    // near slots
    static_assert((TR + 2 * W) % (2 * BLOCK) == 0 && TILE_B % NWAVE == 0 && NN >= 1 && UL <= 8, "tile shape");
    __shared__ __attribute__((aligned(16))) T win[TR + 2 * W];
    __shared__ PairEnt<T> s_pair[TAB];
    __shared__ __attribute__((aligned(16))) uint32_t s_c[NWAVE][CW2];
    __shared__ T red[NWAVE];
    const int run_state = status != nullptr ? *status : (int)ST_RUNNING;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    s_pair[tid] = PairEnt<T>{off_tab[tid] * 8, val_tab[tid]};                  // BLOCK == TAB (the seam rows' own values; the walk below)
    for (int i = lane; i < CW2; i += WAVE) s_c[wv][i] = 0;
    __syncthreads();
    if (run_state != ST_RUNNING) return;
    T d0 = 0.0, d1 = 0.0;

    const int xcd = blockIdx.x & 7;
    const int sstep = gridDim.x >> 3;
    const int send = xstart[xcd + 1];
    int s = xstart[xcd] + (blockIdx.x >> 3);
    // tile_list entries: {first 128-row block, first row}; the next tile's entry is requested a tile ahead
    // ... and the seam words of its blocks (wave-uniform: scalar loads) are read a tile ahead too
    int2 ent = s < send ? tile_list[s] : int2{0, 0};
    int2 ent1 = s + sstep < send ? tile_list[s + sstep] : int2{0, 0};
    uint32_t rbw[NQ]; int nnw[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const BlkDesc d = desc[__builtin_amdgcn_readfirstlane(ent.x) + q * NWAVE + wv]; rbw[q] = (uint32_t)d.rb; nnw[q] = d.nn; }
    for (; s < send; s += sstep) {
        const int ts = __builtin_amdgcn_readfirstlane(ent.y);
        ent = ent1;
        if (s + 2 * sstep < send) ent1 = tile_list[s + 2 * sstep];
        uint32_t rbc[NQ]; int nnc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { rbc[q] = rbw[q]; nnc[q] = nnw[q]; }
        // ---- loads: the window, then the far pairs (and dot operands) of the lane's NQ row pairs
        u4v wreg[NW];
        const T *wbase = x + (ts - W);
#pragma unroll
        for (int i = 0; i < NW; ++i) wreg[i] = *reinterpret_cast<const u4v *>(wbase + 2 * (tid + i * BLOCK));
        D2 far[NQ][FL + FH > 0 ? FL + FH : 1];
        D2 uu[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const T *xr = x + (ts + ((q * NWAVE + wv) << 7) + 2 * lane);
#pragma unroll
            for (int k = 0; k < FL; ++k) far[q][k] = *reinterpret_cast<const D2 *>(xr + pat.off[k]);
#pragma unroll
            for (int k = 0; k < FH; ++k) far[q][FL + k] = *reinterpret_cast<const D2 *>(xr + pat.off[UL - FH + k]);
            if (DOT != 0 && !UX) {
                const u4v w4 = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(u + (ts + ((q * NWAVE + wv) << 7) + 2 * lane)));
                __builtin_memcpy(&uu[q], &w4, 16);
            }
        }
        if (s + sstep < send) {
            const int b1 = __builtin_amdgcn_readfirstlane(ent.x);
#pragma unroll
            for (int q = 0; q < NQ; ++q) { const BlkDesc d = desc[b1 + q * NWAVE + wv]; rbw[q] = (uint32_t)d.rb; nnw[q] = d.nn; }
        }
        __syncthreads();                                                        // the previous tile's window has been read
#pragma unroll
        for (int i = 0; i < NW; ++i) *reinterpret_cast<u4v *>(&win[2 * (tid + i * BLOCK)]) = wreg[i];
        __syncthreads();
        // near slots from the window; the reads of block q + 1 are issued before block q is folded (the seam branch
        // below keeps the compiler from doing that itself, and a fold behind an exposed LDS round trip eight times per
        // tile is 10 % of the launch)
        T npl[NN], nph[NN];
        auto read_near = [&](int q) {
            const int li = W + ((q * NWAVE + wv) << 7) + 2 * lane;
#pragma unroll
            for (int t = 0; t < NN; ++t) { npl[t] = win[li + pat.off[FL + t]]; nph[t] = win[li + pat.off[FL + t] + 1]; }
        };
        read_near(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint32_t rbq = (uint32_t)__builtin_amdgcn_readfirstlane((int)rbc[q]);
            const bool seam = (rbq & SEAM2) != 0;
            const int li = W + ((q * NWAVE + wv) << 7) + 2 * lane;              // window index of x[r0]
            T pl[UL], ph[UL];
#pragma unroll
            for (int t = 0; t < UL; ++t) {
                if (t < FL) { pl[t] = far[q][t].lo; ph[t] = far[q][t].hi; }
                else if (t >= UL - FH) { pl[t] = far[q][t - (UL - FH) + FL].lo; ph[t] = far[q][t - (UL - FH) + FL].hi; }
                else { pl[t] = npl[t - FL]; ph[t] = nph[t - FL]; }
            }
            T ux0 = 0.0, ux1 = 0.0;
            if (DOT != 0 && UX) { ux0 = win[li]; ux1 = win[li + 1]; }
            if (q + 1 < NQ) read_near(q + 1);
            T acc0 = 0.0, acc1 = 0.0;
            if (!seam) {
#pragma unroll
                for (int t = 0; t < UL; ++t) {
                    acc0 = acc0 + pl[t] * pat.val[t];
                    acc1 = acc1 + ph[t] * pat.val[t];
                }
            } else {
                // (full_uniform_block's seam fold) local rows k and k + 1 fold only the slots of their masks, with their
                // own value where they carry one.  A slot BOTH of them have is a plain step for the whole wavefront (a
                // scalar test): a stencil's line seam costs two masked steps, not UL
                const int seam1 = __builtin_amdgcn_readfirstlane(nnc[q]) >> 16, seam2 = (int)(rbq & 0x3ffffffu);
                const int k = seam1 & 127, maskA = (seam1 >> 7) & 255, maskB = seam2 & 255;
                const bool a0 = 2 * lane == k, a1 = 2 * lane + 1 == k, b0s = 2 * lane == k + 1, b1s = 2 * lane + 1 == k + 1;
                if ((seam2 & 0x3000000) == 0) {
                    // no row with a value of its own (a stencil's line seam): a step is plain unless row k or k + 1 lacks
                    // the slot (scalar tests), and then those lanes alone keep their sum
#pragma unroll
                    for (int t = 0; t < UL; ++t) {
                        const bool am = ((maskA >> t) & 1) == 0, bm = ((maskB >> t) & 1) == 0;       // scalar
                        const T n0 = acc0 + pl[t] * pat.val[t], n1 = acc1 + ph[t] * pat.val[t];
                        if (!am && !bm) { acc0 = n0; acc1 = n1; }
                        else {
                            acc0 = ((am && a0) || (bm && b0s)) ? acc0 : n0;
                            acc1 = ((am && a1) || (bm && b1s)) ? acc1 : n1;
                        }
                    }
                } else {
                    const T valA = s_pair[(seam2 >> 8) & 255].val, valB = s_pair[(seam2 >> 16) & 255].val;
                    const bool ovA = ((seam2 >> 24) & 1) != 0, ovB = ((seam2 >> 25) & 1) != 0;
                    const int pm0 = a0 ? maskA : (b0s ? maskB : 255), pm1 = a1 ? maskA : (b1s ? maskB : 255);
                    const bool o0 = (a0 && ovA) || (b0s && ovB), o1 = (a1 && ovA) || (b1s && ovB);
                    const T v0 = a0 ? valA : valB, v1 = a1 ? valA : valB;
#pragma unroll
                    for (int t = 0; t < UL; ++t) {
                        const T n0 = acc0 + pl[t] * (o0 ? v0 : pat.val[t]), n1 = acc1 + ph[t] * (o1 ? v1 : pat.val[t]);
                        acc0 = ((pm0 >> t) & 1) ? n0 : acc0;
                        acc1 = ((pm1 >> t) & 1) ? n1 : acc1;
                    }
                }
            }
            const D2 yy{acc0, acc1};
            u4v qv;
            __builtin_memcpy(&qv, &yy, 16);
            __builtin_nontemporal_store(qv, reinterpret_cast<u4v *>(y + (ts + ((q * NWAVE + wv) << 7) + 2 * lane)));
            if (DOT == 1) { d0 = d0 + (UX ? ux0 : uu[q].lo) * acc0; d0 = d0 + (UX ? ux1 : uu[q].hi) * acc1; }
            if (DOT == 2) {
                const T u0 = UX ? ux0 : uu[q].lo, u1 = UX ? ux1 : uu[q].hi;
                d0 = d0 + acc0 * acc0; d1 = d1 + acc0 * u0; d0 = d0 + acc1 * acc1; d1 = d1 + acc1 * u1;
            }
        }
    }
    if (n_left > 0) {
        __syncthreads();
        pair2_walk<DOT, true>(n_left, 0, desc, left_order, row_ptr, code, x, y, u, nrows, ncols, s_pair, s_c, d0, d1);
    }
    if (DOT >= 1) {
        d0 = block_sum(d0, red);
        if (tid == 0) st_partial(fin, part0 + blockIdx.x, d0);
    }
    if (DOT == 2) {
        d1 = block_sum(d1, red);
        if (tid == 0) st_partial(fin, part1 + blockIdx.x, d1);
    }
    if (DOT >= 1 && fin.counter) finalize_last_block<T, T>(fin, DOT == 2, red, red);
}

// The same tiles for the OFFSET-CODE stream (a value per entry: any stencil or band with variable coefficients).  The 128-row
// descriptors are the offset stream's own (owide_desc: uniform and seam blocks marked on the offset codes).  A full block's
// 128 UL values are consecutive in val[]: the wavefront loads them with 16-byte loads (stream order), passes them through its
// LDS slice one block ahead of the fold and reads them back by row — lane l's rows 2l, 2l + 1 sit at entries (2l) UL and
// (2l + 1) UL behind the block's first one; in a seam block the rows behind the short ones move up by what those lack, and a
// short row steps through its values only on the slots it has.  x as in spmv_tile_kernel.  The 64-row blocks outside the tiles
// go through dict_walk in the same launch.
template <int DOT, bool UX, int UL, int FL, int FH>
__global__ __launch_bounds__(BLOCK) void spmv_tile_off_kernel(const int2 *__restrict__ tile_list, const int32_t *__restrict__ xstart,
                                                              const BlkDesc *__restrict__ desc, const TilePat pat,
                                                              int n_left, const int32_t *__restrict__ left_order, const BlkDesc *__restrict__ desc64,
                                                              const int32_t *__restrict__ row_ptr, const uint8_t *__restrict__ code,
                                                              const int32_t *__restrict__ off_tab, const double *__restrict__ val,
                                                              const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ u,
                                                              double *__restrict__ part0, double *__restrict__ part1,
                                                              const int *__restrict__ status, const Fin fin, const V2d *__restrict__ tail2, int g2_last) {
    using T = double;
    constexpr int TR = TILE_ROWS, W = TILE_W;
    constexpr int NW = (TR + 2 * W) / 2 / BLOCK;        // 16-byte window pieces per lane
    constexpr int NQ = TILE_B / NWAVE;                  // 128-row blocks per wavefront and tile
    constexpr int NN = UL - FL - FH;                    // near slots
    constexpr int VROW = 2 * WAVE * UL;                 // values of a full block
    constexpr int NV = (VROW / 2 + 1 + WAVE - 1) / WAVE;   // 16-byte value loads per lane and block (a block may start on an odd entry)
    constexpr int CAPD = nnz_cap<T>::value;
    constexpr int CWD = (CAPD + 3 + CPAD + 3) / 4;
    constexpr int VS = (VROW + 2 + 15) / 16 * 16 > CAPD + 16 ? (VROW + 2 + 15) / 16 * 16 : CAPD + 16;    // slice stride: the tile phase's block, or dict_walk's
    static_assert((TR + 2 * W) % (2 * BLOCK) == 0 && TILE_B % NWAVE == 0 && NN >= 1 && UL <= 8, "tile shape");
    __shared__ __attribute__((aligned(16))) T win[TR + 2 * W];
    __shared__ __attribute__((aligned(16))) T vsl[NWAVE][VS];
    __shared__ int32_t s_off8[TAB];
    __shared__ uint32_t s_c[NWAVE][CWD];
    __shared__ T red[NWAVE];
    const int run_state = status != nullptr ? *status : (int)ST_RUNNING;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    s_off8[tid] = off_tab[tid] * (int32_t)sizeof(T);                            // BLOCK == TAB
    for (int i = lane; i < CWD; i += WAVE) s_c[wv][i] = 0;
    for (int i = lane; i < VS; i += WAVE) vsl[wv][i] = 0.0;
    __syncthreads();
    if (run_state != ST_RUNNING) return;
    T d0 = 0.0, d1 = 0.0;

    const int xcd = blockIdx.x & 7;
    const int sstep = gridDim.x >> 3;
    const int send = xstart[xcd + 1];
    int s = xstart[xcd] + (blockIdx.x >> 3);
    // tile entries {first 128-row block, first row} and per block its seam words and its first entry: scalar loads a tile ahead
    int2 ent = s < send ? tile_list[s] : int2{0, 0};
    int2 ent1 = s + sstep < send ? tile_list[s + sstep] : int2{0, 0};
    uint32_t rbw[NQ]; int nnw[NQ], vbw[NQ];
    auto load_words = [&](int b0, int ts) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const BlkDesc d = desc[b0 + q * NWAVE + wv];
            rbw[q] = (uint32_t)d.rb; nnw[q] = d.nn; vbw[q] = row_ptr[ts + ((q * NWAVE + wv) << 7)];
        }
    };
    load_words(__builtin_amdgcn_readfirstlane(ent.x), __builtin_amdgcn_readfirstlane(ent.y));
    T *vs = vsl[wv];
    for (; s < send; s += sstep) {
        const int ts = __builtin_amdgcn_readfirstlane(ent.y);
        ent = ent1;
        if (s + 2 * sstep < send) ent1 = tile_list[s + 2 * sstep];
        uint32_t rbc[NQ]; int nnc[NQ], vbc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { rbc[q] = rbw[q]; nnc[q] = nnw[q]; vbc[q] = __builtin_amdgcn_readfirstlane(vbw[q]); }
        // ---- loads: the window, the far pairs (and dot operands) of the lane's NQ row pairs, the first block's values
        u4v wreg[NW];
        const T *wbase = x + (ts - W);
#pragma unroll
        for (int i = 0; i < NW; ++i) wreg[i] = *reinterpret_cast<const u4v *>(wbase + 2 * (tid + i * BLOCK));
        D2 far[NQ][FL + FH > 0 ? FL + FH : 1];
        D2 uu[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const T *xr = x + (ts + ((q * NWAVE + wv) << 7) + 2 * lane);
#pragma unroll
            for (int k = 0; k < FL; ++k) far[q][k] = *reinterpret_cast<const D2 *>(xr + pat.off[k]);
#pragma unroll
            for (int k = 0; k < FH; ++k) far[q][FL + k] = *reinterpret_cast<const D2 *>(xr + pat.off[UL - FH + k]);
            if (DOT != 0 && !UX) {
                const u4v w4 = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(u + (ts + ((q * NWAVE + wv) << 7) + 2 * lane)));
                __builtin_memcpy(&uu[q], &w4, 16);
            }
        }
        u4v vreg[NV];
        auto load_vals = [&](int vb) {                  // chunks [vb >> 1, (vb >> 1) + VROW / 2]: the block's values from its 16-byte boundary
            const u4v *v2 = reinterpret_cast<const u4v *>(val) + (vb >> 1);
#pragma unroll
            for (int i = 0; i < NV; ++i) vreg[i] = __builtin_nontemporal_load(v2 + min(lane + i * WAVE, VROW / 2));      // read once: 720 -> 674 us (profiles/r03_tuning.md §9)
        };
        load_vals(vbc[0]);
        if (s + sstep < send) load_words(__builtin_amdgcn_readfirstlane(ent.x), __builtin_amdgcn_readfirstlane(ent.y));
        __syncthreads();                                                        // the previous tile's window has been read
#pragma unroll
        for (int i = 0; i < NW; ++i) *reinterpret_cast<u4v *>(&win[2 * (tid + i * BLOCK)]) = wreg[i];
        __syncthreads();
        T npl[NN], nph[NN];
        auto read_near = [&](int q) {
            const int li = W + ((q * NWAVE + wv) << 7) + 2 * lane;
#pragma unroll
            for (int t = 0; t < NN; ++t) { npl[t] = win[li + pat.off[FL + t]]; nph[t] = win[li + pat.off[FL + t] + 1]; }
        };
        read_near(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // this block's values: registers -> the wavefront's slice (stream order), then the next block's loads go out
#pragma unroll
            for (int i = 0; i < NV; ++i) *reinterpret_cast<u4v *>(&vs[2 * min(lane + i * WAVE, VROW / 2)]) = vreg[i];
            if (q + 1 < NQ) load_vals(vbc[q + 1 < NQ ? q + 1 : q]);
            wave_lds_fence();
            const uint32_t rbq = (uint32_t)__builtin_amdgcn_readfirstlane((int)rbc[q]);
            const bool seam = (rbq & SEAM2) != 0;
            const int shift = vbc[q] & 1;
            const int li = W + ((q * NWAVE + wv) << 7) + 2 * lane;              // window index of x[r0]
            T pl[UL], ph[UL];
#pragma unroll
            for (int t = 0; t < UL; ++t) {
                if (t < FL) { pl[t] = far[q][t].lo; ph[t] = far[q][t].hi; }
                else if (t >= UL - FH) { pl[t] = far[q][t - (UL - FH) + FL].lo; ph[t] = far[q][t - (UL - FH) + FL].hi; }
                else { pl[t] = npl[t - FL]; ph[t] = nph[t - FL]; }
            }
            T ux0 = 0.0, ux1 = 0.0;
            if (DOT != 0 && UX) { ux0 = win[li]; ux1 = win[li + 1]; }
            if (q + 1 < NQ) read_near(q + 1);
            T acc0 = 0.0, acc1 = 0.0;
            if (!seam) {
                const T *v0 = vs + shift + 2 * lane * UL;
#pragma unroll
                for (int t = 0; t < UL; ++t) {
                    acc0 = acc0 + pl[t] * v0[t];
                    acc1 = acc1 + ph[t] * v0[UL + t];
                }
            } else {
                // rows k, k + 1 (local) hold only the slots of their masks: their values are fewer, and the rows behind them
                // start that much earlier in the slice
                const int seam1 = __builtin_amdgcn_readfirstlane(nnc[q]) >> 16, seam2 = (int)(rbq & 0x3ffffffu);
                const int k = seam1 & 127, full = (1 << UL) - 1, maskA = (seam1 >> 7) & full, maskB = seam2 & full;
                const int cA = UL - __builtin_popcount(maskA), cB = UL - __builtin_popcount(maskB);
                const int i0 = 2 * lane, i1 = i0 + 1;
                int p0 = shift + i0 * UL - (i0 > k ? cA : 0) - (i0 > k + 1 ? cB : 0);
                int p1 = shift + i1 * UL - (i1 > k ? cA : 0) - (i1 > k + 1 ? cB : 0);
                const int pm0 = i0 == k ? maskA : (i0 == k + 1 ? maskB : full), pm1 = i1 == k ? maskA : (i1 == k + 1 ? maskB : full);
#pragma unroll
                for (int t = 0; t < UL; ++t) {
                    const T n0 = acc0 + pl[t] * vs[p0], n1 = acc1 + ph[t] * vs[p1];
                    const bool h0 = ((pm0 >> t) & 1) != 0, h1 = ((pm1 >> t) & 1) != 0;
                    acc0 = h0 ? n0 : acc0; p0 += h0 ? 1 : 0;
                    acc1 = h1 ? n1 : acc1; p1 += h1 ? 1 : 0;
                }
            }
            const D2 yy{acc0, acc1};
            u4v qv;
            __builtin_memcpy(&qv, &yy, 16);
            __builtin_nontemporal_store(qv, reinterpret_cast<u4v *>(y + (ts + ((q * NWAVE + wv) << 7) + 2 * lane)));
            if (DOT == 1) { d0 = d0 + (UX ? ux0 : uu[q].lo) * acc0; d0 = d0 + (UX ? ux1 : uu[q].hi) * acc1; }
            if (DOT == 2) {
                const T u0 = UX ? ux0 : uu[q].lo, u1 = UX ? ux1 : uu[q].hi;
                d0 = d0 + acc0 * acc0; d1 = d1 + acc0 * u0; d0 = d0 + acc1 * acc1; d1 = d1 + acc1 * u1;
            }
            wave_lds_fence();                                                   // the slice is free for the next block's values
        }
    }
    if (n_left > 0) {
        __syncthreads();
        dict_walk<T, DOT, false, false, true>(n_left, 0, desc64, left_order, row_ptr, code, val, x, y, u, tail2, g2_last,
                                             (const PairEnt<T> *)nullptr, s_off8, s_c, &vsl[0][0], VS, d0, d1);
    }
    if (DOT >= 1) {
        d0 = block_sum(d0, red);
        if (tid == 0) st_partial(fin, part0 + blockIdx.x, d0);
    }
    if (DOT == 2) {
        d1 = block_sum(d1, red);
        if (tid == 0) st_partial(fin, part1 + blockIdx.x, d1);
    }
    if (DOT >= 1 && fin.counter) finalize_last_block<T, T>(fin, DOT == 2, red, red);
}

// XCD-period schedule (knob "spmv_period") of a stream's row blocks.  With the far band P = max |col - row| (a 3-D
// stencil's plane), rows are cut into chunks of P/8 and chunk c goes to XCD c mod 8: rows r and r +- P are multiplied on
// the SAME XCD one chunk apart, so x[r + P] is fetched over the fabric once — when row r needs it — and hits that
// XCD's L2 as the centre of row r + P (and, where the stream between them is short enough, as the lower neighbour of
// row r + 2P).  Positions of the walk belong to XCD (pos / NWAVE) mod 8 (workgroups are dealt round-robin over the
// XCDs — a locality hint only, never needed for correctness).  first_row(j) = first row of block j.
template <class FirstRow>
static std::vector<int32_t> xcd_period_order(int nblk, int64_t G, FirstRow first_row) {
    std::vector<std::vector<int32_t>> q(8);
    for (int j = 0; j < nblk; ++j) q[(size_t)((first_row(j) / G) % 8)].push_back(j);
    std::vector<int32_t> ord((size_t)nblk);
    size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0}, filled = 0;
    while (filled < (size_t)nblk) {
        for (int x = 0; x < 8 && filled < (size_t)nblk; ++x) {
            for (int k = 0; k < NWAVE && filled < (size_t)nblk; ++k) {
                int src = x;
                if (pos[src] >= q[src].size()) {          // this XCD's queue ran dry: take from the longest one
                    size_t best = 0;
                    for (int y2 = 0; y2 < 8; ++y2)
                        if (q[y2].size() - pos[y2] > best) { best = q[y2].size() - pos[y2]; src = y2; }
                }
                ord[filled++] = q[src][pos[src]++];
            }
        }
    }
    return ord;
}

// (UL, FL, FH) shapes spmv_tile_kernel is built for: 7-point 3-D, 5-point 2-D with a far or a near line band, 3-point 1-D, and
// bands with two far diagonals
#define SPRS_TILE_SHAPES(X) X(7, 1, 1) X(5, 1, 1) X(5, 0, 0) X(3, 0, 0) X(3, 1, 1) X(7, 0, 0)
// ... and with the wide window (pair-code stream only; grids whose lines are 511 to 1534 long)
#define SPRS_TILE_SHAPES_WIDE(X) X(7, 1, 1) X(5, 1, 1) X(5, 0, 0) X(7, 0, 0)
static bool tile_shape_built(int ul, int fl, int fh, int w) {
#define SPRS_TILE_HAS(U, L, H) if (ul == U && fl == L && fh == H) return true;
    if (w == TILE_W) { SPRS_TILE_SHAPES(SPRS_TILE_HAS) }
    if (w == TILE_W_WIDE) { SPRS_TILE_SHAPES_WIDE(SPRS_TILE_HAS) }
#undef SPRS_TILE_HAS
    return false;
}

struct BlkDescHost2 { int32_t ra, rb, pa, nn; };
template <class T> struct has_val_dict { static constexpr bool value = false; };
template <> struct has_val_dict<double> { static constexpr bool value = true; };
template <> struct has_val_dict<float> { static constexpr bool value = true; };

// Tile plan of a stream's 128-row descriptors `desc_dev` (marked by mark_uniform_kernel on the codes `code_dev`; host copy of
// the unmarked descriptors: wd).  off_tab / val_tab: host tables per code (val_tab null: a stream whose values are per entry).
// make_left(in_tile, left) lists the blocks the tiles do not cover, in the walk order of the stream's per-block kernel.
template <class MakeLeft>
static int build_tile_plan(sprs_csr *A, sprs_tile_plan &TP, const BlkDesc *desc_dev, const uint8_t *code_dev, const std::vector<BlkDescHost2> &wd,
                           const int32_t *off_tab, const double *val_tab, bool wide_window_ok, MakeLeft &&make_left) {
    sprs_ctx *c = A->ctx;
    const int nw = (int)wd.size();
    const int n_cand = nw / TILE_B;
    if (!(tile_wanted(c, (size_t)A->nrows * sizeof(double)) && c->spmv_uniform != 0 && c->spmv_wide != 0 && n_cand >= 16 && A->ncols >= TILE_ROWS + 2 * TILE_W)) return SPRS_OK;
    unsigned long long *pat_d = nullptr; int *len_d = nullptr; uint8_t *flag_d = nullptr;
    auto drop = [&]() { for (void *q : {(void *)pat_d, (void *)len_d, (void *)flag_d}) if (q) (void)hipFree(q); pat_d = nullptr; len_d = nullptr; flag_d = nullptr; };
#define TILE_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { snprintf(c->err, sizeof(c->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); drop(); return SPRS_ERR_HIP; } } while (0)
    TILE_TRY(hipMalloc((void **)&pat_d, sizeof(unsigned long long) * (size_t)n_cand));
    TILE_TRY(hipMalloc((void **)&len_d, sizeof(int) * (size_t)n_cand));
    hipLaunchKernelGGL(tile_mark_kernel, dim3((n_cand + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, n_cand, desc_dev, code_dev, pat_d, len_d);
    TILE_TRY(hipGetLastError());
    std::vector<unsigned long long> h_pat((size_t)n_cand);
    std::vector<int> h_len((size_t)n_cand);
    TILE_TRY(hipMemcpyAsync(h_pat.data(), pat_d, sizeof(unsigned long long) * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    TILE_TRY(hipMemcpyAsync(h_len.data(), len_d, sizeof(int) * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    TILE_TRY(hipStreamSynchronize(c->stream));
    // the pattern: the most frequent one among the aligned candidates
    std::map<std::pair<unsigned long long, int>, int> hist;
    for (int t = 0; t < n_cand; ++t) if (h_len[(size_t)t] > 0) ++hist[{h_pat[(size_t)t], h_len[(size_t)t]}];
    std::pair<unsigned long long, int> canon{0ull, 0};
    int best = 0;
    for (const auto &kv : hist) if (kv.second > best) { best = kv.second; canon = kv.first; }
    const int UL = canon.second;
    int FL = 0, FH = 0, WIN = 0;
    bool shape_ok = false;
    int64_t far_band = 0;
    if (best >= 8 && UL >= 1) {
        int32_t o[8];
        for (int t = 0; t < UL; ++t) {
            const int cd = (int)((canon.first >> (8 * t)) & 255u);
            o[t] = off_tab[cd];
            TP.off[t] = o[t];
            TP.val[t] = val_tab ? val_tab[cd] : 0.0;
        }
        // the window for which the pattern is far* near+ far* with a shape the kernels are built for and the fewest loads per
        // 128 rows: (T + 2w) / T window loads + one per far slot (a 2-D grid of 1500-row lines: 1.25 + 2 narrow, 1.75 + 0 wide)
        double best_cost = 1e9;
        for (int w : {TILE_W, TILE_W_WIDE}) {
            if (w == TILE_W_WIDE && !wide_window_ok) continue;
            const int NEAR = w - 2;
            int fl = 0, fh = 0;
            while (fl < UL && o[fl] < -NEAR) ++fl;
            while (fh < UL - fl && o[UL - 1 - fh] > NEAR) ++fh;
            bool ok = UL - fl - fh >= 1 && tile_shape_built(UL, fl, fh, w);
            for (int t = fl; t < UL - fh; ++t) ok = ok && o[t] >= -NEAR && o[t] <= NEAR;
            const double cost = (double)(TILE_ROWS + 2 * w) / TILE_ROWS + fl + fh;
            if (ok && cost < best_cost) { shape_ok = true; best_cost = cost; FL = fl; FH = fh; WIN = w; }
        }
        for (int t = 0; t < FL; ++t) far_band = std::max<int64_t>(far_band, std::llabs((long long)o[t]));
        for (int t = UL - FH; t < UL; ++t) far_band = std::max<int64_t>(far_band, std::llabs((long long)o[t]));
    }
    if (!shape_ok || A->ncols < TILE_ROWS + 2 * WIN) { drop(); return SPRS_OK; }
    // tiles are placed greedily on the runs of consecutive blocks of that pattern (a run ends where a boundary line or plane
    // changes the pattern; tiles on a fixed lattice would lose a whole tile per break)
    TILE_TRY(hipMalloc((void **)&flag_d, (size_t)nw));
    hipLaunchKernelGGL(tile_flag_kernel, dim3((nw + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, nw, desc_dev, code_dev, canon.first, UL, flag_d);
    TILE_TRY(hipGetLastError());
    std::vector<uint8_t> flag((size_t)nw);
    TILE_TRY(hipMemcpyAsync(flag.data(), flag_d, (size_t)nw, hipMemcpyDeviceToHost, c->stream));
    TILE_TRY(hipStreamSynchronize(c->stream));
    if (!val_tab)      // values per entry: 16-byte value loads may reach one entry past a block's last one — keep the matrix's last entries out
        for (int j = 0; j < nw; ++j) if ((int64_t)wd[(size_t)j].pa + 2 * WAVE * UL + 2 > A->nnz) flag[(size_t)j] = 0;
    std::vector<int32_t> starts;                            // first block of each tile, in row order
    std::vector<char> in_tile((size_t)nw, 0);
    for (int j = 0; j < nw;) {
        if (!flag[(size_t)j]) { ++j; continue; }
        int e = j + 1;
        while (e < nw && flag[(size_t)e] && wd[(size_t)e].ra == wd[(size_t)e - 1].ra + 2 * WAVE) ++e;
        int b = j;
        while (b < e && (int64_t)wd[(size_t)b].ra - WIN < 0) ++b;                    // the window starts inside x
        for (; b + TILE_B <= e; b += TILE_B) {
            if ((int64_t)wd[(size_t)b].ra + TILE_ROWS + WIN > A->ncols) break;        // ... and ends inside it
            starts.push_back(b);
            for (int q = 0; q < TILE_B; ++q) in_tile[(size_t)(b + q)] = 1;
        }
        j = e;
    }
    const int n_elig = (int)starts.size();
    if (n_elig < 8) { drop(); return SPRS_OK; }
    // XCD sections: tiles sorted by their phase within the far period (rows r and r +- far_band on one XCD, a near window
    // apart in its walk) and cut into eight sections of equal COUNT (a plane of 60 tiles cut by phase alone gives four XCDs 8
    // tiles a plane and four 7), or plain eighths where no far slot exists / the band does not repeat
    const bool periodic = far_band >= 8 * (int64_t)TILE_ROWS && far_band * 4 <= A->nrows && c->spmv_period != 0;
    std::vector<std::vector<int32_t>> sec(8);
    {
        std::vector<int32_t> by_phase((size_t)n_elig);
        for (int i = 0; i < n_elig; ++i) by_phase[(size_t)i] = i;
        if (periodic)
            std::stable_sort(by_phase.begin(), by_phase.end(), [&](int32_t a, int32_t b) {
                return (int64_t)wd[(size_t)starts[(size_t)a]].ra % far_band < (int64_t)wd[(size_t)starts[(size_t)b]].ra % far_band; });
        for (int i = 0; i < n_elig; ++i) sec[(size_t)(((int64_t)i * 8) / n_elig)].push_back(starts[(size_t)by_phase[(size_t)i]]);
        for (auto &v : sec) std::sort(v.begin(), v.end());      // each XCD walks its tiles in row order (column by column through the planes: 17 % less fabric traffic, same time — profiles/r03_tuning.md §8)
    }
    std::vector<int32_t> list, xstart(9, 0), left;
    for (int xq = 0; xq < 8; ++xq) {
        xstart[(size_t)xq] = (int32_t)(list.size() / 2);
        for (int32_t b0 : sec[(size_t)xq]) { list.push_back(b0); list.push_back(wd[(size_t)b0].ra); }      // {first block, first row}
    }
    xstart[8] = (int32_t)(list.size() / 2);
    make_left(in_tile, left);
    TILE_TRY(hipMalloc((void **)&TP.list, sizeof(int32_t) * list.size()));
    TILE_TRY(hipMalloc((void **)&TP.xstart, sizeof(int32_t) * 9));
    TILE_TRY(hipMalloc((void **)&TP.left, sizeof(int32_t) * std::max<size_t>(left.size(), 1)));
    TILE_TRY(hipMemcpyAsync(TP.list, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice, c->stream));
    TILE_TRY(hipMemcpyAsync(TP.xstart, xstart.data(), sizeof(int32_t) * 9, hipMemcpyHostToDevice, c->stream));
    if (!left.empty()) TILE_TRY(hipMemcpyAsync(TP.left, left.data(), sizeof(int32_t) * left.size(), hipMemcpyHostToDevice, c->stream));
    TILE_TRY(hipStreamSynchronize(c->stream));
    TP.n_tile = n_elig; TP.n_left = (int)left.size();
    TP.ul = UL; TP.fl = FL; TP.fh = FH; TP.w = WIN;
    TP.h_list = list; TP.h_xstart = xstart;
    drop();
#undef TILE_TRY
    return SPRS_OK;
}

template <class T>
int build_dict_t(sprs_csr *A, const std::vector<int32_t> &blk, const std::vector<int32_t> &blk_pa) {
    sprs_ctx *c = A->ctx;
    constexpr bool VALS = has_val_dict<T>::value;
    CreateTrace tr;
    const bool want_vals = VALS && c->spmv_dict != 1;
    const int n = (int)A->nrows;
    uint32_t *off_h = nullptr; unsigned long long *val_h = nullptr; int *counts = nullptr;
    uint8_t *slot_codes = nullptr;
    auto cleanup = [&]() {
        if (off_h) (void)hipFree(off_h);
        if (val_h) (void)hipFree(val_h);
        if (counts) (void)hipFree(counts);
        if (slot_codes) (void)hipFree(slot_codes);
    };
#define DICT_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { snprintf(c->err, sizeof(c->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); cleanup(); return SPRS_ERR_HIP; } } while (0)
    DICT_TRY(hipMalloc((void **)&off_h, sizeof(uint32_t) * HSLOTS));
    DICT_TRY(hipMalloc((void **)&val_h, sizeof(unsigned long long) * HSLOTS));
    DICT_TRY(hipMalloc((void **)&counts, sizeof(int) * 4));
    DICT_TRY(hipMalloc((void **)&slot_codes, 2 * HSLOTS));
    std::vector<uint32_t> h_off(HSLOTS, EMPTY32);
    std::vector<unsigned long long> h_val(HSLOTS, EMPTY64);
    DICT_TRY(hipMemcpyAsync(off_h, h_off.data(), sizeof(uint32_t) * HSLOTS, hipMemcpyHostToDevice, c->stream));
    DICT_TRY(hipMemcpyAsync(val_h, h_val.data(), sizeof(unsigned long long) * HSLOTS, hipMemcpyHostToDevice, c->stream));
    DICT_TRY(hipMemsetAsync(counts, 0, sizeof(int) * 4, c->stream));
    const int g = std::max(1, std::min(c->num_cu * 8, (n + BLOCK - 1) / BLOCK));
    const T *val = reinterpret_cast<const T *>(A->val);
    if (want_vals)
        hipLaunchKernelGGL((dict_collect_kernel<T, VALS>), dim3(g), dim3(BLOCK), 0, c->stream, n, A->row_ptr, A->col_idx, val, off_h, val_h, counts);
    else
        hipLaunchKernelGGL((dict_collect_kernel<T, false>), dim3(g), dim3(BLOCK), 0, c->stream, n, A->row_ptr, A->col_idx, val, off_h, val_h, counts);
    DICT_TRY(hipGetLastError());
    int h_counts[4] = {0, 0, 0, 0};
    DICT_TRY(hipMemcpyAsync(h_counts, counts, sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
    DICT_TRY(hipMemcpyAsync(h_off.data(), off_h, sizeof(uint32_t) * HSLOTS, hipMemcpyDeviceToHost, c->stream));
    DICT_TRY(hipMemcpyAsync(h_val.data(), val_h, sizeof(unsigned long long) * HSLOTS, hipMemcpyDeviceToHost, c->stream));
    DICT_TRY(hipStreamSynchronize(c->stream));
    tr.lap("    dict collect");
    if (h_counts[0] > TAB || h_counts[0] == 0) { cleanup(); return SPRS_OK; }      // too many offsets: plain CSR
    const bool use_vals = want_vals && h_counts[2] == 0 && h_counts[1] > 0 && h_counts[1] <= TAB;
    // codes in ascending key order (deterministic tables regardless of which thread inserted first)
    std::vector<std::pair<int32_t, int>> offs;
    std::vector<std::pair<uint64_t, int>> vals;
    for (int sl = 0; sl < HSLOTS; ++sl) {
        if (h_off[sl] != EMPTY32) offs.push_back({(int32_t)h_off[sl], sl});
        if (use_vals && h_val[sl] != EMPTY64) vals.push_back({(uint64_t)h_val[sl], sl});
    }
    std::sort(offs.begin(), offs.end());
    std::sort(vals.begin(), vals.end());
    if ((int)offs.size() != h_counts[0] || (use_vals && (int)vals.size() != h_counts[1])) { cleanup(); return SPRS_OK; }
    std::vector<uint8_t> codes(2 * HSLOTS, 0);
    std::vector<int32_t> off_tab(TAB, 0);
    std::vector<T> val_tab(TAB, szero<T>());
    for (size_t i = 0; i < offs.size(); ++i) { off_tab[i] = offs[i].first; codes[(size_t)offs[i].second] = (uint8_t)i; }
    if constexpr (VALS) {
        for (size_t i = 0; i < vals.size(); ++i) {
            T v;
            if constexpr (sizeof(T) == 8) { uint64_t k = vals[i].first; memcpy(&v, &k, 8); }
            else { uint32_t k = (uint32_t)vals[i].first; memcpy(&v, &k, 4); }
            val_tab[i] = v;
            codes[HSLOTS + (size_t)vals[i].second] = (uint8_t)i;
        }
    }
    DICT_TRY(hipMemcpyAsync(slot_codes, codes.data(), 2 * HSLOTS, hipMemcpyHostToDevice, c->stream));
    auto *D = new sprs_dict();
    A->dict = D;   // freed by free_dict on every failure path below (sprs_csr_destroy)
    const size_t nb = ((size_t)A->nnz + 3) / 4 * 4 + 64;     // the kernels read whole dwords / 16-byte pieces past the last code
    uint8_t *val_code = nullptr, *seen = nullptr;   // temporaries of the pair stage
    auto cleanup2 = [&]() { cleanup(); if (val_code) (void)hipFree(val_code); if (seen) (void)hipFree(seen); };
#define DICT_TRY2(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { snprintf(c->err, sizeof(c->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); cleanup2(); free_dict(A); return SPRS_ERR_HIP; } } while (0)
    DICT_TRY2(hipMalloc((void **)&D->idx_code, nb));
    DICT_TRY2(hipMemsetAsync(D->idx_code, 0, nb, c->stream));
    DICT_TRY2(hipMalloc((void **)&D->off_tab, sizeof(int32_t) * TAB));
    DICT_TRY2(hipMemcpyAsync(D->off_tab, off_tab.data(), sizeof(int32_t) * TAB, hipMemcpyHostToDevice, c->stream));
    if (use_vals) {
        DICT_TRY2(hipMalloc((void **)&val_code, nb));
        DICT_TRY2(hipMemsetAsync(val_code, 0, nb, c->stream));
    }
    D->n_off = (int)offs.size(); D->n_val = use_vals ? (int)vals.size() : 0;
    for (const auto &o : offs) D->max_off = std::max<int64_t>(D->max_off, std::llabs((long long)o.first));
    int *bad = counts + 3;
    if (use_vals)
        hipLaunchKernelGGL((dict_encode_kernel<T, VALS>), dim3(g), dim3(BLOCK), 0, c->stream, n, A->row_ptr, A->col_idx, val, off_h,
                           slot_codes, val_h, slot_codes + HSLOTS, D->idx_code, val_code, bad);
    else
        hipLaunchKernelGGL((dict_encode_kernel<T, false>), dim3(g), dim3(BLOCK), 0, c->stream, n, A->row_ptr, A->col_idx, val, off_h,
                           slot_codes, val_h, slot_codes + HSLOTS, D->idx_code, val_code, bad);
    DICT_TRY2(hipGetLastError());
    DICT_TRY2(hipMemcpyAsync(h_counts, counts, sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
    DICT_TRY2(hipStreamSynchronize(c->stream));
    tr.lap("    dict encode");
    if (h_counts[3] != 0) { cleanup2(); free_dict(A); return SPRS_OK; }   // cannot happen (every key was inserted)
    {
        // the same schedule for the 64-row blocks of the offset-code stream (variable coefficients: 8-9 B per entry, at
        // the fabric's ceiling with the x re-reads of the natural sweep — profiles/r02_tuning.md §4)
        int64_t P = 0;
        for (const auto &o : offs) P = std::max<int64_t>(P, std::llabs((long long)o.first));
        const int64_t G = P / 8;
        if (c->spmv_period > 0 && G >= 16 * 128 && P * 4 <= A->nrows && A->n_rowblk >= 8 * NWAVE * 8) {
            const std::vector<int32_t> ord = xcd_period_order(A->n_rowblk, G, [&](int j) { return (int64_t)blk[(size_t)j]; });
            DICT_TRY2(hipMalloc((void **)&D->off_order, sizeof(int32_t) * ord.size()));
            DICT_TRY2(hipMemcpyAsync(D->off_order, ord.data(), sizeof(int32_t) * ord.size(), hipMemcpyHostToDevice, c->stream));
            DICT_TRY2(hipStreamSynchronize(c->stream));
            D->period = P;
        }
    }
    if (c->spmv_uniform != 0 && A->n_rowblk > 0) {
        // Uniform 64-row blocks of the offset-code stream: all rows repeat the first row's (<= UNI_OFF_MAXLEN) offset codes — the
        // interior of any stencil or band, whatever its VALUES.  Such a block needs neither row_ptr nor its code bytes
        // (9 B/nnz + 4 B/row -> 8 B/nnz); flagged in a private copy of the descriptors (the plain kernel keeps its own).
        DICT_TRY2(hipMalloc(&D->off_desc, sizeof(BlkDesc) * (size_t)A->n_rowblk));
        DICT_TRY2(hipMemcpyAsync(D->off_desc, A->blk_desc, sizeof(BlkDesc) * (size_t)A->n_rowblk, hipMemcpyDeviceToDevice, c->stream));
        const int gu = std::max(1, std::min(c->num_cu * 8, (A->n_rowblk + NWAVE - 1) / NWAVE));
        hipLaunchKernelGGL(mark_uniform_kernel, dim3(gu), dim3(BLOCK), 0, c->stream, (int)A->n_rowblk,
                           reinterpret_cast<BlkDesc *>(D->off_desc), A->row_ptr, D->idx_code, UNI_OFF_MAXLEN, (const int32_t *)nullptr, 0, 0, 0);
        DICT_TRY2(hipGetLastError());
        // how many qualified: the auto policy wants to know for matrices that live in the Infinity Cache (small: a
        // few MB of descriptors at most); HBM-sized ones take the offset stream anyway
        if ((double)A->nnz * (sizeof(T) + 4) + 3.0 * A->nrows * sizeof(T) < 192.0 * 1024 * 1024) {
            std::vector<BlkDescHost2> hd((size_t)A->n_rowblk);
            DICT_TRY2(hipMemcpyAsync(hd.data(), D->off_desc, sizeof(BlkDescHost2) * hd.size(), hipMemcpyDeviceToHost, c->stream));
            DICT_TRY2(hipStreamSynchronize(c->stream));
            for (const auto &d : hd) D->n_off_uniform += ((uint32_t)d.rb & UNI2) != 0;
        }
    }
    tr.lap("    period order + uniform marks (offset stream)");
    if (use_vals) {
        // ---- pair stage: which (offset code, value code) pairs occur?  <= 256 of them -> one byte per nnz
        const int gk = (int)std::max<int64_t>(1, std::min<int64_t>(c->num_cu * 8, (A->nnz + BLOCK - 1) / BLOCK));
        DICT_TRY2(hipMalloc((void **)&seen, 65536 + 65536));
        DICT_TRY2(hipMemsetAsync(seen, 0, 65536, c->stream));
        hipLaunchKernelGGL(dict_pair_mark_kernel, dim3(gk), dim3(BLOCK), 0, c->stream, A->nnz, D->idx_code, val_code, seen);
        DICT_TRY2(hipGetLastError());
        std::vector<uint8_t> h_seen(65536), pair_of(65536, 0);
        DICT_TRY2(hipMemcpyAsync(h_seen.data(), seen, 65536, hipMemcpyDeviceToHost, c->stream));
        DICT_TRY2(hipStreamSynchronize(c->stream));
        std::vector<int32_t> pair_off(TAB, 0);
        std::vector<T> pair_val(TAB, szero<T>());
        int np = 0;
        for (int pr = 0; pr < 65536; ++pr) {
            if (!h_seen[pr]) continue;
            if (np < TAB) { pair_of[pr] = (uint8_t)np; pair_off[np] = off_tab[pr & 255]; pair_val[np] = val_tab[pr >> 8]; }
            ++np;
        }
        if (np >= 1 && np <= TAB) {
            DICT_TRY2(hipMalloc((void **)&D->pair_code, nb));
            DICT_TRY2(hipMemsetAsync(D->pair_code, 0, nb, c->stream));
            DICT_TRY2(hipMalloc((void **)&D->pair_off, sizeof(int32_t) * TAB));
            DICT_TRY2(hipMalloc(&D->pair_val, sizeof(T) * TAB));
            DICT_TRY2(hipMemcpyAsync(D->pair_off, pair_off.data(), sizeof(int32_t) * TAB, hipMemcpyHostToDevice, c->stream));
            DICT_TRY2(hipMemcpyAsync(D->pair_val, pair_val.data(), sizeof(T) * TAB, hipMemcpyHostToDevice, c->stream));
            DICT_TRY2(hipMemcpyAsync(seen + 65536, pair_of.data(), 65536, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(dict_pair_encode_kernel, dim3(gk), dim3(BLOCK), 0, c->stream, A->nnz, D->idx_code, val_code,
                               seen + 65536, D->pair_code);
            DICT_TRY2(hipGetLastError());
            DICT_TRY2(hipStreamSynchronize(c->stream));
            tr.lap("    pair mark + encode");
            D->n_pair = np;
            if constexpr (sizeof(T) == 8 && !is_complex<T>::value) {
                // 128-row blocks of the two-rows-per-lane kernel: consecutive pairs of the 64-row blocks
                const int nb64 = A->n_rowblk, nw = (nb64 + 1) / 2;
                std::vector<BlkDescHost2> wd((size_t)nw);
                for (int j = 0; j < nw; ++j) {
                    const size_t b0 = (size_t)2 * j, b1 = (size_t)std::min(2 * j + 2, nb64);
                    const int32_t ra = blk[b0], rb = blk[b1];
                    wd[(size_t)j] = BlkDescHost2{ra, rb, blk_pa[b0], blk_pa[b1] - blk_pa[b0]};
                }
                DICT_TRY2(hipMalloc(&D->wide_desc, sizeof(BlkDescHost2) * (size_t)std::max(nw, 1)));
                DICT_TRY2(hipMemcpyAsync(D->wide_desc, wd.data(), sizeof(BlkDescHost2) * (size_t)nw, hipMemcpyHostToDevice, c->stream));
                if (c->spmv_uniform != 0 && nw > 0) {
                    const int gu = std::max(1, std::min(c->num_cu * 8, (nw + NWAVE - 1) / NWAVE));
                    hipLaunchKernelGGL(mark_uniform_kernel, dim3(gu), dim3(BLOCK), 0, c->stream, nw,
                                       reinterpret_cast<BlkDesc *>(D->wide_desc), A->row_ptr, D->pair_code, UNI2_MAXLEN,
                                       (const int32_t *)D->pair_off, c->spmv_triple != 0 ? 1 : 0, c->spmv_seam != 0 ? 1 : 0,
                                       (int)A->ncols);
                    DICT_TRY2(hipGetLastError());
                }
                DICT_TRY2(hipStreamSynchronize(c->stream));
                D->n_wide = nw;
                // XCD-period schedule: the pair-code kernel streams ~26 B per row, so the x lines a row block gathers
                // stay in its XCD's 4 MiB L2 for several planes' worth of that XCD's rows (xcd_period_order above)
                int64_t P = 0;
                for (const auto &o : offs) P = std::max<int64_t>(P, std::llabs((long long)o.first));
                const int64_t G = P / 8;
                std::vector<int32_t> ord;
                if (c->spmv_period != 0 && G >= 16 * 128 && P * 4 <= A->nrows && nw >= 8 * NWAVE * 8) {      // automatic (-1): on
                    ord = xcd_period_order(nw, G, [&](int j) { return (int64_t)wd[(size_t)j].ra; });
                    DICT_TRY2(hipMalloc((void **)&D->wide_order, sizeof(int32_t) * (size_t)nw));
                    DICT_TRY2(hipMemcpyAsync(D->wide_order, ord.data(), sizeof(int32_t) * (size_t)nw, hipMemcpyHostToDevice, c->stream));
                    DICT_TRY2(hipStreamSynchronize(c->stream));
                    D->period = P;
                }
                tr.lap("    wide descriptors, marks, period order");
                // ---- tile plan (spmv_tile_kernel): runs of TILE_B full uniform blocks with the matrix's most frequent pattern
                {
                    std::vector<int32_t> left_order;
                    if (const int st = build_tile_plan(A, D->tile_pair, reinterpret_cast<const BlkDesc *>(D->wide_desc), (const uint8_t *)D->pair_code, wd,
                                                       pair_off.data(), reinterpret_cast<const double *>(pair_val.data()), true, [&](std::vector<char> &in_tile, std::vector<int32_t> &left) {
                            for (int pos = 0; pos < nw; ++pos) {
                                const int j = ord.empty() ? pos : ord[(size_t)pos];
                                if (!in_tile[(size_t)j]) left.push_back(j);
                            }
                        })) { cleanup2(); free_dict(A); return st; }
                    tr.lap("    tile plan");
                }
            }
        }
    }
    if constexpr (sizeof(T) == 8 && !is_complex<T>::value) {
        // ---- tiles of the OFFSET-CODE stream (spmv_tile_off_kernel), where that is the stream the handle multiplies with:
        // its own 128-row descriptors (pairs of the 64-row blocks), uniform and seam blocks marked on the offset codes
        if (tile_wanted(c, (size_t)A->nrows * sizeof(T)) && (D->pair_code == nullptr || c->spmv_dict == 1) && c->spmv_uniform != 0 && c->spmv_wide != 0 && A->n_rowblk >= 2 * TILE_B * 16) {
            const int nb64 = A->n_rowblk, nw = (nb64 + 1) / 2;
            std::vector<BlkDescHost2> owd((size_t)nw);
            for (int j = 0; j < nw; ++j) {
                const size_t b0 = (size_t)2 * j, b1 = (size_t)std::min(2 * j + 2, nb64);
                owd[(size_t)j] = BlkDescHost2{blk[b0], blk[b1], blk_pa[b0], blk_pa[b1] - blk_pa[b0]};
            }
            DICT_TRY2(hipMalloc(&D->owide_desc, sizeof(BlkDescHost2) * (size_t)nw));
            DICT_TRY2(hipMemcpyAsync(D->owide_desc, owd.data(), sizeof(BlkDescHost2) * (size_t)nw, hipMemcpyHostToDevice, c->stream));
            const int gu = std::max(1, std::min(c->num_cu * 8, (nw + NWAVE - 1) / NWAVE));
            hipLaunchKernelGGL(mark_uniform_kernel, dim3(gu), dim3(BLOCK), 0, c->stream, nw, reinterpret_cast<BlkDesc *>(D->owide_desc), A->row_ptr,
                               D->idx_code, UNI2_MAXLEN, (const int32_t *)D->off_tab, 0, c->spmv_seam != 0 ? 1 : 0, (int)A->ncols);
            DICT_TRY2(hipGetLastError());
            DICT_TRY2(hipStreamSynchronize(c->stream));
            D->n_owide = nw;
            if (const int st = build_tile_plan(A, D->tile_off, reinterpret_cast<const BlkDesc *>(D->owide_desc), (const uint8_t *)D->idx_code, owd,
                                               off_tab.data(), (const double *)nullptr, false, [&](std::vector<char> &in_tile, std::vector<int32_t> &left) {
                    for (int j = 0; j < nw; ++j) {          // the per-block kernel's 64-row blocks, natural order
                        if (in_tile[(size_t)j]) continue;
                        left.push_back(2 * j);
                        if (2 * j + 1 < nb64) left.push_back(2 * j + 1);
                    }
                })) { cleanup2(); free_dict(A); return st; }
            if (D->tile_off.n_tile == 0) { (void)hipFree(D->owide_desc); D->owide_desc = nullptr; D->n_owide = 0; }
            tr.lap("    tile plan (offset stream)");
        }
    }
    cleanup2();
    return SPRS_OK;
#undef DICT_TRY
#undef DICT_TRY2
}

}  // namespace

int tile_blocks() { return TILE_B; }
bool tile_plan_used(const sprs_csr *A) {
    const sprs_ctx *c = A->ctx;
    if (!A->dict || c->spmv_tile == 0 || c->spmv_wide == 0) return false;
    const int dm = dict_mode(A);
    if (dm == 2) return A->dict->tile_pair.n_tile > 0;
    return dm == 1 && A->dict->tile_off.n_tile > 0 && A->tail != nullptr && c->spmv_wideload != 0 && c->spmv_uniform != 0;
}

void free_dict(sprs_csr *A) {
    if (!A || !A->dict) return;
    sprs_dict *D = A->dict;
    for (void *q : {(void *)D->idx_code, (void *)D->pair_code, (void *)D->off_tab, (void *)D->pair_off, D->pair_val, D->wide_desc, D->off_desc, (void *)D->wide_order, (void *)D->off_order,
                    (void *)D->tile_pair.list, (void *)D->tile_pair.xstart, (void *)D->tile_pair.left,
                    (void *)D->tile_off.list, (void *)D->tile_off.xstart, (void *)D->tile_off.left, D->owide_desc})
        if (q) (void)hipFree(q);
    delete D;
    A->dict = nullptr;
}

int build_dict(sprs_csr *A, bool has_vector_blocks, const std::vector<int32_t> &blk, const std::vector<int32_t> &blk_pa) {
    if (A->ctx->spmv_dict == 0 || has_vector_blocks || A->nnz == 0 || A->nrows == 0) return SPRS_OK;
    switch (A->dtype) {
        case DT_D: return build_dict_t<double>(A, blk, blk_pa);
        case DT_Z: return build_dict_t<cplx>(A, blk, blk_pa);
        case DT_S: return build_dict_t<float>(A, blk, blk_pa);
        default: return build_dict_t<cplxf>(A, blk, blk_pa);
    }
}

int dict_mode(const sprs_csr *A) {
    if (!A->dict || A->ctx->spmv_dict == 0) return 0;
    // the kernel addresses x, y, row_ptr and the codes with 32-bit byte offsets from their bases
    if ((uint64_t)std::max(A->ncols, A->nrows + 1) * std::max<size_t>(dtype_size(A->dtype), 4) >= (1ull << 32)) return 0;
    if (A->dict->pair_code && A->ctx->spmv_dict != 1) return 2;
    if (A->ctx->spmv_dict == -1) {
        // offset codes + values for every REAL matrix that has them (measured, profiles/r02_tuning.md): HBM-sized ones
        // (cfg-5 pattern, random values: 885 vs 1130 us) and cache-resident ones alike (cfg 3: 21.8 vs 23.9 us, MINRES
        // 21.8 k vs 20.8 k it/s).  Complex ones run slower (cfg 4: 24.7 vs 15.0 us; 17 instead of 20 B/nnz is not worth
        // the lane-per-row layout): auto keeps the plain stream for those.
        if (dtype_is_complex(A->dtype)) return 0;
    }
    return 1;
}

template <class T>
int launch_spmv_dict(const sprs_csr *A, int mode, const int32_t *order, int count, int g, int xcd_chunk, const T *x, T *y,
                     int dot_mode, const T *u, T *part0, T *part1, const int *status, bool conj_x, const Fin &fin) {
    sprs_ctx *c = A->ctx;
    const sprs_dict *D = A->dict;
    const T *v = reinterpret_cast<const T *>(A->val);
    const T *pv = reinterpret_cast<const T *>(D->pair_val);
    const bool pair = has_val_dict<T>::value && mode == 2;
    if constexpr (sizeof(T) == 8 && !is_complex<T>::value) {
        // f64 pair codes: two rows per lane
        // ... on the whole matrix in natural order, or on the interior / boundary subsets of a distributed operator,
        // whose split is made on pairs of 64-row blocks for this purpose (dist.hip)
        const int32_t *order_w = nullptr;
        int count_w = -1;
        if (order == nullptr && count == A->n_rowblk) { count_w = D->n_wide; order_w = c->spmv_period != 0 ? D->wide_order : nullptr; if (order_w) xcd_chunk = 0; }   // the period order encodes its XCD placement for the round-robin walk
        else if (A->dist && A->dist->order_int_w && order == A->dist->order_int && count == A->dist->n_int) { order_w = A->dist->order_int_w; count_w = A->dist->n_int_w; }
        else if (A->dist && A->dist->order_bnd_w && order == A->dist->order_bnd && count == A->dist->n_bnd) { order_w = A->dist->order_bnd_w; count_w = A->dist->n_bnd_w; }
        // which tile plan this launch runs through: the handle's (whole matrix) or the distributed operator's interior one
        const bool whole = order == nullptr && count == A->n_rowblk;
        const bool interior = A->dist && A->dist->order_int && order == A->dist->order_int && count == A->dist->n_int && A->dist->tile_int.n_tile > 0;
        const sprs_tile_plan *tpp = nullptr;
        if (pair && whole && D->tile_pair.n_tile > 0) tpp = &D->tile_pair;
        else if (pair && interior && !A->dist->tile_int_off) tpp = &A->dist->tile_int;
        if (tpp && c->spmv_tile != 0 && c->spmv_wide != 0 && g % 8 == 0) {
            // LDS x-window tiles + the per-block walk over the blocks outside them, one launch
            const BlkDesc *wd = reinterpret_cast<const BlkDesc *>(D->wide_desc);
            const double *pvd = reinterpret_cast<const double *>(D->pair_val);
            TilePat tp;
            const sprs_tile_plan &TP = *tpp;
            for (int t = 0; t < 8; ++t) { tp.off[t] = TP.off[t]; tp.val[t] = TP.val[t]; }
            const bool ux = dot_mode != 0 && u == x;      // the dot operand is the input vector (mul_vec_dot, MINRES' v.Av, K4 without a preconditioner)
#define SPRS_TSPMV(DM, UXV, U, L, H) SPRS_LAUNCH_SPMV(c, (spmv_tile_kernel<DM, UXV, U, L, H, SPRS_TW>), g, reinterpret_cast<const int2 *>(TP.list), TP.xstart, wd, tp, TP.n_left, \
                                                      TP.left, A->row_ptr, D->pair_code, D->pair_off, pvd, x, y, u, part0, part1, status, (int)A->nrows, (int)A->ncols, fin)
#define SPRS_TSHAPE(U, L, H)                                                                                             \
            if (TP.ul == U && TP.fl == L && TP.fh == H) {                                                 \
                if (dot_mode == 0) SPRS_TSPMV(0, false, U, L, H);                                                        \
                else if (dot_mode == 1) { if (ux) SPRS_TSPMV(1, true, U, L, H); else SPRS_TSPMV(1, false, U, L, H); }    \
                else if (ux) SPRS_TSPMV(2, true, U, L, H); else SPRS_TSPMV(2, false, U, L, H);                           \
            }
#define SPRS_TW TILE_W
            if (TP.w == TILE_W) { SPRS_TILE_SHAPES(SPRS_TSHAPE) }
#undef SPRS_TW
#define SPRS_TW TILE_W_WIDE
            if (TP.w == TILE_W_WIDE) { SPRS_TILE_SHAPES_WIDE(SPRS_TSHAPE) }
#undef SPRS_TW
#undef SPRS_TSHAPE
#undef SPRS_TSPMV
            SPRS_HIP_TRY(c, hipGetLastError());
            return SPRS_OK;
        }
        if (pair && D->wide_desc && c->spmv_wide != 0 && count_w >= 0 && A->nrows >= 2 && A->ncols >= 2) {
            const int gw = g;     // same grid as the 64-row kernel: the consumers reduce exactly spmv_num_partials(A) partials
            const BlkDesc *wd = reinterpret_cast<const BlkDesc *>(D->wide_desc);
            const double *pvd = reinterpret_cast<const double *>(D->pair_val);
#define SPRS_WSPMV(DM, YN) SPRS_LAUNCH_SPMV(c, (spmv_pair2_kernel<DM, YN>), gw, count_w, xcd_chunk, wd, order_w, \
                                          A->row_ptr, D->pair_code, D->pair_off, pvd, x, y, u, part0, part1, status, (int)A->nrows, (int)A->ncols, fin)
            if (stream_loads_nt(c, (size_t)A->nrows * sizeof(T))) {          // HBM-sized result: non-temporal y stores
                if (dot_mode == 0) SPRS_WSPMV(0, true); else if (dot_mode == 1) SPRS_WSPMV(1, true); else SPRS_WSPMV(2, true);
            } else {
                if (dot_mode == 0) SPRS_WSPMV(0, false); else if (dot_mode == 1) SPRS_WSPMV(1, false); else SPRS_WSPMV(2, false);
            }
#undef SPRS_WSPMV
            SPRS_HIP_TRY(c, hipGetLastError());
            return SPRS_OK;
        }
    }
    const uint8_t *code = pair ? D->pair_code : D->idx_code;
    const int32_t *otab = pair ? D->pair_off : D->off_tab;
    if (!pair && order == nullptr && count == A->n_rowblk && c->spmv_period > 0 && D->off_order) { order = D->off_order; xcd_chunk = 0; }
    // the offset-code stream runs on its own descriptors (uniform blocks flagged); same block numbering as blk_desc
    const BlkDesc *dsc = reinterpret_cast<const BlkDesc *>((!pair && D->off_desc && c->spmv_uniform != 0) ? D->off_desc : A->blk_desc);
    const V2d *tail2 = nullptr;
    int g2_last = -1;
    if constexpr (sizeof(T) == 8 && !is_complex<T>::value) {
        if (!pair && A->tail && c->spmv_wideload != 0) {
            // f64 offset codes: 16-byte value loads (the plain stream's measure, profiles/r03_tuning.md §2)
            g2_last = (int)((A->nnz - 1) >> 1);
            tail2 = reinterpret_cast<const V2d *>(reinterpret_cast<const char *>(A->tail) + 16) + (g2_last - 2 * (int)((A->nnz - 1) >> 2));
            const bool whole = order == nullptr && count == A->n_rowblk;
            const bool interior = A->dist && A->dist->order_int && order == A->dist->order_int && count == A->dist->n_int && A->dist->tile_int.n_tile > 0;
            const sprs_tile_plan *tpp = nullptr;
            if (whole && D->tile_off.n_tile > 0) tpp = &D->tile_off;
            else if (interior && A->dist->tile_int_off) tpp = &A->dist->tile_int;
            if (tpp && D->owide_desc && c->spmv_tile != 0 && c->spmv_wide != 0 && c->spmv_uniform != 0 && g % 8 == 0 && !conj_x) {
                // LDS x-window tiles + the per-block walk over the 64-row blocks outside them, one launch
                const sprs_tile_plan &TP = *tpp;
                TilePat tp;
                for (int t = 0; t < 8; ++t) { tp.off[t] = TP.off[t]; tp.val[t] = 0.0; }
                const bool ux = dot_mode != 0 && u == x;      // the dot operand is the input vector (mul_vec_dot, MINRES' v.Av, K4 without a preconditioner)
                const BlkDesc *owd = reinterpret_cast<const BlkDesc *>(D->owide_desc);
#define SPRS_TOSPMV(DM, UXV, U, L, H) SPRS_LAUNCH_SPMV(c, (spmv_tile_off_kernel<DM, UXV, U, L, H>), g, reinterpret_cast<const int2 *>(TP.list), TP.xstart, owd, tp, \
                                                       TP.n_left, TP.left, dsc, A->row_ptr, code, otab, v, x, y, u, part0, part1, status, fin, tail2, g2_last)
#define SPRS_TOSHAPE(U, L, H)                                                                                            \
                if (TP.ul == U && TP.fl == L && TP.fh == H) {                                                            \
                    if (dot_mode == 0) SPRS_TOSPMV(0, false, U, L, H);                                                   \
                    else if (dot_mode == 1) { if (ux) SPRS_TOSPMV(1, true, U, L, H); else SPRS_TOSPMV(1, false, U, L, H); } \
                    else if (ux) SPRS_TOSPMV(2, true, U, L, H); else SPRS_TOSPMV(2, false, U, L, H);                     \
                }
                SPRS_TILE_SHAPES(SPRS_TOSHAPE)
#undef SPRS_TOSHAPE
#undef SPRS_TOSPMV
                SPRS_HIP_TRY(c, hipGetLastError());
                return SPRS_OK;
            }
#define SPRS_DSPMVW(DM) SPRS_LAUNCH_SPMV(c, (spmv_dict_kernel<T, DM, false, false, true>), g, count, xcd_chunk, dsc, order, A->row_ptr, \
                                         code, otab, pv, v, x, y, u, part0, part1, status, fin, tail2, g2_last)
            if (dot_mode == 0) SPRS_DSPMVW(0); else if (dot_mode == 1) SPRS_DSPMVW(1); else SPRS_DSPMVW(2);
#undef SPRS_DSPMVW
            SPRS_HIP_TRY(c, hipGetLastError());
            return SPRS_OK;
        }
    }
#define SPRS_DSPMV2(DM, CJ, PR)                                                                                         \
    SPRS_LAUNCH_SPMV(c, (spmv_dict_kernel<T, DM, CJ, PR>), g, count, xcd_chunk,                                          \
                       dsc, order, A->row_ptr, code, otab, pv, v, x, y, u,                                              \
                       part0, part1, status, fin, tail2, g2_last)
#define SPRS_DSPMV(DM, CJ) do { if (pair) SPRS_DSPMV2(DM, CJ, (has_val_dict<T>::value)); else SPRS_DSPMV2(DM, CJ, false); } while (0)
    if (conj_x && is_complex<T>::value) {
        if (dot_mode == 0) SPRS_DSPMV(0, true);
        else if (dot_mode == 1) SPRS_DSPMV(1, true);
        else SPRS_DSPMV(2, true);
    } else {
        if (dot_mode == 0) SPRS_DSPMV(0, false);
        else if (dot_mode == 1) SPRS_DSPMV(1, false);
        else SPRS_DSPMV(2, false);
    }
#undef SPRS_DSPMV2
#undef SPRS_DSPMV
    SPRS_HIP_TRY(c, hipGetLastError());
    return SPRS_OK;
}

#define SPRS_INST_DSPMV(T)                                                                                              \
    template int launch_spmv_dict<T>(const sprs_csr *, int, const int32_t *, int, int, int, const T *, T *, int, const T *, T *, T *, const int *, bool, const Fin &);
SPRS_INST_DSPMV(double)
SPRS_INST_DSPMV(cplx)
SPRS_INST_DSPMV(float)
SPRS_INST_DSPMV(cplxf)

}  // namespace sprs
