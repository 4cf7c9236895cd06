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

#include "spmv_dict_dev.hpp"
#include "minres_fuse.hpp"

namespace sprs {

namespace {

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

// ---- complex scalars: pair codes with a ROW-VALUE slot (round 4) -------------------------------------------------------------
// The value dictionaries above key a value by its 64-bit pattern: real scalars only.  Complex operators on grids — the
// reference's own complex tests (tests/test_complex_solve.rs:95-151, test_complex_solve2.rs:35-96): constant off-diagonals, a
// diagonal that differs from row to row; any shifted Laplacian / Helmholtz-like operator — repeat their OFF-DIAGONAL entries
// only.  So: the distinct (offset code, value) pairs of all entries but those at offset 0 are collected (<= 255 of them, matched
// by bit pattern), an offset-0 entry gets the reserved code 255 = "this row's own value", and those values are kept densely, one
// per row (`rowval`).  The SpMV then reads 1 B per entry + 16 B per row instead of 20 B per entry; the products and their order
// are the original ones: y bit-identical.  The collection needs no key wider than the machine's atomics: a slot of the table
// holds the INDEX of the first entry that claimed it, and candidates are compared with that entry's offset code and value bits.
constexpr int CP_SLOTS = 1024;
constexpr int CP_DIAG = 255;
template <class T> __device__ __forceinline__ bool same_bits(const T &a, const T &b) {
    static_assert(sizeof(T) % 8 == 0, "complex scalars: one or two 64-bit words");
    uint64_t wa[sizeof(T) / 8], wb[sizeof(T) / 8];
    __builtin_memcpy(wa, &a, sizeof(T)); __builtin_memcpy(wb, &b, sizeof(T));
    bool eq = true;
    for (size_t i = 0; i < sizeof(T) / 8; ++i) eq = eq && wa[i] == wb[i];
    return eq;
}
template <class T> __device__ __forceinline__ uint32_t cp_hash(int oc, const T &v) {
    uint64_t w[sizeof(T) / 8 ? sizeof(T) / 8 : 1] = {0};
    __builtin_memcpy(w, &v, sizeof(T));
    uint64_t h = (uint64_t)(oc + 1) * 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < sizeof(w) / 8; ++i) h = (h ^ w[i]) * 0xff51afd7ed558ccdull;
    return (uint32_t)(h ^ (h >> 29));
}
// count[0] = pairs claimed; a thread abandons the pass when it is about to claim one beyond CP_DIAG.  Loads of the same few table
// words from every thread of the chip serialise in L2 (3.4 ms on cfg 4), so every workgroup keeps an LDS mirror of the slots it has
// seen filled (a filled slot never changes) and ONE thread seeds it from the workgroup's first row before the others start: the
// steady state of a stencil — every entry already known — then costs LDS reads and L1 hits only.
template <class T>
__global__ __launch_bounds__(BLOCK) void cpair_collect_kernel(int n, const int32_t *__restrict__ row_ptr, const uint8_t *__restrict__ idx_code,
                                                              const T *__restrict__ val, int diag_code, int32_t *tab, int *count) {
    __shared__ int32_t s_tab[CP_SLOTS];
    for (int i = threadIdx.x; i < CP_SLOTS; i += BLOCK) s_tab[i] = -1;
    __syncthreads();
    auto visit = [&](int row) -> bool {                     // false: more than CP_DIAG pairs, give up
        for (int k = row_ptr[row], ke = row_ptr[row + 1]; k < ke; ++k) {
            const int oc = idx_code[k];
            if (oc == diag_code) continue;
            const T v = val[k];
            uint32_t h = cp_hash(oc, v) & (CP_SLOTS - 1);
            for (int t = 0; t < CP_SLOTS; ++t, h = (h + 1) & (CP_SLOTS - 1)) {
                int cur = s_tab[h];
                if (cur < 0) {
                    cur = __hip_atomic_load(tab + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur < 0) {
                        if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > CP_DIAG) return false;
                        const int old = atomicCAS(tab + h, -1, k);
                        if (old < 0) { atomicAdd(count, 1); s_tab[h] = k; break; }
                        cur = old;
                    }
                    s_tab[h] = cur;        // (racing writers store the same index)
                }
                if (idx_code[cur] == oc && same_bits(val[cur], v)) break;
            }
        }
        return true;
    };
    const int row0 = blockIdx.x * BLOCK;
    if (threadIdx.x == 0 && row0 + BLOCK / 2 < n) (void)visit(row0 + BLOCK / 2);
    __syncthreads();
    for (int row = row0 + threadIdx.x; row < n; row += gridDim.x * BLOCK)
        if (!visit(row)) return;
}
// the representatives' offset codes and values, slot by slot (slot empty: code 255)
template <class T>
__global__ __launch_bounds__(BLOCK) void cpair_gather_kernel(const int32_t *__restrict__ tab, const uint8_t *__restrict__ idx_code,
                                                             const T *__restrict__ val, uint8_t *__restrict__ rep_oc, T *__restrict__ rep_val) {
    const int sl = blockIdx.x * BLOCK + threadIdx.x;
    if (sl >= CP_SLOTS) return;
    const int k = tab[sl];
    rep_oc[sl] = k < 0 ? (uint8_t)255 : idx_code[k];
    rep_val[sl] = k < 0 ? szero<T>() : val[k];
}
template <class T>
__global__ __launch_bounds__(BLOCK) void cpair_encode_kernel(int n, const int32_t *__restrict__ row_ptr, const uint8_t *__restrict__ idx_code,
                                                             const T *__restrict__ val, int diag_code, const int32_t *__restrict__ tab,
                                                             const uint8_t *__restrict__ rep_oc, const T *__restrict__ rep_val,
                                                             const uint8_t *__restrict__ code_of_slot, uint8_t *__restrict__ pair_code,
                                                             T *__restrict__ rowval, int *__restrict__ bad) {
    __shared__ T s_val[CP_SLOTS];
    __shared__ uint8_t s_oc[CP_SLOTS], s_code[CP_SLOTS], s_used[CP_SLOTS];
    for (int i = threadIdx.x; i < CP_SLOTS; i += BLOCK) { s_val[i] = rep_val[i]; s_oc[i] = rep_oc[i]; s_code[i] = code_of_slot[i]; s_used[i] = tab[i] >= 0; }
    __syncthreads();
    for (int row = blockIdx.x * BLOCK + threadIdx.x; row < n; row += gridDim.x * BLOCK) {
        T dv = szero<T>();
        bool have_dv = false;
        for (int k = row_ptr[row], ke = row_ptr[row + 1]; k < ke; ++k) {
            const int oc = idx_code[k];
            const T v = val[k];
            if (oc == diag_code) {
                if (have_dv) *bad = 1;          // two entries at offset 0 (an unsorted/duplicated row): one row-value slot cannot hold both
                pair_code[k] = (uint8_t)CP_DIAG; dv = v; have_dv = true; continue;
            }
            uint32_t h = cp_hash(oc, v) & (CP_SLOTS - 1);
            int code = -1;
            for (int t = 0; t < CP_SLOTS; ++t, h = (h + 1) & (CP_SLOTS - 1)) {
                if (!s_used[h]) break;
                if (s_oc[h] == oc && same_bits(s_val[h], v)) { code = s_code[h]; break; }
            }
            if (code < 0) { *bad = 1; code = 0; }
            pair_code[k] = (uint8_t)code;
        }
        rowval[row] = dv;
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
                                                          const V2d *__restrict__ tail2, int g2_last, const T *__restrict__ rowval) {
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
    if constexpr (PAIR) stage_pair(s_pair, tid, off_tab[tid] * (int32_t)sizeof(T), val_tab[tid]);    // BLOCK == TAB
    else s_off8[tid] = off_tab[tid] * (int32_t)sizeof(T);
    for (int i = lane; i < CW; i += WAVE) s_c[wv][i] = 0;       // the pad is read (and ignored) before it is ever written
    if constexpr (!PAIR) for (int i = lane; i < CAP + 16; i += WAVE) s_v[wv][i] = szero<T>();
    __syncthreads();                                // the only workgroup barrier: tables are read-only afterwards
    if (run_state != ST_RUNNING) { fin_idle(fin, DOT == 2); return; }

    T d0 = szero<T>(), d1 = szero<T>();
    dict_walk<T, DOT, CONJX, PAIR, WV>(n_rowblk, xcd_chunk, desc, order, row_ptr, code, val, x, y, u, tail2, g2_last, s_pair, s_off8, s_c,
                                       &s_v[0][0], PAIR ? 0 : CAP + 16, d0, d1, rowval);
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

// M1 of MINRES' iteration k + 1 on the UN-NORMALISED v_new of iteration k (minres_fuse.hpp; krylov.hip "M3 deferred"): the launch runs
// M3's prologue for beta_new and multiplies A by x * (1 / beta_new) formed in the gathers (ScaleEw); the dot operand is the scaled
// row value.  x is not modified and nothing of the solver's state is written.  The stopping rule is M3's: a launch of a stopped
// solve returns at once.
template <class T, bool CONJX, bool PAIR, bool WV, class M3>
__global__ __launch_bounds__(BLOCK) void spmv_dict_scaled_kernel(int n_rowblk, int xcd_chunk, const BlkDesc *__restrict__ desc, const int32_t *__restrict__ order,
                                                                 const int32_t *__restrict__ row_ptr, const uint8_t *__restrict__ code,
                                                                 const int32_t *__restrict__ off_tab, const T *__restrict__ val_tab,
                                                                 const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y,
                                                                 T *__restrict__ part0, const V2d *__restrict__ tail2, int g2_last,
                                                                 const T *__restrict__ rowval, M3 m3) {
    static_assert(!WV || (sizeof(T) == 8 && !PAIR), "wide value loads: f64 offset-code stream");
    constexpr int CAP = nnz_cap<T>::value;
    constexpr int CW = (CAP + 3 + CPAD + 3) / 4;
    __shared__ PairEnt<T> s_pair[PAIR ? TAB : 1];
    __shared__ int32_t s_off8[PAIR ? 1 : TAB];
    __shared__ uint32_t s_c[NWAVE][CW];
    __shared__ __attribute__((aligned(16))) T s_v[PAIR ? 1 : NWAVE][PAIR ? 1 : CAP + 16];
    __shared__ T red[NWAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (PAIR) stage_pair(s_pair, tid, off_tab[tid] * (int32_t)sizeof(T), val_tab[tid]);    // BLOCK == TAB
    else s_off8[tid] = off_tab[tid] * (int32_t)sizeof(T);
    for (int i = lane; i < CW; i += WAVE) s_c[wv][i] = 0;
    if constexpr (!PAIR) for (int i = lane; i < CAP + 16; i += WAVE) s_v[wv][i] = szero<T>();
    __syncthreads();
    if (!m3.prologue()) return;                     // stopped (or an invalid hand-off): the same decision in every workgroup
    T d0 = szero<T>(), d1 = szero<T>();
    dict_walk<T, 1, CONJX, PAIR, WV, ScaleEw<T>>(n_rowblk, xcd_chunk, desc, order, row_ptr, code, val, x, y, x, tail2, g2_last, s_pair, s_off8, s_c,
                                                  &s_v[0][0], PAIR ? 0 : CAP + 16, d0, d1, rowval, ScaleEw<T>{m3.inv});
    d0 = block_sum(d0, red);
    if (tid == 0) part0[blockIdx.x] = d0;
}

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
    stage_pair(s_pair, tid, off_tab[tid] * 8, val_tab[tid]);                  // BLOCK == TAB
    for (int i = lane; i < CW2; i += WAVE) s_c[wv][i] = 0;                     // the pad is read (and ignored) before it is written
    __syncthreads();
    if (run_state != ST_RUNNING) { fin_idle(fin, DOT == 2); return; }

    T d0 = 0.0, d1 = 0.0;
    pair2_walk<DOT, YNT>(n_wide, xcd_chunk, desc, order, row_ptr, code, XPlain{reinterpret_cast<const char *>(x)}, y, u, nrows, ncols, s_pair, s_c, d0, d1);
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
// ... a pair-code stream: real scalars through the value dictionary, complex ones through the row-value slot (cpair stage)
template <class T> struct has_pair_codes { static constexpr bool value = has_val_dict<T>::value || is_complex<T>::value; };

// Chain plan (spmv_chain.hip) of the pair-code stream: flag[j] = 128-row block j is a full uniform (plain or seam) block of the
// canonical pattern off[0..UL), which has exactly one far slot a side.  Tiles of CH_B blocks are placed greedily on the runs of
// flagged blocks; a tile at row ts is linked to the tile that starts within 127 rows of ts + Pf (runs in consecutive planes start
// at their own first full block, so the 128-row grid drifts against the plane by a few rows per plane); the linked lists are the
// chains.  Every chain is cut into equal segments so that all chains together give about one segment per workgroup (2 per CU:
// three windows are 72 KiB of LDS), XCD x takes a contiguous eighth of the chains ordered by their column position (neighbouring
// columns share the 2W window margins through that XCD's L2), segment-of-all-its-chains by segment.  The plan is dropped unless it
// covers most of the flagged blocks in segments long enough to amortise their two priming windows and fills the chip.
template <class MakeLeft>
static int build_chain_plan(sprs_csr *A, sprs_chain_plan &CP, const std::vector<BlkDescHost2> &wd, const std::vector<uint8_t> &flag, int UL,
                            const int32_t *off, const double *val, MakeLeft &&make_left) {
    sprs_ctx *c = A->ctx;
    const int nw = (int)wd.size();
    const int64_t Pf = off[UL - 1];
    if (UL < 3 || UL > 7 || (UL & 1) == 0 || -(int64_t)off[0] != Pf || (Pf & 1) || Pf < 2 * (int64_t)TILE_W) return SPRS_OK;
    for (int t = 1; t + 1 < UL; ++t) if (off[t] < -(TILE_W - 2) || off[t] > TILE_W - 2) return SPRS_OK;     // far, near.., far on the narrow window
    bool shape_ok = false;
#define SPRS_CH_HAS(U, TR) if (UL == U) shape_ok = true;
    SPRS_CHAIN_SHAPES(SPRS_CH_HAS)
#undef SPRS_CH_HAS
    if (!shape_ok) return SPRS_OK;
    const int CHR = CH_B * 2 * WAVE;
    // TRI: near slots (.., c - 1, c, c + 1, ..) around an even centre, every other near offset even
    const int NN = UL - 2, TC = NN / 2;
    bool tri = (off[1 + TC] & 1) == 0;
    for (int t = 0; t < NN && tri; ++t) {
        if (t == TC - 1) tri = off[1 + t] == off[1 + TC] - 1;
        else if (t == TC + 1) tri = off[1 + t] == off[1 + TC] + 1;
        else tri = (off[1 + t] & 1) == 0;
    }
    if (!tri && UL == 3) return SPRS_OK;                                  // (only the 16-byte shape is built for UL = 3)
    std::vector<int32_t> tb;                                              // first block of each tile, row order
    std::vector<char> in_chain((size_t)nw, 0);
    int64_t flagged = 0;
    for (int j = 0; j < nw;) {
        if (!flag[(size_t)j]) { ++j; continue; }
        int e = j + 1;
        while (e < nw && flag[(size_t)e] && wd[(size_t)e].ra == wd[(size_t)e - 1].ra + 2 * WAVE) ++e;
        flagged += e - j;
        for (int b = j; b + CH_B <= e; b += CH_B) {
            const int64_t ra = wd[(size_t)b].ra;
            if (ra - Pf < 0 || ra + CHR + Pf > A->ncols) continue;        // the -Pf / +Pf centres lie inside x (they do for every flagged block; belt and braces)
            tb.push_back(b);
        }
        j = e;
    }
    const int nt = (int)tb.size();
    if (nt < 64) return SPRS_OK;
    // links: tile i -> the tile that starts within 127 rows of ra_i + Pf
    std::vector<int32_t> next((size_t)nt, -1), prev((size_t)nt, -1);
    {
        std::vector<int64_t> ras((size_t)nt);
        for (int i = 0; i < nt; ++i) ras[(size_t)i] = wd[(size_t)tb[(size_t)i]].ra;
        for (int i = 0; i < nt; ++i) {
            const int64_t want = ras[(size_t)i] + Pf;
            const auto it = std::lower_bound(ras.begin(), ras.end(), want - 127);
            if (it == ras.end() || *it > want + 127) continue;
            const int j2 = (int)(it - ras.begin());
            if (prev[(size_t)j2] >= 0) continue;
            next[(size_t)i] = j2; prev[(size_t)j2] = i;
        }
    }
    struct Chain { int head; int len; int64_t col; };
    std::vector<Chain> chains;
    for (int i = 0; i < nt; ++i) {
        if (prev[(size_t)i] >= 0) continue;
        int len = 0;
        for (int q = i; q >= 0; q = next[(size_t)q]) ++len;
        chains.push_back(Chain{i, len, (int64_t)wd[(size_t)tb[(size_t)i]].ra % Pf});
    }
    std::stable_sort(chains.begin(), chains.end(), [](const Chain &a, const Chain &b) { return a.col < b.col; });
    const int nch = (int)chains.size();
    const int G = 2 * c->num_cu;                                          // workgroups of the chain launch
    // segments: q per chain, about G / nch, none shorter than 8 tiles (two priming windows each)
    std::vector<int32_t> tiles, segs, xstart(9, 0);
    int64_t covered = 0;
    int n_seg = 0;
    for (int xq = 0; xq < 8; ++xq) {
        xstart[(size_t)xq] = n_seg;
        const int c_lo = (int)((int64_t)nch * xq / 8), c_hi = (int)((int64_t)nch * (xq + 1) / 8);
        std::vector<std::vector<std::pair<int, int>>> cs((size_t)(c_hi - c_lo));   // per chain: (first position in chain, count) of its segments
        size_t maxq = 0;
        for (int ci = c_lo; ci < c_hi; ++ci) {
            const int len = chains[(size_t)ci].len;
            int q = std::max(1, G / std::max(nch, 1));
            q = std::max(1, std::min(q, len / 8));
            for (int k = 0; k < q; ++k) {
                const int a = (int)((int64_t)len * k / q), b = (int)((int64_t)len * (k + 1) / q);
                if (b > a) cs[(size_t)(ci - c_lo)].push_back({a, b - a});
            }
            maxq = std::max(maxq, cs[(size_t)(ci - c_lo)].size());
        }
        // flatten every chain of this XCD once: tile ids in chain order
        std::vector<std::vector<int32_t>> ids((size_t)(c_hi - c_lo));
        for (int ci = c_lo; ci < c_hi; ++ci)
            for (int q = chains[(size_t)ci].head; q >= 0; q = next[(size_t)q]) ids[(size_t)(ci - c_lo)].push_back(q);
        for (size_t k = 0; k < maxq; ++k)
            for (int ci = c_lo; ci < c_hi; ++ci) {
                const auto &sv = cs[(size_t)(ci - c_lo)];
                if (k >= sv.size()) continue;
                const auto &idv = ids[(size_t)(ci - c_lo)];
                segs.push_back((int32_t)(tiles.size() / 4)); segs.push_back(sv[k].second);
                for (int p2 = sv[k].first; p2 < sv[k].first + sv[k].second; ++p2) {
                    const int ti = idv[(size_t)p2];
                    const int b0 = tb[(size_t)ti];
                    const int32_t ra = wd[(size_t)b0].ra;
                    const int32_t ra_next = next[(size_t)ti] >= 0 ? wd[(size_t)tb[(size_t)next[(size_t)ti]]].ra : (int32_t)(ra + Pf);
                    tiles.push_back(b0); tiles.push_back(ra); tiles.push_back(ra_next); tiles.push_back(0);
                    for (int qb = 0; qb < CH_B; ++qb) in_chain[(size_t)(b0 + qb)] = 1;
                    covered += CH_B;
                }
                ++n_seg;
            }
    }
    xstart[8] = n_seg;
    const int n_tile = (int)(tiles.size() / 4);
    // worth it?  (forced by knob 1 whenever chains exist)
    const bool fills = n_seg * 2 >= G && covered * 2 >= flagged && n_tile >= 6 * n_seg;
    if (n_tile == 0 || (c->spmv_chain < 0 && !fills)) return SPRS_OK;
    std::vector<int32_t> left;
    make_left(in_chain, left);
    auto drop = [&]() { for (void *q : {(void *)CP.tiles, (void *)CP.segs, (void *)CP.xstart, (void *)CP.left}) if (q) (void)hipFree(q); CP = sprs_chain_plan(); };
#define CH_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { snprintf(c->err, sizeof(c->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); drop(); return SPRS_ERR_HIP; } } while (0)
    CH_TRY(hipMalloc((void **)&CP.tiles, sizeof(int32_t) * tiles.size()));
    CH_TRY(hipMalloc((void **)&CP.segs, sizeof(int32_t) * segs.size()));
    CH_TRY(hipMalloc((void **)&CP.xstart, sizeof(int32_t) * 9));
    CH_TRY(hipMalloc((void **)&CP.left, sizeof(int32_t) * std::max<size_t>(left.size(), 1)));
    CH_TRY(hipMemcpyAsync(CP.tiles, tiles.data(), sizeof(int32_t) * tiles.size(), hipMemcpyHostToDevice, c->stream));
    CH_TRY(hipMemcpyAsync(CP.segs, segs.data(), sizeof(int32_t) * segs.size(), hipMemcpyHostToDevice, c->stream));
    CH_TRY(hipMemcpyAsync(CP.xstart, xstart.data(), sizeof(int32_t) * 9, hipMemcpyHostToDevice, c->stream));
    if (!left.empty()) CH_TRY(hipMemcpyAsync(CP.left, left.data(), sizeof(int32_t) * left.size(), hipMemcpyHostToDevice, c->stream));
    CH_TRY(hipStreamSynchronize(c->stream));
#undef CH_TRY
    CP.n_tile = n_tile; CP.n_seg = n_seg; CP.n_chain = nch; CP.n_left = (int)left.size();
    CP.ul = UL; CP.tri = tri ? 1 : 0;
    for (int t = 0; t < 8; ++t) { CP.off[t] = t < UL ? off[t] : 0; CP.val[t] = t < UL ? val[t] : 0.0; }
    return SPRS_OK;
}

// Tile plan of a stream's 128-row descriptors `desc_dev` (marked by mark_uniform_kernel on the codes `code_dev`; host copy of
// the unmarked descriptors: wd).  off_tab / val_tab: host tables per code (val_tab null: a stream whose values are per entry).
// make_left(in_tile, left) lists the blocks the tiles do not cover, in the walk order of the stream's per-block kernel.
template <class MakeLeft>
static int build_tile_plan(sprs_csr *A, sprs_tile_plan &TP, const BlkDesc *desc_dev, const uint8_t *code_dev, const std::vector<BlkDescHost2> &wd,
                           const int32_t *off_tab, const double *val_tab, bool wide_window_ok, MakeLeft &&make_left, sprs_chain_plan *CP = nullptr) {
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
    if (CP && c->spmv_chain != 0 && val_tab) {
        // plane-streaming chains (spmv_chain.hip) on the same runs; the tile plan below is kept beside them (knob, distributed interior)
        if (const int st = build_chain_plan(A, *CP, wd, flag, UL, TP.off, TP.val, make_left)) { drop(); return st; }
    }
    std::vector<int32_t> starts;                            // first block of each tile, in row order
    std::vector<char> in_tile((size_t)nw, 0);
    for (int j = 0; j < nw;) {
        if (!flag[(size_t)j]) { ++j; continue; }
        int e = j + 1;
        while (e < nw && flag[(size_t)e] && wd[(size_t)e].ra == wd[(size_t)e - 1].ra + 2 * WAVE) ++e;
        int b = j;
        // every 16-byte load of a tile stays inside x — the near window [ra - WIN, ra + TILE_ROWS + WIN) and each far slot's
        // pairs x[r + o], x[r + o + 1] for ALL the tile's rows: the kernels load them for every lane, also for a seam row whose
        // mask lacks that slot (damaged-row or band matrices can put such a row where r + o leaves [0, ncols))
        int64_t lo_need = WIN, hi_need = WIN;
        for (int t = 0; t < FL; ++t) lo_need = std::max<int64_t>(lo_need, -(int64_t)TP.off[t]);
        for (int t = UL - FH; t < UL; ++t) hi_need = std::max<int64_t>(hi_need, (int64_t)TP.off[t]);
        while (b < e && (int64_t)wd[(size_t)b].ra - lo_need < 0) ++b;                // the window / far pairs start inside x
        for (; b + TILE_B <= e; b += TILE_B) {
            if ((int64_t)wd[(size_t)b].ra + TILE_ROWS + hi_need > A->ncols) break;    // ... and end inside it
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
    if constexpr (is_complex<T>::value) {
        // ---- complex pair stage (cpair_* kernels above): (offset code, value) pairs of the entries off the diagonal, <= 255 of
        // them, + one value per row for the offset-0 entries
        if (c->spmv_dict != 1 && A->nnz > 0) {
            int diag_code = -1;
            for (size_t i = 0; i < offs.size(); ++i) if (offs[i].first == 0) diag_code = (int)i;
            int32_t *cp_tab = nullptr; uint8_t *cp_oc = nullptr, *cp_code = nullptr; T *cp_val = nullptr; int *cp_cnt = counts + 1;
            auto cp_free = [&]() { for (void *q : {(void *)cp_tab, (void *)cp_oc, (void *)cp_code, (void *)cp_val}) if (q) (void)hipFree(q); };
#define CP_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { snprintf(c->err, sizeof(c->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); cp_free(); cleanup2(); free_dict(A); return SPRS_ERR_HIP; } } while (0)
            CP_TRY(hipMalloc((void **)&cp_tab, sizeof(int32_t) * CP_SLOTS));
            CP_TRY(hipMalloc((void **)&cp_oc, CP_SLOTS));
            CP_TRY(hipMalloc((void **)&cp_code, CP_SLOTS));
            CP_TRY(hipMalloc((void **)&cp_val, sizeof(T) * CP_SLOTS));
            CP_TRY(hipMemsetAsync(cp_tab, 0xff, sizeof(int32_t) * CP_SLOTS, c->stream));
            CP_TRY(hipMemsetAsync(cp_cnt, 0, sizeof(int) * 3, c->stream));
            hipLaunchKernelGGL((cpair_collect_kernel<T>), dim3(g), dim3(BLOCK), 0, c->stream, n, A->row_ptr, D->idx_code, val, diag_code, cp_tab, cp_cnt);
            CP_TRY(hipGetLastError());
            int np = 0;
            CP_TRY(hipMemcpyAsync(&np, cp_cnt, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            CP_TRY(hipStreamSynchronize(c->stream));
            if (np <= CP_DIAG) {
                hipLaunchKernelGGL((cpair_gather_kernel<T>), dim3(CP_SLOTS / BLOCK), dim3(BLOCK), 0, c->stream, cp_tab, D->idx_code, val, cp_oc, cp_val);
                CP_TRY(hipGetLastError());
                std::vector<int32_t> h_tab(CP_SLOTS);
                std::vector<uint8_t> h_oc(CP_SLOTS), h_code(CP_SLOTS, 0);
                std::vector<T> h_val(CP_SLOTS);
                CP_TRY(hipMemcpyAsync(h_tab.data(), cp_tab, sizeof(int32_t) * CP_SLOTS, hipMemcpyDeviceToHost, c->stream));
                CP_TRY(hipMemcpyAsync(h_oc.data(), cp_oc, CP_SLOTS, hipMemcpyDeviceToHost, c->stream));
                CP_TRY(hipMemcpyAsync(h_val.data(), cp_val, sizeof(T) * CP_SLOTS, hipMemcpyDeviceToHost, c->stream));
                CP_TRY(hipStreamSynchronize(c->stream));
                // codes in ascending (offset, value bits) order: deterministic whatever thread claimed a slot first
                std::vector<int> slots;
                for (int sl = 0; sl < CP_SLOTS; ++sl) if (h_tab[(size_t)sl] >= 0) slots.push_back(sl);
                std::sort(slots.begin(), slots.end(), [&](int a, int b) {
                    const int32_t oa = off_tab[h_oc[(size_t)a]], ob = off_tab[h_oc[(size_t)b]];
                    if (oa != ob) return oa < ob;
                    return memcmp(&h_val[(size_t)a], &h_val[(size_t)b], sizeof(T)) < 0;
                });
                if ((int)slots.size() == np) {
                    std::vector<int32_t> pair_off(TAB, 0);
                    std::vector<T> pair_val(TAB, szero<T>());
                    for (size_t i = 0; i < slots.size(); ++i) {
                        h_code[(size_t)slots[i]] = (uint8_t)i;
                        pair_off[i] = off_tab[h_oc[(size_t)slots[i]]]; pair_val[i] = h_val[(size_t)slots[i]];
                    }
                    CP_TRY(hipMalloc((void **)&D->pair_code, nb));
                    CP_TRY(hipMemsetAsync(D->pair_code, 0, nb, c->stream));
                    CP_TRY(hipMalloc((void **)&D->pair_off, sizeof(int32_t) * TAB));
                    CP_TRY(hipMalloc(&D->pair_val, sizeof(T) * TAB));
                    CP_TRY(hipMalloc(&D->rowval, sizeof(T) * (size_t)std::max(n, 1)));
                    CP_TRY(hipMemcpyAsync(D->pair_off, pair_off.data(), sizeof(int32_t) * TAB, hipMemcpyHostToDevice, c->stream));
                    CP_TRY(hipMemcpyAsync(D->pair_val, pair_val.data(), sizeof(T) * TAB, hipMemcpyHostToDevice, c->stream));
                    CP_TRY(hipMemcpyAsync(cp_code, h_code.data(), CP_SLOTS, hipMemcpyHostToDevice, c->stream));
                    CP_TRY(hipMemsetAsync(bad, 0, sizeof(int), c->stream));
                    hipLaunchKernelGGL((cpair_encode_kernel<T>), dim3(g), dim3(BLOCK), 0, c->stream, n, A->row_ptr, D->idx_code, val, diag_code, cp_tab,
                                       cp_oc, cp_val, cp_code, D->pair_code, reinterpret_cast<T *>(D->rowval), bad);
                    CP_TRY(hipGetLastError());
                    int h_bad = 0;
                    CP_TRY(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                    CP_TRY(hipStreamSynchronize(c->stream));
                    if (h_bad) {        // a row with two entries at offset 0: keep the offset-code stream
                        for (void **q : {(void **)&D->pair_code, (void **)&D->pair_off, &D->pair_val, &D->rowval}) { (void)hipFree(*q); *q = nullptr; }
                    } else {
                        D->n_pair = np + (diag_code >= 0 ? 1 : 0);
                    }
                }
            }
            cp_free();
#undef CP_TRY
            tr.lap("    complex pair codes (row-value slot)");
        }
    }
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
                        }, &D->chain_pair)) { cleanup2(); free_dict(A); return st; }
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

bool chain_plan_used(const sprs_csr *A) {
    const sprs_ctx *c = A->ctx;
    if (!A->dict || A->dist || c->spmv_chain == 0 || c->spmv_tile == 0 || c->spmv_wide == 0 || c->spmv_uniform == 0) return false;
    return dict_mode(A) == 2 && A->dict->chain_pair.n_tile > 0;
}

void free_dict(sprs_csr *A) {
    if (!A || !A->dict) return;
    sprs_dict *D = A->dict;
    for (void *q : {(void *)D->idx_code, (void *)D->pair_code, (void *)D->off_tab, (void *)D->pair_off, D->pair_val, D->wide_desc, D->off_desc, (void *)D->wide_order, (void *)D->off_order,
                    (void *)D->tile_pair.list, (void *)D->tile_pair.xstart, (void *)D->tile_pair.left,
                    (void *)D->tile_off.list, (void *)D->tile_off.xstart, (void *)D->tile_off.left, D->owide_desc,
                    (void *)D->chain_pair.tiles, (void *)D->chain_pair.segs, (void *)D->chain_pair.xstart, (void *)D->chain_pair.left, D->rowval})
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
    const bool pair = has_pair_codes<T>::value && mode == 2;
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
        if (pair && whole && g % 8 == 0 && chain_plan_used(A))      // plane-streaming chains + the per-block walk over the blocks outside them, one launch (spmv_chain.hip)
            return launch_chain_pair(A, D->chain_pair, g, x, y, dot_mode, u, part0, part1, status, fin);
        if (tpp && c->spmv_tile != 0 && c->spmv_wide != 0 && g % 8 == 0) {
            // LDS x-window tiles + the per-block walk over the blocks outside them, one launch (spmv_tile.hip)
            return launch_tile_pair(A, *tpp, g, x, y, dot_mode, u, part0, part1, status, fin);
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
                // LDS x-window tiles + the per-block walk over the 64-row blocks outside them, one launch (spmv_tile_off.hip)
                return launch_tile_off(A, *tpp, g, dsc, x, y, dot_mode, u, part0, part1, status, fin, tail2, g2_last);
            }
#define SPRS_DSPMVW(DM) SPRS_LAUNCH_SPMV(c, (spmv_dict_kernel<T, DM, false, false, true>), g, count, xcd_chunk, dsc, order, A->row_ptr, \
                                         code, otab, pv, v, x, y, u, part0, part1, status, fin, tail2, g2_last, (const T *)nullptr)
            if (dot_mode == 0) SPRS_DSPMVW(0); else if (dot_mode == 1) SPRS_DSPMVW(1); else SPRS_DSPMVW(2);
#undef SPRS_DSPMVW
            SPRS_HIP_TRY(c, hipGetLastError());
            return SPRS_OK;
        }
    }
#define SPRS_DSPMV2(DM, CJ, PR)                                                                                         \
    SPRS_LAUNCH_SPMV(c, (spmv_dict_kernel<T, DM, CJ, PR>), g, count, xcd_chunk,                                          \
                       dsc, order, A->row_ptr, code, otab, pv, v, x, y, u,                                              \
                       part0, part1, status, fin, tail2, g2_last, reinterpret_cast<const T *>(D->rowval))
#define SPRS_DSPMV(DM, CJ) do { if (pair) SPRS_DSPMV2(DM, CJ, (has_pair_codes<T>::value)); else SPRS_DSPMV2(DM, CJ, false); } while (0)
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

// ---- M3 deferred (spmv_dict_scaled_kernel): which handles, and the launch with the arguments launch_spmv_dict would take for the
// whole matrix in natural order
bool spmv_scaled_available(const sprs_csr *A) {
    const sprs_ctx *c = A->ctx;
    if (!A->dict || A->dist || c->spmv_fuse == 0) return false;
    const int dm = dict_mode(A);
    if (A->dtype == DT_D) return dm == 1 && !tile_plan_used(A) && A->tail != nullptr && c->spmv_wideload != 0;     // f64 offset codes, 16-byte value loads
    if (A->dtype == DT_Z) return dm == 1 || dm == 2;
    return false;
}

template <class T, bool SAUNDERS>
int launch_spmv_scaled(const sprs_csr *A, const MinresM3<T, false, SAUNDERS> &m3, const T *raw, T *y, T *partAlpha) {
    sprs_ctx *c = A->ctx;
    const sprs_dict *D = A->dict;
    const int dm = dict_mode(A);
    const bool pair = has_pair_codes<T>::value && dm == 2;
    const int g = spmv_num_partials(A);
    int xcd_chunk = c->xcd_chunk < 0 ? (is_cache_resident(A) ? 1 : 0) : c->xcd_chunk;
    const int32_t *order = nullptr;
    const int count = A->n_rowblk;
    const uint8_t *code = pair ? D->pair_code : D->idx_code;
    const int32_t *otab = pair ? D->pair_off : D->off_tab;
    if (!pair && c->spmv_period > 0 && D->off_order) { order = D->off_order; xcd_chunk = 0; }
    const BlkDesc *dsc = reinterpret_cast<const BlkDesc *>((!pair && D->off_desc && c->spmv_uniform != 0) ? D->off_desc : A->blk_desc);
    const T *v = reinterpret_cast<const T *>(A->val);
    const T *pv = reinterpret_cast<const T *>(D->pair_val);
    typedef MinresM3<T, false, SAUNDERS> M3;
    constexpr bool CJ = SAUNDERS && is_complex<T>::value;      // CSMINRES multiplies A by conj(q) (cs_minres.rs:99)
    if constexpr (sizeof(T) == 8 && !is_complex<T>::value) {
        const int g2_last = (int)((A->nnz - 1) >> 1);
        const V2d *tail2 = reinterpret_cast<const V2d *>(reinterpret_cast<const char *>(A->tail) + 16) + (g2_last - 2 * (int)((A->nnz - 1) >> 2));
        SPRS_LAUNCH_SPMV(c, (spmv_dict_scaled_kernel<T, false, false, true, M3>), g, count, xcd_chunk, dsc, order, A->row_ptr, code,
                         otab, pv, v, raw, y, partAlpha, tail2, g2_last, (const T *)nullptr, m3);
    } else {
#define SPRS_M3L(PR) SPRS_LAUNCH_SPMV(c, (spmv_dict_scaled_kernel<T, CJ, PR, false, M3>), g, count, xcd_chunk, dsc, order, A->row_ptr, code, otab, pv, v, \
                                      raw, y, partAlpha, (const V2d *)nullptr, -1, reinterpret_cast<const T *>(D->rowval), m3)
        if (pair) SPRS_M3L((has_pair_codes<T>::value)); else SPRS_M3L(false);
#undef SPRS_M3L
    }
    SPRS_HIP_TRY(c, hipGetLastError());
    return SPRS_OK;
}
template int launch_spmv_scaled<double, false>(const sprs_csr *, const MinresM3<double, false, false> &, const double *, double *, double *);
template int launch_spmv_scaled<cplx, false>(const sprs_csr *, const MinresM3<cplx, false, false> &, const cplx *, cplx *, cplx *);
template int launch_spmv_scaled<cplx, true>(const sprs_csr *, const MinresM3<cplx, false, true> &, const cplx *, cplx *, cplx *);

#define SPRS_INST_DSPMV(T)                                                                                              \
    template int launch_spmv_dict<T>(const sprs_csr *, int, const int32_t *, int, int, int, const T *, T *, int, const T *, T *, T *, const int *, bool, const Fin &);
SPRS_INST_DSPMV(double)
SPRS_INST_DSPMV(cplx)
SPRS_INST_DSPMV(float)
SPRS_INST_DSPMV(cplxf)

}  // namespace sprs
