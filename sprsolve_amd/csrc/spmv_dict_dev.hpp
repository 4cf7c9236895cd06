// Device-side building blocks of the compressed-stream SpMV kernels, shared by their translation units: spmv_dict.hip (creation,
// the per-block kernels, the launch dispatcher), spmv_tile.hip (LDS x-window tiles of the f64 pair-code stream) and
// spmv_tile_off.hip (the same for the f64 offset-code stream).  Split out of spmv_dict.hip in round 4 so that the three
// compile in parallel (profiles/r04_tuning.md "build cost"); the code is unchanged.
#pragma once
#include "device.hpp"

namespace sprs {

struct alignas(16) V2d { double a, b; };   // two consecutive f64 values of the offset-code stream (16-byte value loads)

namespace {

constexpr int TAB = 256;          // dictionary entries (one byte per code)

// ---------------------------------------------------------------------------------------------------------
constexpr int CPAD = 16;    // readable bytes behind a block's codes: the row phase reads up to 7 + 3 bytes past them

// LDS table entry of the pair stream: byte offset of the column relative to the row, and the value
template <class T> struct alignas(sizeof(T) >= 8 ? 16 : 8) PairEnt { int32_t off8; T val; };
// Filling a table entry: two member stores.  An aggregate store (`s_pair[i] = PairEnt<T>{...}`) also copies the padding, and for
// the 32-byte complex<double> entry the compiler stages that copy through a per-thread LDS temporary: measured 13 us on every launch
// of a 256-thread kernel on gfx950 (profiles/r04_tuning.md §6).
template <class T>
__device__ __forceinline__ void stage_pair(PairEnt<T> *s_pair, int i, int32_t off8, T val) {
    s_pair[i].off8 = off8;
    s_pair[i].val = val;
}

// What a wavefront loads for one row block before it can work on it.  The loads of block i+1 are issued
// before block i is processed (and the descriptor of block i+2 before that), so a block costs one exposed
// memory round trip — its x gather — instead of three dependent ones (descriptor -> codes/row_ptr -> x).
template <class T, bool PAIR, int ITEMS>
struct BlkLoads {
    int ra, rb, pa, nn;      // descriptor (nn = entries of the block, also for uniform blocks)
    int ulen;                // > 0: uniform block — every row repeats the first row's ulen (<= UNI_OFF_MAXLEN) codes; no row_ptr, 1-9 code dwords
    int s;                   // row_ptr[row] of this lane's row
    T uu;                    // dot operand of this lane's row
    T rv;                    // complex pair codes: this lane's row value (code 255; spmv_dict.hip, cpair stage)
    uint32_t wc[2];          // code dwords
    int di[2];               // ... and the LDS slots they go to
    T vv[PAIR ? 1 : ITEMS];  // values (offset-code stream only)
};

// A scaled operand for the lane-per-row walk ("M3 deferred", minres_fuse.hpp / krylov.hip): the SpMV multiplies by x[c] * scale —
// MINRES' v_new / beta_new, which is then never stored — with the product M3 would have stored (smulr), in the gathers and for the
// dot operand of the lane's own row.
struct NoScale { static constexpr bool ON = false; };
template <class T>
struct ScaleEw {
    static constexpr bool ON = true;
    Real<T> scale;
};

// WV (f64 offset codes only): the block's values are read with 16 bytes per lane over its 16-byte-aligned window (entries
// 2l, 2l + 1 of [pa - (pa & 1), ..) per load: 4 loads per 512 entries instead of 8) and staged to LDS with 16-byte stores;
// the last 2-entry group of val, which may reach one entry past the array, comes from the handle's zero-padded tail copy.
// The walk of the 64-row-block kernels of the compressed streams over `n_rowblk` blocks (positions of `order`, or natural
// order), shared by spmv_dict_kernel (the whole matrix) and spmv_tile_kernel's offset-code flavour (the blocks outside its
// tiles).  s_pair / s_off8: the staged tables; s_c: NWAVE zeroed code slices of CW dwords; s_v: NWAVE zeroed value slices of
// s_v_stride (>= CAP + 16) entries, 16-byte aligned (offset-code stream).  d0 / d1: the lane's running dot partials.
template <class T, int DOT, bool CONJX, bool PAIR, bool WV, class EW = NoScale>
__device__ __forceinline__ void dict_walk(int n_rowblk, int xcd_chunk, const BlkDesc *__restrict__ desc, const int32_t *__restrict__ order,
                                          const int32_t *__restrict__ row_ptr, const uint8_t *__restrict__ code,
                                          const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y, const T *__restrict__ u,
                                          const V2d *__restrict__ tail2, int g2_last,
                                          const PairEnt<T> *s_pair, const int32_t *s_off8, uint32_t (*s_c)[(nnz_cap<T>::value + 3 + CPAD + 3) / 4],
                                          T *s_v, int s_v_stride, T &d0, T &d1, const T *__restrict__ rowval = nullptr, const EW ew = EW{}) {
    static_assert(!EW::ON || DOT == 1, "the scaled operand is also the dot operand: u == x");
    constexpr int CAP = nnz_cap<T>::value;          // nnz per row block (per wavefront)
    constexpr bool RV = PAIR && is_complex<T>::value;      // pair code 255 = the row's own value
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
        if constexpr (RV) L.rv = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(rowval) + (uint32_t)rcl * (uint32_t)sizeof(T));
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
    [[maybe_unused]] T c_rv = szero<T>();
    auto adopt = [&](const Loads &L) {
        if constexpr (RV) c_rv = L.rv;
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
                if constexpr (PAIR) {
                    const PairEnt<T> e = s_pair[cd]; off8 = e.off8; av[t] = e.val;
                    if constexpr (RV) { if (cd == 255) av[t] = c_rv; }
                }
                else { off8 = s_off8[cd]; av[t] = vs[(WV ? c_vsh : 0) + kb + t]; }
                const uint32_t vo = valid ? r8 + (uint32_t)off8 : 0u;     // lanes past their row gather x[0] and drop it
                xg[t] = *reinterpret_cast<const T *>(xbytes + vo);
            }
            // all 8 gathers go out before the first product is formed (the scheduler otherwise hoists the
            // first multiply between them and with it a wait for the first gather: two round trips per block)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (j0 + t < len) {
                    T xv = xg[t];
                    if constexpr (EW::ON) xv = smulr(xv, ew.scale);          // the normalised operand, as M3 would have stored it
                    acc = sadd(acc, smul(CONJX ? sconj(xv) : xv, av[t]));
                }
        }
        if (r < c_rb) {
            *reinterpret_cast<T *>(reinterpret_cast<char *>(y) + r8) = acc;
            if constexpr (EW::ON) c_uu = smulr(c_uu, ew.scale);                                  // minres.rs:121
            if (DOT == 1) d0 = sadd(d0, smul(sconj(c_uu), acc));
            if (DOT == 2) { d0 = sadd(d0, smul(sconj(acc), acc)); d1 = sadd(d1, smul(sconj(acc), c_uu)); }
        }
        wave_lds_fence();   // the row phase is done with the LDS slice: refill it for the next block
        if (more) { stage(nxt); adopt(nxt); }
        dn = uniform(dn2); o2 = __builtin_amdgcn_readfirstlane(o3);
    }
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

// Where the two-rows-per-lane kernels take x from.  XPlain: a vector in memory.  XFused<NV>: x is the RESULT of the vector update
// that precedes the SpMV in BiCGStab's recurrence, formed on the fly from that update's operands with the update's own rounding
// sequence (krylov.hip, "fused SpMV input": K3 into K4, s = r + v * (-alpha), bicg_stab.rs:172; K1 into K2,
// p = (v * (-beta w) + p * beta) + r * 1, bicg_stab.rs:155-156) — the updated vector is never read back from memory.
// ld2(soff, voff): the pair at byte offset soff (wave-uniform) + voff (per lane); ld1: one element; lds: one element at a
// wave-uniform offset (a scalar load where the compiler can prove it).
__device__ __forceinline__ double comb_k3(double r, double v, double na) { return r + v * na; }
__device__ __forceinline__ double comb_k1(double v, double p, double r, double a, double beta) { double t = v * a + p * beta; t = t + r * 1.0; return t; }
struct XPlain {
    static constexpr bool FUSED = false;
    const char *xb;
    __device__ __forceinline__ D2 ld2(int64_t soff, uint32_t voff) const { return *reinterpret_cast<const D2 *>(xb + soff + voff); }
    __device__ __forceinline__ double ld1(uint32_t voff) const { return *reinterpret_cast<const double *>(xb + voff); }
    __device__ __forceinline__ double lds(int64_t soff) const { return *reinterpret_cast<const double *>(xb + soff); }
};
template <int NV>
struct XFused {
    static constexpr bool FUSED = true;
    const char *b0, *b1, *b2;      // NV = 2: r, v;  NV = 3: v, p, r
    double c0, c1;                 // NV = 2: -alpha;  NV = 3: -beta w, beta
    __device__ __forceinline__ double one(double a, double b, double c) const { return NV == 2 ? comb_k3(a, b, c0) : comb_k1(a, b, c, c0, c1); }
    __device__ __forceinline__ D2 ld2(int64_t soff, uint32_t voff) const {
        const D2 a = *reinterpret_cast<const D2 *>(b0 + soff + voff), b = *reinterpret_cast<const D2 *>(b1 + soff + voff);
        D2 c{0.0, 0.0};
        if (NV == 3) c = *reinterpret_cast<const D2 *>(b2 + soff + voff);
        return D2{one(a.lo, b.lo, c.lo), one(a.hi, b.hi, c.hi)};
    }
    __device__ __forceinline__ double ld1(uint32_t voff) const {
        const double a = *reinterpret_cast<const double *>(b0 + voff), b = *reinterpret_cast<const double *>(b1 + voff);
        const double c = NV == 3 ? *reinterpret_cast<const double *>(b2 + voff) : 0.0;
        return one(a, b, c);
    }
    __device__ __forceinline__ double lds(int64_t soff) const {
        const double a = *reinterpret_cast<const double *>(b0 + soff), b = *reinterpret_cast<const double *>(b1 + soff);
        const double c = NV == 3 ? *reinterpret_cast<const double *>(b2 + soff) : 0.0;
        return one(a, b, c);
    }
};

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
template <int UL, int SC, class XS, class AfterLoads>
__device__ __forceinline__ void full_uniform_block(const PairEnt<double> *s_pair, uint64_t pat, int ulen, bool seam, int seam1, int seam2, const XS &xs,
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
        const D2 px = xs.ld2((int64_t)off8, r8);
        pl[t] = px.lo; ph[t] = px.hi;
        if (SC > 0 && t == SC) off8c = off8;
    }
    // the two ends of the block, x[ra - 1 + o] and x[ra + 128 + o]: wave-uniform addresses, read through the SCALAR
    // cache after the last LDS read of the block (scalar loads return out of order and share the LDS counter) — a
    // 64-lane load of them would cost the vector-memory pipe as much as a gather
    T e_lo = 0.0, e_hi = 0.0;
    if (SC > 0) {
        const int64_t xe = (int64_t)off8c + ra8;
        e_lo = xs.lds(xe - 8); e_hi = xs.lds(xe + 2 * WAVE * 8);
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
template <class XS, class AfterLoads>
__device__ __forceinline__ void full_uniform_dispatch(const PairEnt<double> *s_pair, uint64_t pat, int ulen, int sc, bool seam, int seam1, int seam2, const XS &xs,
                                                      uint32_t r8, uint32_t ra8, int lane, AfterLoads &&after_loads,
                                                      double &acc0, double &acc1) {
    // (scalar branches) the stencils: 7-point 3-D, 5-point 2-D, 3-point 1-D with sorted columns; anything else generic
    if (ulen == 7 && sc == 3) full_uniform_block<7, 3>(s_pair, pat, ulen, seam, seam1, seam2, xs, r8, ra8, lane, after_loads, acc0, acc1);
    else if (ulen == 5 && sc == 2) full_uniform_block<5, 2>(s_pair, pat, ulen, seam, seam1, seam2, xs, r8, ra8, lane, after_loads, acc0, acc1);
    else if (ulen == 3 && sc == 1) full_uniform_block<3, 1>(s_pair, pat, ulen, seam, seam1, seam2, xs, r8, ra8, lane, after_loads, acc0, acc1);
    else full_uniform_block<0, 0>(s_pair, pat, ulen, seam, seam1, seam2, xs, r8, ra8, lane, after_loads, acc0, acc1);
}


struct Blk2Loads {
    int ra, rb, pa, nn;      // descriptor of the 128-row block
    bool uni; int ulen;      // uniform block (every row = the first row's ulen codes)
    int tri;                 // ... and the slot of its column triple's centre (0: none)
    bool is_seam; int seam1, seam2;   // ... or uniform but for one or two rows (mark_uniform_kernel's encoding: nn >> 16, rb's low bits)
    int a, b;                // row_ptr[i0], row_ptr[i0 + 1], i0 = min(ra + 2 lane, rb - 1)
    double u0, u1;           // dot operands of the lane's two rows
    double o0, o1;           // XFused: the formed vector at the lane's two rows (stored by the walk)
    u4v wc;                  // 16 code bytes
    int di;                  // ... and the b128 slot they go to
};

// The walk of the two-rows-per-lane kernels over `n_wide` 128-row blocks (positions of `order`, or natural order): a
// wavefront takes every (gridDim.x * NWAVE)-th position, the next block's loads are issued before this block's products.
// Shared by spmv_pair2_kernel (the whole matrix) and spmv_tile_kernel (the blocks outside its tiles).  d0 / d1: the
// lane's running dot partials (DOT as in launch_spmv).  s_pair: the staged (byte offset, value) table; s_c: NWAVE
// zero-initialised code slices.
// YNT: y is written with non-temporal stores (HBM-sized vectors: the result is not read again before it has been evicted)
// XS: where x comes from (XPlain: memory; XFused: formed on the fly).  With XFused the walk also STORES the formed vector for the
// rows it owns (`own`, unless null), and where `u` is null the dot operand is that vector (K4's t.s).
template <int DOT, bool YNT, class XS = XPlain>
__device__ __forceinline__ void pair2_walk(int n_wide, int xcd_chunk, const BlkDesc *__restrict__ desc,
                                           const int32_t *__restrict__ order, const int32_t *__restrict__ row_ptr,
                                           const uint8_t *__restrict__ code, const XS xs,
                                           double *__restrict__ y, const double *__restrict__ u, int nrows, int ncols,
                                           const PairEnt<double> *s_pair, uint32_t (*s_c)[CW2], double &d0, double &d1,
                                           double *__restrict__ own = nullptr) {
    using T = double;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint8_t *cb = reinterpret_cast<const uint8_t *>(s_c[wv]);
    const uint32_t xlast_pair = (uint32_t)(ncols - 2) * 8u;                    // last byte offset a 16-byte x load may start at
    const bool u_is_x = XS::FUSED && u == nullptr;                             // the dot operand is the vector formed on the fly

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
        if (DOT != 0 && !u_is_x) {
            const int p0 = min(r0, nrows - 2);                                  // the pair (u[p0], u[p0 + 1]) is inside u
            const D2 uu = *reinterpret_cast<const D2 *>(reinterpret_cast<const char *>(u) + (uint32_t)p0 * 8u);
            L.u0 = r0 == p0 ? uu.lo : uu.hi;                                    // r0 == nrows - 1: its operand is the pair's second half
            L.u1 = uu.hi;
        }
        if constexpr (XS::FUSED) {                                              // the formed vector at the lane's own rows
            const int p0 = min(r0, nrows - 2);
            const D2 oo = xs.ld2(0, (uint32_t)p0 * 8u);
            L.o0 = r0 == p0 ? oo.lo : oo.hi;
            L.o1 = oo.hi;
            if (DOT != 0 && u_is_x) { L.u0 = L.o0; L.u1 = L.o1; }
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
    [[maybe_unused]] T c_o0 = 0.0, c_o1 = 0.0;
    auto stage = [&](const Blk2Loads &L) {
        const int shift = L.pa & 3;
        c_uni = L.uni;
        if constexpr (XS::FUSED) { c_o0 = L.o0; c_o1 = L.o1; }
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
                full_uniform_dispatch(s_pair, c_pat, ulen, c_tri, c_seam, c_seam1, c_seam2, xs, r8, (uint32_t)c_ra * 8u, lane, after_gathers, acc0, acc1);
            } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t >= ulen) break;
                const PairEnt<T> e = s_pair[(int)((c_pat >> (8 * t)) & 255u)];
                av[t] = e.val;
                const uint32_t vo0 = len0 > 0 ? r8 + (uint32_t)e.off8 : 0u;
                const uint32_t vp = min(vo0, xlast_pair);                       // only a single-row lane at the matrix end is ever clamped
                hi_bits |= (vo0 != vp ? 1u : 0u) << t;
                const D2 px = xs.ld2(0, vp);
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
            T xg1[8];                 // row 1's own gather where its column is not row 0's + 1
            uint32_t same_bits = 0, hi_bits = 0;      // per slot: row 1 shares the gather / row 0's x is the pair's second half
#pragma unroll
            for (int t = 0; t < 8; ++t) { pl[t] = 0.0; ph[t] = 0.0; xg1[t] = 0.0; }
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
                const D2 px = xs.ld2(0, vp);
                pl[t] = px.lo; ph[t] = px.hi;
                const bool need1 = v1 && !same;
                if (__builtin_amdgcn_ballot_w64(need1) != 0)                    // scalar branch: interior stencil blocks skip it
                    xg1[t] = xs.ld1(need1 ? r8 + 8u + (uint32_t)off1 : 0u);
            }
            if (j0 == 0) after_gathers();
            __builtin_amdgcn_sched_barrier(0);                                  // every gather out before the first product
            asm volatile("" ::: "memory");      // the values are looked up again below rather than held in 32 registers across the wait
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (j0 + t < len0) acc0 = acc0 + (((hi_bits >> t) & 1u) ? ph[t] : pl[t]) * s_pair[cp0[t]].val;
                if (j0 + t < len1) acc1 = acc1 + (((same_bits >> t) & 1u) ? ph[t] : xg1[t]) * s_pair[cp1[t]].val;
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
            if constexpr (XS::FUSED) { if (own != nullptr) *reinterpret_cast<D2 *>(reinterpret_cast<char *>(own) + r8) = D2{c_o0, c_o1}; }
        } else if (r0 < c_rb) {
            if constexpr (XS::FUSED) { if (own != nullptr) *reinterpret_cast<T *>(reinterpret_cast<char *>(own) + r8) = c_o0; }
            *reinterpret_cast<T *>(reinterpret_cast<char *>(y) + r8) = acc0;
            if (DOT == 1) d0 = d0 + c_u0 * acc0;
            if (DOT == 2) { d0 = d0 + acc0 * acc0; d1 = d1 + acc0 * c_u0; }
        }
        wave_lds_fence();
        if (more) stage(nxt);
        dn = uniform(dn2); o2 = __builtin_amdgcn_readfirstlane(o3);
    }
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

// (UL, FL, FH) shapes spmv_tile_kernel is built for: 7-point 3-D, 5-point 2-D with a far or a near line band, 3-point 1-D, and
// bands with two far diagonals
#define SPRS_TILE_SHAPES(X) X(7, 1, 1) X(5, 1, 1) X(5, 0, 0) X(3, 0, 0) X(3, 1, 1) X(7, 0, 0)
// ... and with the wide window (pair-code stream only; grids whose lines are 511 to 1534 long)
#define SPRS_TILE_SHAPES_WIDE(X) X(7, 1, 1) X(5, 1, 1) X(5, 0, 0) X(7, 0, 0)
// plane-streaming chains (spmv_chain.hip): 128-row blocks per chain tile, and the (UL, TRI) shapes the kernel is built for
constexpr int CH_B = 16;
#define SPRS_CHAIN_SHAPES(X) X(7, true) X(7, false) X(5, true) X(5, false) X(3, true)
}  // namespace

// ---- spmv_chain.hip: the chains of plan CP + the per-block walk over the blocks outside them, one launch
int launch_chain_pair(const sprs_csr *A, const sprs_chain_plan &CP, int g, const double *x, double *y, int dot_mode, const double *u,
                      double *part0, double *part1, const int *status, const Fin &fin);
int chain_rows();
// ---- spmv_tile.hip / spmv_tile_off.hip: one launch = the tiles of plan TP + the per-block walk over the blocks outside them
int launch_tile_pair(const sprs_csr *A, const sprs_tile_plan &TP, int g, const double *x, double *y, int dot_mode, const double *u,
                     double *part0, double *part1, const int *status, const Fin &fin);
int launch_tile_off(const sprs_csr *A, const sprs_tile_plan &TP, int g, const BlkDesc *desc64, const double *x, double *y, int dot_mode,
                    const double *u, double *part0, double *part1, const int *status, const Fin &fin, const V2d *tail2, int g2_last);

}  // namespace sprs
