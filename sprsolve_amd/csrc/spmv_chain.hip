// Plane-streaming chains of the f64 PAIR-CODE stream (knob "spmv_chain"; profiles/r04_tuning.md §2, scripts/micro/fused_window.hip).
//
// The LDS x-window tiles (spmv_tile.hip) stage the NEAR columns of 4096 rows once and load the FAR windows — the +-plane
// neighbours of a 3-D stencil — row pair by row pair: (T + 2W) / T + 2 sixteen-byte loads per lane and 128 rows, and the far
// loads are served by an XCD's L2 only while what that XCD's workgroups hold in flight fits it (scripts/micro/fused_window.hip:
// the same kernel with two input vectors falls from 5.6 to 3 TB/s when it does not).  Here a workgroup owns a COLUMN of tiles —
// CH_ROWS rows at ts, ts + Pf, ts + 2 Pf, ... (Pf = the far offset; each tile within 127 rows of that on the 128-row block grid) —
// and keeps the windows of three consecutive tiles in LDS: the -Pf / +Pf operands of a tile are the centres of its neighbours'
// windows.  No far load exists; every element of x crosses the vector L1 (T + 2W) / T = 1.5 times instead of 3.25, independent of
// what survives in an L2; the loads of the window after next fly over the fold of the current tile.  A tile = CH_B consecutive
// FULL uniform 128-row blocks (plain or seam, mark_uniform_kernel) of the matrix's most frequent pattern, which must be
// far, near.., far with offsets -Pf .. +Pf, Pf even (build_chain_plan, spmv_dict.hip).  The fold per row is full_uniform_block's /
// spmv_tile_kernel's — the same products in the same left-to-right order, seam rows under scalar tests — with every operand
// taken from LDS: y bit-identical.  The 128-row blocks outside the chains (boundary planes, the blocks around boundary lines,
// remainders of runs) are walked by the same launch afterwards (pair2_walk), so a launch writes all of y and one dot partial per
// workgroup.
#include "spmv_dict_dev.hpp"
#include "bicg_fuse.hpp"

namespace sprs {
namespace {

constexpr int CH_ROWS = CH_B * 2 * WAVE;          // 2048 rows per tile
constexpr int CH_W = TILE_W;                      // half-width of a window: near = |col - row| <= CH_W - 2
constexpr int CH_WL = CH_ROWS + 2 * CH_W;         // 3072 doubles per window, three of them: 72 KiB — two workgroups per CU

// UX: the dot operand is the input vector itself (taken from the window).  TRI: the near slots are (.., c - 1, c, c + 1, ..)
// around an even centre offset and all other near offsets are even (every stencil with sorted columns on a grid of even line
// length): 16-byte LDS reads, the centre's pair serves the +-1 columns' inner halves.
// FUSE (krylov.hip, "fused SpMV input"): 0 = x is a vector in memory.  2 = x is BiCGStab's s = r + v * (-alpha) (K3 into K4:
// x = r, in1 = v), 3 = its p' = (v * (-beta w) + p * beta) + r (K1 into K2: x = v, in1 = p, in2 = r): `pro` is that update's kernel
// structure — its prologue (the scalars of the recurrence and its convergence / restart / breakdown decisions, taken identically by
// every workgroup from the same partials) runs at the top of this launch, the update itself while a window is staged, with the
// update's own rounding sequence; the updated vector is written for the launch's own rows to `own` and never read back.
struct NoPro { __device__ __forceinline__ bool prologue() { return true; } };
template <int FUSE, class PRO, int DOT, bool UX, int UL, bool TRI>
__global__ __launch_bounds__(BLOCK) void spmv_chain_kernel(const int4 *__restrict__ tiles, const int2 *__restrict__ segs, const int32_t *__restrict__ xstart,
                                                           const BlkDesc *__restrict__ desc, const TilePat pat,
                                                           int n_left, const int32_t *__restrict__ left_order,
                                                           const int32_t *__restrict__ row_ptr, const uint8_t *__restrict__ code,
                                                           const int32_t *__restrict__ off_tab, const double *__restrict__ val_tab,
                                                           const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ u,
                                                           double *__restrict__ part0, double *__restrict__ part1,
                                                           const int *__restrict__ status, int nrows, int ncols, const Fin fin,
                                                           PRO pro, const double *__restrict__ in1, const double *__restrict__ in2,
                                                           double *__restrict__ own) {
    using T = double;
    constexpr int NV = FUSE == 0 ? 1 : FUSE;            // input vectors
    static_assert(FUSE == 0 || (FUSE == 2 && DOT == 2 && UX) || (FUSE == 3 && DOT == 1 && !UX), "fused flavours: K3 into K4, K1 into K2");
    constexpr int W = CH_W, WL = CH_WL;
    constexpr int NW = WL / 2 / BLOCK;                  // 16-byte window pieces per lane (6)
    constexpr int NQ = CH_B / NWAVE;                    // 128-row blocks per wavefront and tile (4)
    constexpr int NN = UL - 2;                          // near slots
    constexpr int TC = NN / 2;                          // TRI: the centre's position among the near slots
    static_assert(WL % (2 * BLOCK) == 0 && CH_B % NWAVE == 0 && NN >= 1 && UL <= 8 && (!TRI || (NN & 1)), "chain shape");
    static_assert(sizeof(T) * WL >= sizeof(uint32_t) * NWAVE * CW2, "the left walk's code slices live in the first window");
    __shared__ __attribute__((aligned(16))) T win[3][WL];
    __shared__ PairEnt<T> s_pair[TAB];
    __shared__ T red[NWAVE];
    const int run_state = status != nullptr ? *status : (int)ST_RUNNING;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    stage_pair(s_pair, tid, off_tab[tid] * 8, val_tab[tid]);                  // BLOCK == TAB (the seam rows' own values; the walk below)
    __syncthreads();
    if (run_state != ST_RUNNING) { fin_idle(fin, DOT == 2); return; }
    [[maybe_unused]] T c0 = 0.0, c1 = 0.0;              // coefficients of the fused update
    if constexpr (FUSE != 0) {
        if (!pro.prologue()) return;                    // converged / restart requested / breakdown: the same decision in every workgroup
        if constexpr (FUSE == 2) c0 = pro.na; else { c0 = pro.a; c1 = pro.beta; }
    }
    T d0 = 0.0, d1 = 0.0;

    const int Pf = pat.off[UL - 1];                     // == -pat.off[0]
    const int xhi = (ncols - 2) & ~1;                   // last (even) index a 16-byte load of x may start at
    const int xcd = blockIdx.x & 7;
    const int sstep = gridDim.x >> 3;
    const int send = xstart[xcd + 1];
    // PF windows are in flight while a tile is folded: the register sets R[0 .. PF).  Two for the plain kernel and K3-in-K4; K1-in-K2
    // holds three vectors per window and keeps one (a second set would spill).
    constexpr int PF = FUSE == 3 ? 1 : 2;
    struct WinRegs { u4v a[NW]; u4v b[NV >= 2 ? NW : 1]; u4v c[NV == 3 ? NW : 1]; };
    WinRegs R0;
    [[maybe_unused]] WinRegs R1;
    // raw window of the tile that starts at row tw: [tw - W, tw + CH_ROWS + W).  The margins of a window at either end of x are
    // clamped piece by piece — what they hold is never folded (every column a tile's rows have is inside x)
    auto issue = [&](int tw, WinRegs &R) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            int g = tw - W + 2 * (tid + i * BLOCK);
            g = g < 0 ? 0 : (g > xhi ? xhi : g);
            R.a[i] = *reinterpret_cast<const u4v *>(x + g);              // (non-temporal window loads, allocating result stores: no difference, profiles/r04_tuning.md §3)
            if constexpr (NV >= 2) R.b[i] = *reinterpret_cast<const u4v *>(in1 + g);
            if constexpr (NV == 3) R.c[i] = *reinterpret_cast<const u4v *>(in2 + g);
        }
    };
    auto stage = [&](int slot, const WinRegs &R) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if constexpr (FUSE == 0) {
                *reinterpret_cast<u4v *>(&win[slot][2 * (tid + i * BLOCK)]) = R.a[i];
            } else {
                D2 a, b, cc{0.0, 0.0};
                __builtin_memcpy(&a, &R.a[i], 16); __builtin_memcpy(&b, &R.b[i], 16);
                if constexpr (NV == 3) __builtin_memcpy(&cc, &R.c[i], 16);
                const D2 o = FUSE == 2 ? D2{comb_k3(a.lo, b.lo, c0), comb_k3(a.hi, b.hi, c0)}
                                       : D2{comb_k1(a.lo, b.lo, cc.lo, c0, c1), comb_k1(a.hi, b.hi, cc.hi, c0, c1)};
                *reinterpret_cast<D2 *>(&win[slot][2 * (tid + i * BLOCK)]) = o;
            }
        }
    };
    for (int s = xstart[xcd] + (blockIdx.x >> 3); s < send; s += sstep) {
        const int2 sg = segs[s];
        const int t0 = __builtin_amdgcn_readfirstlane(sg.x), L = __builtin_amdgcn_readfirstlane(sg.y);
        int4 ent = tiles[t0];
        uint32_t rbw[NQ]; int nnw[NQ];                  // seam words of the tile's blocks (wave-uniform: scalar loads), a tile ahead
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const BlkDesc d = desc[__builtin_amdgcn_readfirstlane(ent.x) + q * NWAVE + wv]; rbw[q] = (uint32_t)d.rb; nnw[q] = d.nn; }
        __syncthreads();                                // the previous segment's windows have been read
        int ts_prev = __builtin_amdgcn_readfirstlane(ent.y) - Pf;      // the segment's first tile takes its -Pf operands from the window at ts - Pf
        int sp = 0, sc = 1, sn = 2;                     // slots of the previous / current / next window
        issue(ts_prev, R0); stage(sp, R0);
        issue(__builtin_amdgcn_readfirstlane(ent.y), R0); stage(sc, R0);
        issue(__builtin_amdgcn_readfirstlane(ent.z), R0);                                        // tile 0's next window
        if constexpr (PF == 2) { if (L > 1) issue(tiles[t0 + 1].z, R1); }                       // tile 1's next window
        // one tile: stage its next window from RS (issued PF tiles ago), then refill RS with the next window of tile k + PF
        auto body = [&](int k, WinRegs &RS) {
            const int ts = __builtin_amdgcn_readfirstlane(ent.y), ts_next = __builtin_amdgcn_readfirstlane(ent.z);
            uint32_t rbc[NQ]; int nnc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) { rbc[q] = rbw[q]; nnc[q] = nnw[q]; }
            stage(sn, RS);                              // the next tile's window (its loads flew over the previous folds)
            __syncthreads();
            if (k + PF < L) issue(tiles[t0 + k + PF].z, RS);
            if (k + 1 < L) {
                ent = tiles[t0 + k + 1];
                const int b1 = __builtin_amdgcn_readfirstlane(ent.x);
#pragma unroll
                for (int q = 0; q < NQ; ++q) { const BlkDesc d = desc[b1 + q * NWAVE + wv]; rbw[q] = (uint32_t)d.rb; nnw[q] = d.nn; }
            }
            // x[r - Pf] sits in the previous window at (r - ts) + W + dlo, x[r + Pf] in the next one at (r - ts) + W + dhi
            const int dlo = ts - Pf - ts_prev, dhi = ts + Pf - ts_next;
            const T *wp = &win[sp][W + dlo], *wc = &win[sc][W], *wn = &win[sn][W + dhi];
            D2 uu[NQ];
            if (DOT != 0 && !UX) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const u4v w4 = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(u + (ts + ((q * NWAVE + wv) << 7) + 2 * lane)));
                    __builtin_memcpy(&uu[q], &w4, 16);
                }
            }
            // operands of block q + 1 are read before block q is folded (the seam branch keeps the compiler from doing it)
            T opl[UL], oph[UL], ux0 = 0.0, ux1 = 0.0;
            auto read_ops = [&](int q) {
                const int l = ((q * NWAVE + wv) << 7) + 2 * lane;
                const D2 f0 = *reinterpret_cast<const D2 *>(wp + l), f1 = *reinterpret_cast<const D2 *>(wn + l);
                opl[0] = f0.lo; oph[0] = f0.hi; opl[UL - 1] = f1.lo; oph[UL - 1] = f1.hi;
                if constexpr (TRI) {
#pragma unroll
                    for (int t = 0; t < NN; ++t) {
                        if (t == TC - 1 || t == TC + 1) continue;
                        const D2 v2 = *reinterpret_cast<const D2 *>(wc + l + pat.off[1 + t]);
                        opl[1 + t] = v2.lo; oph[1 + t] = v2.hi;
                    }
                    if constexpr (NN >= 3) {
                        const int oc = pat.off[1 + TC];
                        opl[TC] = wc[l + oc - 1]; oph[TC] = opl[1 + TC];                 // column c - 1: (x[r0 - 1 + c], x[r0 + c])
                        opl[2 + TC] = oph[1 + TC]; oph[2 + TC] = wc[l + oc + 2];         // column c + 1: (x[r0 + 1 + c], x[r0 + 2 + c])
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < NN; ++t) { opl[1 + t] = wc[l + pat.off[1 + t]]; oph[1 + t] = wc[l + pat.off[1 + t] + 1]; }
                }
                if ((DOT != 0 && UX) || FUSE != 0) { const D2 c2 = *reinterpret_cast<const D2 *>(wc + l); ux0 = c2.lo; ux1 = c2.hi; }
            };
            read_ops(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const uint32_t rbq = (uint32_t)__builtin_amdgcn_readfirstlane((int)rbc[q]);
                const bool seam = (rbq & SEAM2) != 0;
                T pl[UL], ph[UL];
#pragma unroll
                for (int t = 0; t < UL; ++t) { pl[t] = opl[t]; ph[t] = oph[t]; }
                const T u0 = (DOT != 0 && !UX) ? uu[q].lo : ux0, u1 = (DOT != 0 && !UX) ? uu[q].hi : ux1;
                [[maybe_unused]] const D2 mine{ux0, ux1};              // FUSE: the formed vector at the lane's own two rows
                if (q + 1 < NQ) read_ops(q + 1);
                T acc0 = 0.0, acc1 = 0.0;
                if (!seam) {
#pragma unroll
                    for (int t = 0; t < UL; ++t) {
                        acc0 = acc0 + pl[t] * pat.val[t];
                        acc1 = acc1 + ph[t] * pat.val[t];
                    }
                } else {
                    // (spmv_tile_kernel's seam fold) local rows k and k + 1 fold only the slots of their masks, with their own
                    // value where they carry one
                    const int seam1 = __builtin_amdgcn_readfirstlane(nnc[q]) >> 16, seam2 = (int)(rbq & 0x3ffffffu);
                    const int kk = seam1 & 127, maskA = (seam1 >> 7) & 255, maskB = seam2 & 255;
                    const bool a0 = 2 * lane == kk, a1 = 2 * lane + 1 == kk, b0s = 2 * lane == kk + 1, b1s = 2 * lane + 1 == kk + 1;
                    if ((seam2 & 0x3000000) == 0) {
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
                if constexpr (FUSE == 3) {          // (K3 in K4 stores nothing: K5 forms s again, krylov.hip BicgK5<SV>)
                    u4v ov;
                    __builtin_memcpy(&ov, &mine, 16);
                    __builtin_nontemporal_store(ov, reinterpret_cast<u4v *>(own + (ts + ((q * NWAVE + wv) << 7) + 2 * lane)));
                }
                if (DOT == 1) { d0 = d0 + u0 * acc0; d0 = d0 + u1 * acc1; }
                if (DOT == 2) { d0 = d0 + acc0 * acc0; d1 = d1 + acc0 * u0; d0 = d0 + acc1 * acc1; d1 = d1 + acc1 * u1; }
            }
            __syncthreads();                            // the previous window's slot is free for the window after next
            ts_prev = ts;
            const int o = sp; sp = sc; sc = sn; sn = o;
        };
        if constexpr (PF == 2) {
            for (int k = 0; k < L; k += 2) { body(k, R0); if (k + 1 < L) body(k + 1, R1); }
        } else {
            for (int k = 0; k < L; ++k) body(k, R0);
        }
    }
    if (n_left > 0) {
        __syncthreads();
        uint32_t (*s_c)[CW2] = reinterpret_cast<uint32_t (*)[CW2]>(&win[0][0]);
        for (int i = lane; i < CW2; i += WAVE) s_c[wv][i] = 0;                 // (wavefront-private slices: no barrier)
        if constexpr (FUSE == 0)
            pair2_walk<DOT, true>(n_left, 0, desc, left_order, row_ptr, code, XPlain{reinterpret_cast<const char *>(x)}, y, u, nrows, ncols, s_pair, s_c, d0, d1);
        else
            pair2_walk<DOT, true, XFused<NV>>(n_left, 0, desc, left_order, row_ptr, code,
                                              XFused<NV>{reinterpret_cast<const char *>(x), reinterpret_cast<const char *>(in1), reinterpret_cast<const char *>(in2), c0, c1},
                                              y, UX ? nullptr : u, nrows, ncols, s_pair, s_c, d0, d1, own);
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

}  // namespace

int chain_rows() { return CH_ROWS; }

int launch_chain_pair(const sprs_csr *A, const sprs_chain_plan &CP, int g, const double *x, double *y, int dot_mode, const double *u,
                      double *part0, double *part1, const int *status, const Fin &fin) {
    sprs_ctx *c = A->ctx;
    const sprs_dict *D = A->dict;
    const BlkDesc *wd = reinterpret_cast<const BlkDesc *>(D->wide_desc);
    const double *pvd = reinterpret_cast<const double *>(D->pair_val);
    TilePat tp;
    for (int t = 0; t < 8; ++t) { tp.off[t] = CP.off[t]; tp.val[t] = CP.val[t]; }
    const bool ux = dot_mode != 0 && u == x;
#define SPRS_CSPMV(DM, UXV, U, TR) SPRS_LAUNCH_SPMV(c, (spmv_chain_kernel<0, NoPro, DM, UXV, U, TR>), g, reinterpret_cast<const int4 *>(CP.tiles), reinterpret_cast<const int2 *>(CP.segs), \
                                                    CP.xstart, wd, tp, CP.n_left, CP.left, A->row_ptr, D->pair_code, D->pair_off, pvd, x, y, u, part0, part1, status,   \
                                                    (int)A->nrows, (int)A->ncols, fin, NoPro{}, nullptr, nullptr, nullptr)
#define SPRS_CSHAPE(U, TR)                                                                                           \
    if (CP.ul == U && (CP.tri != 0) == TR) {                                                                         \
        if (dot_mode == 0) SPRS_CSPMV(0, false, U, TR);                                                              \
        else if (dot_mode == 1) { if (ux) SPRS_CSPMV(1, true, U, TR); else SPRS_CSPMV(1, false, U, TR); }            \
        else if (ux) SPRS_CSPMV(2, true, U, TR); else SPRS_CSPMV(2, false, U, TR);                                   \
    }
    SPRS_CHAIN_SHAPES(SPRS_CSHAPE)
#undef SPRS_CSHAPE
#undef SPRS_CSPMV
    SPRS_HIP_TRY(c, hipGetLastError());
    return SPRS_OK;
}


int launch_chain_k4f(const sprs_csr *A, int g, const BicgK3<double, double, false> &pro, const double *r, const double *v, double *s_out,
                     double *t, double *partTT, double *partTR, const int *status) {
    sprs_ctx *c = A->ctx;
    const sprs_dict *D = A->dict;
    const sprs_chain_plan &CP = D->chain_pair;
    const BlkDesc *wd = reinterpret_cast<const BlkDesc *>(D->wide_desc);
    const double *pvd = reinterpret_cast<const double *>(D->pair_val);
    TilePat tp;
    for (int k = 0; k < 8; ++k) { tp.off[k] = CP.off[k]; tp.val[k] = CP.val[k]; }
    typedef BicgK3<double, double, false> K3;
#define SPRS_C4F(U, TR)                                                                                                 \
    if (CP.ul == U && (CP.tri != 0) == TR)                                                                              \
        SPRS_LAUNCH_SPMV(c, (spmv_chain_kernel<2, K3, 2, true, U, TR>), g, reinterpret_cast<const int4 *>(CP.tiles), reinterpret_cast<const int2 *>(CP.segs), \
                         CP.xstart, wd, tp, CP.n_left, CP.left, A->row_ptr, D->pair_code, D->pair_off, pvd, r, t, (const double *)nullptr, partTT, partTR, status, \
                         (int)A->nrows, (int)A->ncols, Fin{}, pro, v, (const double *)nullptr, s_out);
    SPRS_CHAIN_SHAPES(SPRS_C4F)
#undef SPRS_C4F
    SPRS_HIP_TRY(c, hipGetLastError());
    return SPRS_OK;
}

int launch_chain_k2f(const sprs_csr *A, int g, const BicgK1<double, double, false> &pro, const double *v_old, const double *p, const double *r,
                     double *p_out, double *v_out, const double *r0, double *partB, const int *status) {
    sprs_ctx *c = A->ctx;
    const sprs_dict *D = A->dict;
    const sprs_chain_plan &CP = D->chain_pair;
    const BlkDesc *wd = reinterpret_cast<const BlkDesc *>(D->wide_desc);
    const double *pvd = reinterpret_cast<const double *>(D->pair_val);
    TilePat tp;
    for (int k = 0; k < 8; ++k) { tp.off[k] = CP.off[k]; tp.val[k] = CP.val[k]; }
    typedef BicgK1<double, double, false> K1;
#define SPRS_C2F(U, TR)                                                                                                 \
    if (CP.ul == U && (CP.tri != 0) == TR)                                                                              \
        SPRS_LAUNCH_SPMV(c, (spmv_chain_kernel<3, K1, 1, false, U, TR>), g, reinterpret_cast<const int4 *>(CP.tiles), reinterpret_cast<const int2 *>(CP.segs), \
                         CP.xstart, wd, tp, CP.n_left, CP.left, A->row_ptr, D->pair_code, D->pair_off, pvd, v_old, v_out, r0, partB, (double *)nullptr, status, \
                         (int)A->nrows, (int)A->ncols, Fin{}, pro, p, r, p_out);
    SPRS_CHAIN_SHAPES(SPRS_C2F)
#undef SPRS_C2F
    SPRS_HIP_TRY(c, hipGetLastError());
    return SPRS_OK;
}

}  // namespace sprs
