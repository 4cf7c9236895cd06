// LDS x-window tiles of the f64 PAIR-CODE stream (knob "spmv_tile"; design notes in spmv_dict_dev.hpp, profiles/r03_tuning.md §8).
#include "spmv_dict_dev.hpp"

namespace sprs {
namespace {

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
    stage_pair(s_pair, tid, off_tab[tid] * 8, val_tab[tid]);                  // BLOCK == TAB (the seam rows' own values; the walk below)
    for (int i = lane; i < CW2; i += WAVE) s_c[wv][i] = 0;
    __syncthreads();
    if (run_state != ST_RUNNING) { fin_idle(fin, DOT == 2); return; }
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
        pair2_walk<DOT, true>(n_left, 0, desc, left_order, row_ptr, code, XPlain{reinterpret_cast<const char *>(x)}, y, u, nrows, ncols, s_pair, s_c, d0, d1);
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

int launch_tile_pair(const sprs_csr *A, const sprs_tile_plan &TP, int g, const double *x, double *y, int dot_mode, const double *u,
                     double *part0, double *part1, const int *status, const Fin &fin) {
    sprs_ctx *c = A->ctx;
    const sprs_dict *D = A->dict;
    const BlkDesc *wd = reinterpret_cast<const BlkDesc *>(D->wide_desc);
    const double *pvd = reinterpret_cast<const double *>(D->pair_val);
    TilePat tp;
    for (int t = 0; t < 8; ++t) { tp.off[t] = TP.off[t]; tp.val[t] = TP.val[t]; }
    const bool ux = dot_mode != 0 && u == x;      // the dot operand is the input vector (mul_vec_dot, MINRES' v.Av, K4 without a preconditioner)
#define SPRS_TSPMV(DM, UXV, U, L, H) SPRS_LAUNCH_SPMV(c, (spmv_tile_kernel<DM, UXV, U, L, H, SPRS_TW>), g, reinterpret_cast<const int2 *>(TP.list), TP.xstart, wd, tp, TP.n_left, \
                                                      TP.left, A->row_ptr, D->pair_code, D->pair_off, pvd, x, y, u, part0, part1, status, (int)A->nrows, (int)A->ncols, fin)
#define SPRS_TSHAPE(U, L, H)                                                                                         \
    if (TP.ul == U && TP.fl == L && TP.fh == H) {                                                                    \
        if (dot_mode == 0) SPRS_TSPMV(0, false, U, L, H);                                                            \
        else if (dot_mode == 1) { if (ux) SPRS_TSPMV(1, true, U, L, H); else SPRS_TSPMV(1, false, U, L, H); }        \
        else if (ux) SPRS_TSPMV(2, true, U, L, H); else SPRS_TSPMV(2, false, U, L, H);                               \
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

}  // namespace sprs
