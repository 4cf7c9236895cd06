// LDS x-window tiles of the f64 OFFSET-CODE stream (a value per entry; design notes in spmv_dict_dev.hpp, profiles/r03_tuning.md §9).
#include "spmv_dict_dev.hpp"

namespace sprs {
namespace {

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
    if (run_state != ST_RUNNING) { fin_idle(fin, DOT == 2); return; }
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

}  // namespace

int launch_tile_off(const sprs_csr *A, const sprs_tile_plan &TP, int g, const BlkDesc *desc64, const double *x, double *y, int dot_mode,
                    const double *u, double *part0, double *part1, const int *status, const Fin &fin, const V2d *tail2, int g2_last) {
    sprs_ctx *c = A->ctx;
    const sprs_dict *D = A->dict;
    TilePat tp;
    for (int t = 0; t < 8; ++t) { tp.off[t] = TP.off[t]; tp.val[t] = 0.0; }
    const bool ux = dot_mode != 0 && u == x;      // the dot operand is the input vector (mul_vec_dot, MINRES' v.Av, K4 without a preconditioner)
    const BlkDesc *owd = reinterpret_cast<const BlkDesc *>(D->owide_desc);
    const double *v = reinterpret_cast<const double *>(A->val);
#define SPRS_TOSPMV(DM, UXV, U, L, H) SPRS_LAUNCH_SPMV(c, (spmv_tile_off_kernel<DM, UXV, U, L, H>), g, reinterpret_cast<const int2 *>(TP.list), TP.xstart, owd, tp, \
                                                       TP.n_left, TP.left, desc64, A->row_ptr, D->idx_code, D->off_tab, v, x, y, u, part0, part1, status, fin, tail2, g2_last)
#define SPRS_TOSHAPE(U, L, H)                                                                                        \
    if (TP.ul == U && TP.fl == L && TP.fh == H) {                                                                    \
        if (dot_mode == 0) SPRS_TOSPMV(0, false, U, L, H);                                                           \
        else if (dot_mode == 1) { if (ux) SPRS_TOSPMV(1, true, U, L, H); else SPRS_TOSPMV(1, false, U, L, H); }      \
        else if (ux) SPRS_TOSPMV(2, true, U, L, H); else SPRS_TOSPMV(2, false, U, L, H);                             \
    }
    SPRS_TILE_SHAPES(SPRS_TOSHAPE)
#undef SPRS_TOSHAPE
#undef SPRS_TOSPMV
    SPRS_HIP_TRY(c, hipGetLastError());
    return SPRS_OK;
}

}  // namespace sprs
