// CSR SpMV for gfx950 — the dominant kernel of the path (reference src/mat.rs:68-129).
//
// Design (MI355X-first, not a translation of the rayon loop):
//  * The HBM-resident stream is (col_idx, val): 12 B/nnz for f64.  It is read with fully
//    coalesced loads — consecutive lanes take consecutive nnz — regardless of row length.
//  * "stream" row blocks (rows of <= LONG_ROW nnz; every BASELINE config): a WAVEFRONT takes a
//    run of <= 64 consecutive rows whose nnz fit in its private LDS slice, writes the products
//    x[col]*val into LDS in nnz order, then one lane per row adds its row's products left to right
//    starting from zero (no workgroup barrier: a wavefront's LDS operations execute in order).  That is exactly the reference's fold (mat.rs:100-105: acc + x[col]*val, unfused),
//    so y is BIT-IDENTICAL to the reference for these rows.
//  * "vector" row blocks (rows longer than LONG_ROW): one wavefront per row, lanes stride the
//    row with coalesced loads, 64-lane __shfl_down butterfly.  Sums are re-associated
//    (documented tolerance: relative 1e-13 of sum |x*val|).
//  * Row blocks are computed once at handle creation (the analogue of mkl_sparse_optimize,
//    src/mkl_mat.rs:81-148).  The grid is persistent (4 workgroups per CU) and, with
//    xcd_chunk, each XCD walks its own contiguous chunk of row blocks so the x-gather of
//    neighbouring rows hits that XCD's L2.
//  * Optional fused epilogue: per-workgroup partials of conj(u).y, or of conj(y).y and
//    conj(y).u, so the solvers' dot products cost no extra pass over y.
#include <algorithm>
#include <memory>
#include <system_error>
#include <thread>

#include "device.hpp"

namespace sprs {

struct BlkDescHost { int32_t ra, rb, pa, nn; };   // == BlkDesc (device side)

// f(begin, end[, thread index]) over [0, n) on up to 16 host threads (fewer when n is small)
template <class F>
static void host_parallel_for(int64_t n, int64_t min_chunk, F &&f) {
    auto call = [&](int64_t a, int64_t b, int t) {
        if constexpr (std::is_invocable_v<F, int64_t, int64_t, int>) f(a, b, t); else f(a, b);
    };
    int64_t T = std::min<int64_t>(std::max(1u, std::thread::hardware_concurrency()), 16);
    T = std::max<int64_t>(1, std::min<int64_t>(T, n / std::max<int64_t>(min_chunk, 1)));
    if (T <= 1) { call(0, n, 0); return; }
    std::vector<std::thread> th;
    const int64_t per = (n + T - 1) / T;
    for (int64_t t = 0; t < T; ++t) {
        const int64_t a = t * per, b = std::min<int64_t>(n, a + per);
        if (a >= b) break;
        // a thread that cannot be started (std::system_error: resource limits) must not unwind past the joinable ones
        // already in `th` (std::terminate): its chunk runs here instead
        try { th.emplace_back(call, a, b, (int)t); } catch (const std::system_error &) { call(a, b, (int)t); }
    }
    for (auto &x : th) x.join();
}

// Per 64-row group: where its entries start, the shortest and the longest of its rows, and whether row_ptr descends
// inside it.  With these 16 bytes per group the host cuts a regular matrix (every stencil, every band of <= 7 entries
// per row) into row blocks WITHOUT seeing row_ptr: 12.5 MB cross PCIe for 50 M rows instead of 200 MB, and no per-row
// host loop runs at all.
struct alignas(16) GroupSummary { int32_t first, minlen, maxlen, bad; };
__global__ __launch_bounds__(BLOCK) void rowgroup_summary_kernel(int64_t n, int64_t ngrp, const int32_t *__restrict__ row_ptr,
                                                                 GroupSummary *__restrict__ out) {
    const int lane = threadIdx.x & (WAVE - 1);
    for (int64_t g = (int64_t)blockIdx.x * NWAVE + (threadIdx.x >> 6); g < ngrp; g += (int64_t)gridDim.x * NWAVE) {
        const int64_t r = g * ROWS_CAP + lane;
        const bool has = r < n;
        const int32_t a = row_ptr[has ? r : n - 1], b = row_ptr[has ? r + 1 : n];
        int mn = has ? b - a : INT32_MAX, mx = has ? b - a : INT32_MIN, bad = has && b < a ? 1 : 0;
        for (int off = WAVE / 2; off > 0; off >>= 1) {
            mn = min(mn, __shfl_xor(mn, off, WAVE)); mx = max(mx, __shfl_xor(mx, off, WAVE)); bad |= __shfl_xor(bad, off, WAVE);
        }
        const int32_t first = __shfl(a, 0, WAVE);
        if (lane == 0) out[g] = GroupSummary{first, mn, mx, bad};
    }
}

// Row blocks of a matrix whose row_ptr lives in HBM, from the per-group summaries.  Fills blk (row starts, VEC_FLAG on
// vector blocks, n at the end), blk_pa (first entry of every block, nnz at the end) and eq (block has rows of one length).
// Returns SPRS_OK with *done = false when the matrix is too irregular for this path (the caller then copies row_ptr).
static int rowblocks_from_summaries(sprs_csr *A, int cap, std::vector<int32_t> &blk, std::vector<int32_t> &blk_pa,
                                    std::vector<uint8_t> &eq, bool *done) {
    sprs_ctx *c = A->ctx;
    const int64_t n = A->nrows, ngrp = (n + ROWS_CAP - 1) / ROWS_CAP;
    *done = false;
    GroupSummary *d_sum = nullptr;
    SPRS_HIP_TRY(c, hipMalloc((void **)&d_sum, sizeof(GroupSummary) * (size_t)ngrp));
    std::vector<GroupSummary> sum((size_t)ngrp);
    const int g = (int)std::min<int64_t>((ngrp + NWAVE - 1) / NWAVE, 4096);
    hipLaunchKernelGGL(rowgroup_summary_kernel, dim3(g), dim3(BLOCK), 0, c->stream, n, ngrp, A->row_ptr, d_sum);
    hipError_t e1 = hipGetLastError();
    hipError_t e2 = hipMemcpyAsync(sum.data(), d_sum, sizeof(GroupSummary) * (size_t)ngrp, hipMemcpyDeviceToHost, c->stream);
    hipError_t e3 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_sum);
    SPRS_HIP_TRY(c, e1); SPRS_HIP_TRY(c, e2); SPRS_HIP_TRY(c, e3);
    // irregular groups (a long row, or more entries than a block holds) need their rows; a matrix that has many of them
    // (ragged rows, bands of 8+ entries per row) takes the per-row path on the whole row_ptr instead
    int64_t irregular = 0;
    for (int64_t q = 0; q < ngrp; ++q) {
        if (sum[(size_t)q].bad) return SPRS_INVALID_ARGUMENT;                 // row_ptr descends
        const int64_t nxt = q + 1 < ngrp ? sum[(size_t)q + 1].first : A->nnz;
        if (nxt < sum[(size_t)q].first) return SPRS_INVALID_ARGUMENT;
        irregular += sum[(size_t)q].maxlen > LONG_ROW || nxt - sum[(size_t)q].first > cap;
    }
    if (sum[0].first != 0) return SPRS_INVALID_ARGUMENT;
    if (irregular > std::max<int64_t>(16, ngrp / 1024)) return SPRS_OK;     // (each run of irregular groups costs one small blocking copy)
    blk.clear(); blk_pa.clear(); eq.clear();
    blk.reserve((size_t)ngrp + 64); blk_pa.reserve((size_t)ngrp + 65); eq.reserve((size_t)ngrp + 64);
    std::vector<int32_t> rows;                                               // row_ptr of one run of irregular groups
    for (int64_t q = 0; q < ngrp;) {
        const GroupSummary &S = sum[(size_t)q];
        const int64_t nxt = q + 1 < ngrp ? sum[(size_t)q + 1].first : A->nnz;
        if (S.maxlen <= LONG_ROW && nxt - S.first <= cap) {
            blk.push_back((int32_t)(q * ROWS_CAP)); blk_pa.push_back(S.first);
            eq.push_back(S.minlen == S.maxlen && S.maxlen >= 1 && S.maxlen <= 0x7fff);
            ++q;
            continue;
        }
        int64_t q1 = q;                                                      // the run of irregular groups [q, q1)
        while (q1 < ngrp) {
            const int64_t nx2 = q1 + 1 < ngrp ? sum[(size_t)q1 + 1].first : A->nnz;
            if (sum[(size_t)q1].maxlen <= LONG_ROW && nx2 - sum[(size_t)q1].first <= cap) break;
            ++q1;
        }
        const int64_t r0 = q * ROWS_CAP, r1 = std::min<int64_t>(q1 * ROWS_CAP, n);
        rows.resize((size_t)(r1 - r0 + 1));
        SPRS_HIP_TRY(c, hipMemcpyAsync(rows.data(), A->row_ptr + r0, sizeof(int32_t) * rows.size(), hipMemcpyDeviceToHost, c->stream));
        SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        const int32_t *rp = rows.data() - r0;                                // rp[r] for r in [r0, r1]
        for (int64_t r = r0; r < r1;) {                                      // the row-by-row walk of build_rowblocks, inside the run
            const int64_t len = (int64_t)rp[r + 1] - rp[r];
            if (len > LONG_ROW) { blk.push_back((int32_t)((uint32_t)r | VEC_FLAG)); blk_pa.push_back(rp[r]); eq.push_back(0); ++r; continue; }
            int64_t e = r + 1;
            const int64_t base = rp[r];
            bool same = true;
            while (e < r1 && e - r < ROWS_CAP) {
                const int64_t l2 = (int64_t)rp[e + 1] - rp[e];
                if (l2 > LONG_ROW || (int64_t)rp[e + 1] - base > cap) break;
                same = same && l2 == len;
                ++e;
            }
            blk.push_back((int32_t)r); blk_pa.push_back(rp[r]); eq.push_back(same && len >= 1 && len <= 0x7fff);
            r = e;
        }
        q = q1;
    }
    blk.push_back((int32_t)n); blk_pa.push_back((int32_t)A->nnz);
    *done = true;
    return SPRS_OK;
}

// Host-side analysis: greedy partition of the rows into blocks (see header comment).  rp: the host's copy of row_ptr, or
// null for a matrix built in HBM (the blocks then come from per-group summaries where the matrix is regular enough, and
// row_ptr is copied to the host only where it is not).
int build_rowblocks(sprs_csr *A, const int32_t *rp) {
    // f64: 4 entries less than the kernels' LDS slice, so that a block's 16-byte-aligned window (up to 3 entries of the
    // previous block in front) still fits two 16-byte loads per lane (spmv_wide_kernel)
    const int cap = A->dtype == DT_D ? nnz_cap_of(A->dtype) - 4 : nnz_cap_of(A->dtype);
    CreateTrace tr;
    sprs_ctx *c = A->ctx;
    std::vector<int32_t> blk, blk_pa;
    std::vector<uint8_t> eqv;
    const int64_t n = A->nrows;
    bool from_summaries = false;
    std::unique_ptr<int32_t[]> rp_own;
    if (rp == nullptr && n >= (1 << 16)) {
        int32_t last = -1;          // row_ptr[n] must be nnz before anything walks the arrays by it (the summaries check the rest)
        SPRS_HIP_TRY(c, hipMemcpyAsync(&last, A->row_ptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (last != A->nnz) return SPRS_INVALID_ARGUMENT;
        SPRS_TRY(rowblocks_from_summaries(A, cap, blk, blk_pa, eqv, &from_summaries));
        tr.lap(from_summaries ? "  row blocks from group summaries" : "  group summaries (matrix too irregular)");
    }
    if (!from_summaries && rp == nullptr) {
        // (uninitialised storage: value-initialising 200 MB for 50 M rows cost 45 ms before the copy overwrote it)
        rp_own.reset(new (std::nothrow) int32_t[(size_t)n + 1]);
        if (!rp_own) { snprintf(c->err, sizeof(c->err), "build_rowblocks: no host memory for %lld row_ptr entries", (long long)n + 1); return SPRS_ERR_HIP; }
        SPRS_HIP_TRY(c, hipMemcpyAsync(rp_own.get(), A->row_ptr, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, c->stream));
        SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        rp = rp_own.get();
        if (rp[0] != 0 || rp[(size_t)n] != A->nnz) return SPRS_INVALID_ARGUMENT;
        int worst = 0;           // (vectorisable: no early exit)
        for (int64_t i = 0; i < n; ++i) worst |= rp[i + 1] < rp[i] ? 1 : 0;
        if (worst) return SPRS_INVALID_ARGUMENT;
        tr.lap("  row_ptr to host + check");
    }
    if (!from_summaries) {
    blk.reserve((size_t)(A->nrows / ROWS_CAP + 16));
    int64_t r = 0;
    // Handle creation should stay cheap next to the solve it prepares (50 M rows: the per-row host loops were ~100 ms).
    // Which 64-row groups hold a row longer than LONG_ROW is found by all host threads; the greedy walk then takes a whole
    // 64-row block in one step wherever no such row is near and the 64 rows fit the cap (every prefix of them fits too, so
    // the row-by-row walk would have stopped at the same place) and falls back to the row-by-row walk elsewhere.
    const int64_t ngrp = (n + ROWS_CAP - 1) / ROWS_CAP;
    std::vector<uint8_t> grp_long((size_t)ngrp + 2, 0);
    host_parallel_for(ngrp, 4096, [&](int64_t g0, int64_t g1) {
        for (int64_t g = g0; g < g1; ++g) {
            const int64_t a = g * ROWS_CAP, b = std::min<int64_t>(a + ROWS_CAP, n);
            int mx = 0;
            for (int64_t q = a; q < b; ++q) mx = std::max(mx, rp[q + 1] - rp[q]);
            grp_long[(size_t)g] = mx > LONG_ROW;
        }
    });
    while (r < n) {
        if (r + ROWS_CAP <= n && !grp_long[(size_t)(r / ROWS_CAP)] && !grp_long[(size_t)((r + ROWS_CAP - 1) / ROWS_CAP)] &&
            (int64_t)rp[r + ROWS_CAP] - rp[r] <= cap) {
            blk.push_back((int32_t)r);
            r += ROWS_CAP;
            continue;
        }
        int64_t len = (int64_t)rp[r + 1] - rp[r];
        if (len > LONG_ROW) {  // vector block: one long row, its wavefront strides it
            blk.push_back((int32_t)((uint32_t)r | VEC_FLAG));
            r = r + 1;
        } else {
            int64_t e = r + 1;
            const int64_t base = rp[r];
            while (e < n && e - r < ROWS_CAP) {
                int64_t l2 = (int64_t)rp[e + 1] - rp[e];
                if (l2 > LONG_ROW || (int64_t)rp[e + 1] - base > cap) break;
                ++e;
            }
            blk.push_back((int32_t)r);
            r = e;
        }
    }
    blk.push_back((int32_t)n);
    blk_pa.resize(blk.size());
    for (size_t b = 0; b < blk.size(); ++b) blk_pa[b] = rp[(size_t)((uint32_t)blk[b] & ~VEC_FLAG)];
    tr.lap("  row-block partition");
    }
    A->n_rowblk = (int32_t)blk.size() - 1;
    {
        std::vector<BlkDescHost> desc((size_t)A->n_rowblk);
        for (int b = 0; b < A->n_rowblk; ++b) {
            const uint32_t r0 = (uint32_t)blk[b], r1 = (uint32_t)blk[b + 1];
            const int32_t ra = (int32_t)(r0 & ~VEC_FLAG), rbv = (int32_t)(r1 & ~VEC_FLAG);
            desc[b] = BlkDescHost{ra, (int32_t)((uint32_t)rbv | (r0 & VEC_FLAG)), blk_pa[b], blk_pa[b + 1] - blk_pa[b]};
        }
        SPRS_HIP_TRY(c, hipMalloc(&A->blk_desc, sizeof(BlkDescHost) * (desc.size() ? desc.size() : 1)));
        if (!desc.empty())
            SPRS_HIP_TRY(c, hipMemcpyAsync(A->blk_desc, desc.data(), sizeof(BlkDescHost) * desc.size(), hipMemcpyHostToDevice, c->stream));
        SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        // The plain-CSR kernel's own copy: stream blocks whose rows all have the same length L are flagged (rb bit 30,
        // L in nn's upper half) — the kernel then takes the row extents from the descriptor (row i starts at i*L) and
        // does not read row_ptr for the block: 4 B/row less traffic on every stencil / band interior.
        if (A->nrows < (1 << 30) && c->spmv_eqrows != 0 && from_summaries) {
            for (int b = 0; b < A->n_rowblk; ++b) {
                BlkDescHost &d = desc[(size_t)b];
                if (d.rb < 0 || !eqv[(size_t)b]) continue;
                const int rows = d.rb - d.ra;
                const int L = d.nn / rows;                                   // all rows have this length (group summary / row walk)
                d.rb = (int32_t)((uint32_t)d.rb | UNI2); d.nn = d.nn | (L << 16); A->n_eq_blocks++;
            }
        } else if (A->nrows < (1 << 30) && c->spmv_eqrows != 0) {
            std::vector<int32_t> neq(64, 0);
            host_parallel_for(A->n_rowblk, 1024, [&](int64_t b0, int64_t b1, int tid) {
                int32_t cnt = 0;
                for (int64_t b = b0; b < b1; ++b) {
                    BlkDescHost &d = desc[(size_t)b];
                    if (d.rb < 0) continue;                                  // vector block
                    const int rows = d.rb - d.ra;
                    if (rows < 1 || d.nn % rows != 0) continue;
                    const int L = d.nn / rows;
                    bool eq = L >= 1 && L <= 0x7fff;
                    for (int q = d.ra; eq && q < d.rb; ++q) eq = rp[q + 1] - rp[q] == L;
                    if (eq) { d.rb = (int32_t)((uint32_t)d.rb | UNI2); d.nn = d.nn | (L << 16); ++cnt; }
                }
                neq[(size_t)tid] = cnt;
            });
            for (int32_t v : neq) A->n_eq_blocks += v;
        }
        if (A->nrows < (1 << 30) && c->spmv_eqrows != 0) {
            SPRS_HIP_TRY(c, hipMalloc(&A->blk_desc_eq, sizeof(BlkDescHost) * (desc.size() ? desc.size() : 1)));
            if (!desc.empty())
                SPRS_HIP_TRY(c, hipMemcpyAsync(A->blk_desc_eq, desc.data(), sizeof(BlkDescHost) * desc.size(), hipMemcpyHostToDevice, c->stream));
            SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    tr.lap("  descriptors + eq flags");
    SPRS_HIP_TRY(c, hipMalloc((void **)&A->rowblk, blk.size() * sizeof(int32_t)));
    SPRS_HIP_TRY(c, hipMemcpyAsync(A->rowblk, blk.data(), blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    bool has_vec = false;
    for (int b = 0; b < A->n_rowblk; ++b) has_vec |= ((uint32_t)blk[b] & VEC_FLAG) != 0;
    if (A->dtype == DT_D && A->nnz > 0 && c->spmv_wideload != 0 && (reinterpret_cast<uintptr_t>(A->col_idx) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(A->val) & 15) == 0) {
        // zero-padded copy of the arrays' last 4-entry group for the 16-byte loads of spmv_wide_kernel
        const int64_t g_last = (A->nnz - 1) >> 2, have = A->nnz - 4 * g_last;
        SPRS_HIP_TRY(c, hipMalloc(&A->tail, 64));
        SPRS_HIP_TRY(c, hipMemsetAsync(A->tail, 0, 64, c->stream));
        SPRS_HIP_TRY(c, hipMemcpyAsync(A->tail, A->col_idx + 4 * g_last, sizeof(int32_t) * (size_t)have, hipMemcpyDeviceToDevice, c->stream));
        SPRS_HIP_TRY(c, hipMemcpyAsync(reinterpret_cast<char *>(A->tail) + 16, reinterpret_cast<const double *>(A->val) + 4 * g_last,
                                       sizeof(double) * (size_t)have, hipMemcpyDeviceToDevice, c->stream));
        SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    tr.lap("  rowblk upload + tail");
    const int st = build_dict(A, has_vec, blk, blk_pa);   // dictionary-compressed stream when the matrix qualifies (spmv_dict.hip)
    tr.lap("  build_dict total");
    return st;
}

template <class T, int DOT, bool CONJX>
__global__ __launch_bounds__(BLOCK) void spmv_kernel(int n_rowblk, int xcd_chunk, int eq_desc, const BlkDesc *__restrict__ desc,
                                                     const int32_t *__restrict__ order,
                                                     const int32_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ col_idx, const T *__restrict__ val,
                                                     const T *__restrict__ x, T *__restrict__ y,
                                                     const T *__restrict__ u, T *__restrict__ part0,
                                                     T *__restrict__ part1, const int *__restrict__ status, const Fin fin) {
    constexpr int CAP = nnz_cap<T>::value;      // per wavefront
    constexpr int ITEMS = CAP / WAVE;
    __shared__ T prod_all[NWAVE][CAP];
    __shared__ T red[NWAVE];
    // requested now, looked at when the first descriptor has arrived: the two loads share one round trip
    // (a kernel of a finished solve must not store anything; it may load)
    const int run_state = status != nullptr ? *status : (int)ST_RUNNING;

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    T *prod = prod_all[wv];
    T d0 = szero<T>(), d1 = szero<T>();

    // Persistent walk: every WAVEFRONT owns whole row blocks (<= 64 rows, <= CAP nnz) and a private LDS
    // slice, so there is no workgroup barrier anywhere in the loop — wavefronts progress independently and
    // a slow gather stalls one wavefront, not four (measured +3.6 % over the workgroup-wide variant).  The 4
    // wavefronts of a workgroup take 4 consecutive row blocks, workgroups are dealt round-robin.
    // With xcd_chunk the 8 XCDs (workgroup id mod 8, observed round-robin placement — a locality hint
    // only, never needed for correctness) each own a contiguous eighth of the row blocks.
    int b, bstep, bend;
    if (xcd_chunk) {
        const int chunk = (n_rowblk + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        b = xcd * chunk + (blockIdx.x >> 3) * NWAVE + wv;
        bstep = (gridDim.x >> 3) * NWAVE;
        bend = min(n_rowblk, (xcd + 1) * chunk);
    } else {
        b = blockIdx.x * NWAVE + wv; bstep = gridDim.x * NWAVE; bend = n_rowblk;
    }

    // Every stream load is unconditional on a clamped (always valid) address so the compiler issues
    // them back to back — 2*ITEMS stream loads, then ITEMS gathers, all in flight together (per-element
    // `if (k < nn)` branches made it serialise every gather behind an s_waitcnt vmcnt(0)); clamped
    // duplicates hit the same cache line.
    for (; b < bend; b += bstep) {
        const BlkDesc d = desc[order ? order[b] : b];
        if (run_state != ST_RUNNING) { fin_idle(fin, DOT == 2); return; }       // uniform over the grid; nothing has been stored yet
        // eq_desc: the descriptors are the flagged copy (bit 30 of rb = equal-length rows; only built when nrows < 2^30)
        const bool eq_rows = eq_desc != 0 && ((uint32_t)d.rb & UNI2) != 0;
        const int ra = d.ra, rb = d.rb & (eq_desc ? 0x3fffffff : 0x7fffffff);
        if (d.rb >= 0) {
            // ---------------- stream block: products to LDS, then one lane per row
            const int pa = d.pa, nn = eq_desc ? (d.nn & 0xffff) : d.nn;
            // row extents and the dot operand for the reduce phase: requested now, used after the products
            const int r = ra + lane;
            const bool has_row = r < rb;
            const int rcl = has_row ? r : rb - 1;
            int s, e;
            if (eq_rows) {                           // equal-length rows: extents from the descriptor, row_ptr is not read
                const int L = d.nn >> 16;
                s = (rcl - ra) * L; e = s + L;
            } else {
                s = row_ptr[rcl] - pa; e = row_ptr[rcl + 1] - pa;
            }
            [[maybe_unused]] T uu;
            if (DOT != 0) uu = u[rcl];
            if (nn > 0) {
                const int last = nn - 1;
                int cidx[ITEMS];
                T vv[ITEMS], xg[ITEMS];
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) {
                    const int k = min(lane + i * WAVE, last);
                    cidx[i] = col_idx[pa + k];
                    vv[i] = val[pa + k];
                }
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) xg[i] = x[cidx[i]];
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) {
                    const int k = lane + i * WAVE;
                    if (k < nn) prod[k] = smul(CONJX ? sconj(xg[i]) : xg[i], vv[i]);   // mat.rs:104  x[col] * val
                }
            }
            wave_lds_fence();
            if (has_row) {
                T acc = szero<T>();                               // mat.rs:103  fold(T::zero(), ..)
                const int len = e - s;
                if (len <= 8) {
                    // short rows (every stencil): fetch up to 8 products at once, add the valid ones in order
                    T pv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) pv[j] = prod[min(s + j, CAP - 1)];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j < len) acc = sadd(acc, pv[j]);
                } else {
                    for (int k = s; k < e; ++k) acc = sadd(acc, prod[k]);
                }
                y[r] = acc;
                if (DOT == 1) d0 = sadd(d0, smul(sconj(uu), acc));
                if (DOT == 2) { d0 = sadd(d0, smul(sconj(acc), acc)); d1 = sadd(d1, smul(sconj(acc), uu)); }
            }
            wave_lds_fence();  // prod[] is rewritten by this wavefront's next row block
        } else {
            // ---------------- vector block: this wavefront strides one long row
            const int r = ra;
            const int s = row_ptr[r], e = row_ptr[r + 1];
            T acc = szero<T>();
            for (int k = s + lane; k < e; k += WAVE) { const int cj = col_idx[k]; const T vj = val[k]; acc = sadd(acc, smul(CONJX ? sconj(x[cj]) : x[cj], vj)); }
            acc = wave_sum(acc);
            if (lane == 0) {
                y[r] = acc;
                if (DOT == 1) d0 = sadd(d0, smul(sconj(u[r]), acc));
                if (DOT == 2) { d0 = sadd(d0, smul(sconj(acc), acc)); d1 = sadd(d1, smul(sconj(acc), u[r])); }
            }
        }
    }
    if (run_state != ST_RUNNING) { fin_idle(fin, DOT == 2); return; }           // a wavefront that had no row block comes straight here
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

// ---------------------------------------------------------------------------------------------
// f64, 16 bytes per lane (round 3, profiles/r03_tuning.md §2).  The kernel above keeps 8 + 8 stream loads of 4 and 8 bytes
// per lane in flight per wavefront.  Here lane l takes entries 4l .. 4l + 3 of the block's 16-byte-ALIGNED window
// [pa - (pa & 3), ..): two loads of col_idx and four of val per 512 entries, the same gathers, the products written to LDS
// in window order, the same per-row left fold (y bit-identical).  By itself that changes little at four workgroups per
// CU; with wider loads three per CU become the optimum (fewer wavefronts, each with the same 6 KB in flight), and that
// shape is 6 - 10 % faster on the cfg-5 stream (1085 - 1107 -> 1008 - 1027 us on three boxes, same binary, back to back).
// Needs 16-byte aligned col_idx / val arrays and blocks of <= 508 entries (build_rowblocks); the last aligned group of
// the arrays, which may reach past their end, is read from a zero-padded copy (tail_c / tail_v).
typedef int v4i32 __attribute__((ext_vector_type(4)));
struct alignas(16) D2v { double a, b; };
template <int DOT, bool YNT>
__global__ __launch_bounds__(BLOCK) void spmv_wide_kernel(int n_rowblk, int xcd_chunk, int eq_desc, const BlkDesc *__restrict__ desc,
                                                          const int32_t *__restrict__ order, const int32_t *__restrict__ row_ptr,
                                                          const int32_t *__restrict__ col_idx, const double *__restrict__ val,
                                                          const v4i32 *__restrict__ tail_c, const D2v *__restrict__ tail_v, int g_last,
                                                          const double *__restrict__ x, double *__restrict__ y,
                                                          const double *__restrict__ u, double *__restrict__ part0,
                                                          double *__restrict__ part1, const int *__restrict__ status, const Fin fin) {
    using T = double;
    constexpr int CAP = nnz_cap<T>::value;
    __shared__ __attribute__((aligned(16))) T prod_all[NWAVE][CAP + 8];
    __shared__ T red[NWAVE];
    const int run_state = status != nullptr ? *status : (int)ST_RUNNING;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    T *prod = prod_all[wv];
    T d0 = 0.0, d1 = 0.0;
    int b, bstep, bend;
    if (xcd_chunk) {
        const int chunk = (n_rowblk + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        b = xcd * chunk + (blockIdx.x >> 3) * NWAVE + wv;
        bstep = (gridDim.x >> 3) * NWAVE;
        bend = min(n_rowblk, (xcd + 1) * chunk);
    } else {
        b = blockIdx.x * NWAVE + wv; bstep = gridDim.x * NWAVE; bend = n_rowblk;
    }
    const v4i32 *col4 = reinterpret_cast<const v4i32 *>(col_idx);
    const D2v *val2 = reinterpret_cast<const D2v *>(val);
    for (; b < bend; b += bstep) {
        const BlkDesc d = desc[order ? order[b] : b];
        if (run_state != ST_RUNNING) { fin_idle(fin, DOT == 2); return; }
        const bool eq_rows = eq_desc != 0 && ((uint32_t)d.rb & UNI2) != 0;
        const int ra = d.ra, rb = d.rb & (eq_desc ? 0x3fffffff : 0x7fffffff);
        if (d.rb >= 0) {
            const int pa = d.pa, nn = eq_desc ? (d.nn & 0xffff) : d.nn;
            const int shift = pa & 3, tot = nn + shift;                     // the aligned window starts `shift` entries before the block
            const int r = ra + lane;
            const bool has_row = r < rb;
            const int rcl = has_row ? r : rb - 1;
            int s, e;
            if (eq_rows) { const int L = d.nn >> 16; s = (rcl - ra) * L; e = s + L; }
            else { s = row_ptr[rcl] - pa; e = row_ptr[rcl + 1] - pa; }
            [[maybe_unused]] T uu;
            if (DOT != 0) uu = u[rcl];
            if (nn > 0) {
                const int g0 = (pa - shift) >> 2, lastq = (tot - 1) >> 2;
                v4i32 c[2]; D2v v[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int G = g0 + min(lane + i * WAVE, lastq);
                    const bool tl = G == g_last;                            // the arrays' last group: from the padded copy
                    const v4i32 *cp = tl ? tail_c : col4 + G;
                    const D2v *vp = tl ? tail_v : val2 + 2 * (int64_t)G;
                    c[i] = *cp; v[i][0] = vp[0]; v[i][1] = vp[1];
                }
                T xg[8];
#pragma unroll
                for (int i = 0; i < 2; ++i) { xg[4 * i] = x[c[i].x]; xg[4 * i + 1] = x[c[i].y]; xg[4 * i + 2] = x[c[i].z]; xg[4 * i + 3] = x[c[i].w]; }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = 4 * (lane + i * WAVE);
                    // mat.rs:104  x[col] * val — entries of the window outside the block are multiplied too and never read
                    const D2v p0{xg[4 * i] * v[i][0].a, xg[4 * i + 1] * v[i][0].b}, p1{xg[4 * i + 2] * v[i][1].a, xg[4 * i + 3] * v[i][1].b};
                    if (k < tot) { *reinterpret_cast<D2v *>(prod + k) = p0; *reinterpret_cast<D2v *>(prod + k + 2) = p1; }
                }
            }
            wave_lds_fence();
            if (has_row) {
                T acc = 0.0;                                          // mat.rs:103  fold(T::zero(), ..)
                const int len = e - s;
                const T *pr = prod + shift;
                if (len <= 8) {
                    T pv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) pv[j] = pr[min(s + j, CAP + 3)];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j < len) acc = acc + pv[j];
                } else {
                    for (int k = s; k < e; ++k) acc = acc + pr[k];
                }
                if constexpr (YNT) __builtin_nontemporal_store(acc, y + r);
                else y[r] = acc;
                if (DOT == 1) d0 = d0 + uu * acc;
                if (DOT == 2) { d0 = d0 + acc * acc; d1 = d1 + acc * uu; }
            }
            wave_lds_fence();
        } else {
            // ---------------- vector block: this wavefront strides one long row (as in spmv_kernel)
            const int r = ra;
            const int s = row_ptr[r], e = row_ptr[r + 1];
            T acc = 0.0;
            for (int k = s + lane; k < e; k += WAVE) { const int cj = col_idx[k]; const T vj = val[k]; acc = acc + x[cj] * vj; }
            acc = wave_sum(acc);
            if (lane == 0) {
                y[r] = acc;
                if (DOT == 1) d0 = d0 + u[r] * acc;
                if (DOT == 2) { d0 = d0 + acc * acc; d1 = d1 + acc * u[r]; }
            }
        }
    }
    if (run_state != ST_RUNNING) { fin_idle(fin, DOT == 2); return; }
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

// ---------------------------------------------------------------------------------------------
// Row-block column spans (used by the optional schedules and by the distributed interior/boundary split).
__global__ __launch_bounds__(BLOCK) void rowblk_span_kernel(int n_rowblk, const int32_t *__restrict__ rowblk,
                                                            const int32_t *__restrict__ row_ptr,
                                                            const int32_t *__restrict__ col_idx, int32_t *__restrict__ lo,
                                                            int32_t *__restrict__ hi) {
    // one wavefront per row block
    const int lane = threadIdx.x & (WAVE - 1);
    for (int b = blockIdx.x * NWAVE + (threadIdx.x >> 6); b < n_rowblk; b += gridDim.x * NWAVE) {
        const int ra = (int)((uint32_t)rowblk[b] & ~VEC_FLAG), rb = (int)((uint32_t)rowblk[b + 1] & ~VEC_FLAG);
        const int pa = row_ptr[ra], pb = row_ptr[rb];
        int mn = INT32_MAX, mx = -1;
        for (int k = pa + lane; k < pb; k += WAVE) { const int cidx = col_idx[k]; mn = min(mn, cidx); mx = max(mx, cidx); }
        for (int off = WAVE / 2; off > 0; off >>= 1) { mn = min(mn, __shfl_down(mn, off, WAVE)); mx = max(mx, __shfl_down(mx, off, WAVE)); }
        if (lane == 0) { lo[b] = mn; hi[b] = mx; }
    }
}

// Device-side validation of a CSR that was built in HBM (the host never sees col_idx): every column
// index must address x.  A malformed matrix must be refused here, not fault the GPU in the SpMV.
__global__ __launch_bounds__(BLOCK) void validate_cols_kernel(int64_t nnz, const int32_t *__restrict__ col_idx, int32_t ncols,
                                                              int *__restrict__ bad) {
    int local = 0;
    for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * BLOCK) {
        const int32_t cidx = col_idx[k];
        local |= (cidx < 0) | (cidx >= ncols);
    }
    if (local) atomicOr(bad, 1);
}
int validate_cols_device(const sprs_csr *A) {
    sprs_ctx *c = A->ctx;
    if (A->nnz == 0) return SPRS_OK;
    CtxLock lock(c);
    int *d_bad = reinterpret_cast<int *>(c->d_scal);
    SPRS_HIP_TRY(c, hipMemsetAsync(d_bad, 0, sizeof(int), c->stream));
    const int g = (int)std::min<int64_t>((A->nnz + BLOCK - 1) / BLOCK, 2048);
    hipLaunchKernelGGL(validate_cols_kernel, dim3(g), dim3(BLOCK), 0, c->stream, A->nnz, A->col_idx, (int32_t)A->ncols, d_bad);
    SPRS_HIP_TRY(c, hipGetLastError());
    int bad = 0;
    SPRS_HIP_TRY(c, hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return bad ? SPRS_INVALID_ARGUMENT : SPRS_OK;
}

int rowblk_spans(const sprs_csr *A, std::vector<int32_t> &lo, std::vector<int32_t> &hi) {
    sprs_ctx *c = A->ctx;
    const int nb = A->n_rowblk;
    lo.assign((size_t)nb, INT32_MAX); hi.assign((size_t)nb, -1);
    if (nb == 0) return SPRS_OK;
    int32_t *d_lo = nullptr, *d_hi = nullptr;
    SPRS_HIP_TRY(c, hipMalloc((void **)&d_lo, sizeof(int32_t) * nb));
    SPRS_HIP_TRY(c, hipMalloc((void **)&d_hi, sizeof(int32_t) * nb));
    const int g = std::max(1, std::min(2048, (nb + NWAVE - 1) / NWAVE));
    hipLaunchKernelGGL(rowblk_span_kernel, dim3(g), dim3(BLOCK), 0, c->stream, nb, A->rowblk, A->row_ptr, A->col_idx, d_lo, d_hi);
    hipError_t e1 = hipMemcpyAsync(lo.data(), d_lo, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipMemcpyAsync(hi.data(), d_hi, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, c->stream);
    hipError_t e3 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_lo); (void)hipFree(d_hi);
    SPRS_HIP_TRY(c, e1); SPRS_HIP_TRY(c, e2); SPRS_HIP_TRY(c, e3);
    return SPRS_OK;
}

// number of workgroups launch_spmv uses == number of partials it writes
// Matrices whose whole stream fits the 256 MiB Infinity Cache behave differently from HBM-bound ones
// (profiles/r01_tuning.md): they want one contiguous chunk of row blocks per XCD (x stays in that XCD's
// L2); HBM-bound ones want the row blocks dealt round-robin over the XCDs.
// (is_cache_resident: internal.hpp)
// the 16-byte-per-lane kernel runs this handle's plain stream (f64, aligned arrays, knob "spmv_wideload")
static inline bool wide_loads(const sprs_csr *A) { return A->tail != nullptr && A->ctx->spmv_wideload != 0 && dict_mode(A) == 0; }
static inline int base_grid(const sprs_csr *A) {
    int g = A->ctx->spmv_grid;
    // measured (A/B on the full solve): 4 workgroups per CU, for HBM-bound and cache-resident matrices and for all
    // three streams alike (the pair-code kernel runs its stand-alone best at 6 per CU but loses that inside the
    // solve, where it alternates with the BLAS-1 kernels) — except the 16-byte-per-lane plain kernel on an HBM-sized
    // stream: 3 per CU (profiles/r03_tuning.md §2; only multiples of the CU count spread evenly)
    if (g <= 0) {
        g = A->ctx->num_cu * ((wide_loads(A) && !is_cache_resident(A)) ? 3 : 4);
        // LDS-window tiles are coarse work items (4096 rows): a matrix with few of them gets fewer workgroups, about three
        // tiles each, rather than 1024 workgroups with one or two (the N = 8 slab: 1525 tiles; profiles/r03_tuning.md §9)
        if (chain_plan_used(A)) {
            g = A->ctx->num_cu * 2;           // plane-streaming chains: three x windows = 72 KiB of LDS, two workgroups per CU, about one chain segment each
        } else if (tile_plan_used(A)) {
            const int nt = dict_mode(A) == 2 ? A->dict->tile_pair.n_tile : A->dict->tile_off.n_tile;
            g = std::min(g, std::max(64, (nt / 3) & ~7));
        }
    }
    if (g < 8) g = 8;
    if (g > MAX_GRID / 2) g = MAX_GRID / 2;
    return g & ~7;
}
static inline int grid_for_blocks(const sprs_csr *A, int count) {
    const int g = base_grid(A);
    // at least one row block per wavefront, keep it a multiple of 8 (one slice per XCD)
    int need = (((count + NWAVE - 1) / NWAVE + 7) / 8) * 8;
    if (need < 8) need = 8;
    return g < need ? g : need;
}
static inline int spmv_grid(const sprs_csr *A) { return grid_for_blocks(A, A->n_rowblk); }
int spmv_subset_grid(const sprs_csr *A, int count) { return grid_for_blocks(A, count); }

// The fused recurrence kernels deal their tiles round-robin over the workgroups, i.e. over the XCDs.  Where the SpMV
// gives every XCD one contiguous eighth of the rows (cache-resident matrices, xcd_chunk) and nearly all of a row's columns
// lie inside that eighth (far band <= 1/4 of it), the vector kernels take the same eighths: what K1 writes is gathered by
// the SpMV from the same XCD's L2 instead of crossing the fabric (cfg 2: +4.5 %, cfg 3 / 4: +1-2 %; a 500x500x4 grid,
// whose plane neighbours live two XCDs away, loses 6 % with it — profiles/r02_tuning.md §22).
bool fused_chunked(const sprs_csr *A) {
    const sprs_ctx *c = A->ctx;
    if (c->ew_chunk >= 0) return c->ew_chunk != 0;
    if (A->dist || c->xcd_chunk == 0 || !is_cache_resident(A)) return false;
    return A->dict && A->dict->max_off >= 0 && A->dict->max_off * 32 <= (int64_t)A->nrows;
}

template <class T>
static int launch_spmv_impl(const sprs_csr *A, const int32_t *order, int count, int g, const T *x, T *y, int dot_mode,
                            const T *u, T *part0, T *part1, const int *status, bool conj_x, const Fin *finp) {
    const Fin fin = (finp && dot_mode != 0) ? *finp : Fin{};
    sprs_ctx *c = A->ctx;
    const T *v = reinterpret_cast<const T *>(A->val);
    // cache-resident matrices: one contiguous chunk of row blocks per XCD; HBM-bound ones: round-robin (see above)
    const int xcd_chunk = c->xcd_chunk < 0 ? (is_cache_resident(A) ? 1 : 0) : c->xcd_chunk;
    if (const int dm = dict_mode(A))
        return launch_spmv_dict<T>(A, dm, order, count, g, xcd_chunk, x, y, dot_mode, u, part0, part1, status, conj_x, fin);
    if constexpr (dtype_of<T>::value == DT_D) {
        if (wide_loads(A)) {
            const BlkDesc *dsc = reinterpret_cast<const BlkDesc *>(A->blk_desc_eq ? A->blk_desc_eq : A->blk_desc);
            const v4i32 *tc = reinterpret_cast<const v4i32 *>(A->tail);
            const D2v *tv = reinterpret_cast<const D2v *>(reinterpret_cast<const char *>(A->tail) + 16);
            const int g_last = (int)((A->nnz - 1) >> 2);
#define SPRS_WIDE(D, YN)                                                                                             \
    SPRS_LAUNCH_SPMV(c, (spmv_wide_kernel<D, YN>), g, count, xcd_chunk, A->blk_desc_eq ? 1 : 0, dsc, order, A->row_ptr, \
                     A->col_idx, v, tc, tv, g_last, x, y, u, part0, part1, status, fin)
            if (stream_loads_nt(c, (size_t)A->nrows * sizeof(T))) {       // HBM-sized result: non-temporal y stores (-1 %)
                if (dot_mode == 0) SPRS_WIDE(0, true); else if (dot_mode == 1) SPRS_WIDE(1, true); else SPRS_WIDE(2, true);
            } else {
                if (dot_mode == 0) SPRS_WIDE(0, false); else if (dot_mode == 1) SPRS_WIDE(1, false); else SPRS_WIDE(2, false);
            }
#undef SPRS_WIDE
            SPRS_HIP_TRY(c, hipGetLastError());
            return SPRS_OK;
        }
    }
#define SPRS_SPMV(D, CJ)                                                                                              \
    SPRS_LAUNCH_SPMV(c, (spmv_kernel<T, D, CJ>), g, count,                                                            \
                       xcd_chunk, A->blk_desc_eq ? 1 : 0, reinterpret_cast<const BlkDesc *>(A->blk_desc_eq ? A->blk_desc_eq : A->blk_desc), order, A->row_ptr, A->col_idx, v, x, y, u, part0, part1, status, fin)
    if (conj_x && is_complex<T>::value) {  // only CSMINRES on complex data needs the conjugated gather
        if (dot_mode == 0) SPRS_SPMV(0, true);
        else if (dot_mode == 1) SPRS_SPMV(1, true);
        else SPRS_SPMV(2, true);
    } else {
        if (dot_mode == 0) SPRS_SPMV(0, false);
        else if (dot_mode == 1) SPRS_SPMV(1, false);
        else SPRS_SPMV(2, false);
    }
#undef SPRS_SPMV
    SPRS_HIP_TRY(c, hipGetLastError());
    return SPRS_OK;
}

template <class T>
int launch_spmv(const sprs_csr *A, const T *x, T *y, int dot_mode, const T *u, T *part0, T *part1, const int *status,
                bool conj_x, const Fin *fin) {
    return launch_spmv_impl<T>(A, nullptr, (int)A->n_rowblk, spmv_grid(A), x, y, dot_mode, u, part0, part1, status, conj_x, fin);
}

template <class T>
int launch_spmv_subset(const sprs_csr *A, const int32_t *order, int count, const T *x, T *y, int dot_mode, const T *u,
                       T *part0, T *part1, const int *status, bool conj_x, const Fin *fin) {
    return launch_spmv_impl<T>(A, order, count, grid_for_blocks(A, count), x, y, dot_mode, u, part0, part1, status, conj_x, fin);
}

// distributed operators with an interior/boundary split run two launches whose partials are concatenated
int spmv_num_partials(const sprs_csr *A) {
    if (A->dist && A->dist->order_int) return grid_for_blocks(A, A->dist->n_int) + grid_for_blocks(A, A->dist->n_bnd);
    return spmv_grid(A);
}

#define SPRS_INST_SPMV(T)                                                                                              \
    template int launch_spmv<T>(const sprs_csr *, const T *, T *, int, const T *, T *, T *, const int *, bool, const Fin *); \
    template int launch_spmv_subset<T>(const sprs_csr *, const int32_t *, int, const T *, T *, int, const T *, T *, T *, const int *, bool, const Fin *);
SPRS_INST_SPMV(double)
SPRS_INST_SPMV(cplx)
SPRS_INST_SPMV(float)
SPRS_INST_SPMV(cplxf)

}  // namespace sprs
