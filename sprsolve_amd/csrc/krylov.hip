// Krylov recurrences (host side, C++) over device-resident vectors and scalars.
//
// Reference: src/bicg_stab.rs:35-366, src/minres.rs:31-341, src/cs_minres.rs:29-158.
//
// Two execution modes per solver (sprs_solver_set_mode):
//  * fused (default): the reference's 13 (BiCGStab) / 11 (MINRES) full-vector passes per
//    iteration are regrouped into 5 / 3 kernels.  Every scalar of the recurrence (rho, alpha, w,
//    beta, Givens c/s, ...) lives in HBM: a kernel that needs the result of a dot product
//    re-reduces that product's per-workgroup partials in its prologue (same partials, same
//    order in every workgroup => bit-identical scalars everywhere) and workgroup 0 records the
//    scalar for later kernels.  The host never waits for a scalar; it polls a status word
//    every `poll` iterations.  After convergence / breakdown / restart-request every later
//    kernel returns at its first instruction, so x, r and the iteration number are exactly the
//    reference's at the moment of the event.
//    The arithmetic of every element keeps the reference's rounding sequence (e.g.
//    y = (v*(-beta*w) + y*beta) + r*1, bicg_stab.rs:155-156); only the summation order of the
//    dot products / norms differs from the reference's serial fold.
//  * literal: one kernel per reference op, scalars consumed on the host where the reference
//    consumes them.  Slow (5 host syncs per iteration); kept as the on-GPU cross-check.
#include "krylov.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <type_traits>
#include <utility>

#include "bicg_fuse.hpp"
#include "minres_fuse.hpp"
#include "device.hpp"

namespace sprs {


// ======================================================================= fused kernel skeleton
// NT: the vector operands are read and the results written with non-temporal accesses.  At HBM sizes every vector is
// streamed once per pass and evicted long before its next use; accesses that do not allocate on the way leave the caches
// to the SpMV's gathers and run faster in the read/write mix (profiles/r02_tuning.md §20).  Vectors that live in the
// Infinity Cache lose with it.
template <int PK, bool NT, class F>
__global__ __launch_bounds__(BLOCK) void fused_kernel(int64_t n, F f, int chunked) {
    if (!f.prologue()) return;
    if (chunked) {
        // one contiguous eighth of the vectors per XCD (workgroup id mod 8), the same eighth in every kernel and the one
        // whose rows that XCD multiplies in the SpMV (xcd_chunk): what a kernel writes is read from the same L2
        const int64_t np = n / PK, chunk = ((np + 7) / 8 + BLOCK - 1) / BLOCK * BLOCK;
        const int xcd = blockIdx.x & 7;
        const int64_t end = min(np, (int64_t)(xcd + 1) * chunk), st = (int64_t)(gridDim.x >> 3) * BLOCK;
        for (int64_t i = xcd * chunk + (int64_t)(blockIdx.x >> 3) * BLOCK + threadIdx.x; i < end; i += st) f.template run<PK, NT>(i);
    } else {
        SPRS_FOREACH_PACK(n, PK, i) f.template run<PK, NT>(i);
    }
    if (PK > 1) {
        int64_t i = (n / PK) * PK + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
        if (i < n) f.template run<1, NT>(i);
    }
    f.epilogue();
}

template <class T, class F>
static int launch_fused(sprs_ctx *c, size_t n, int grid, int chunked_walk, F f) {
    constexpr int PKW = pack_width<T>::value;
    const int chunked = (chunked_walk && grid % 8 == 0 && grid >= 8) ? 1 : 0;
    if (stream_loads_nt(c, n * sizeof(T)))
        hipLaunchKernelGGL((fused_kernel<PKW, true, F>), dim3(grid), dim3(BLOCK), 0, c->stream, (int64_t)n, f, chunked);
    else
        hipLaunchKernelGGL((fused_kernel<PKW, false, F>), dim3(grid), dim3(BLOCK), 0, c->stream, (int64_t)n, f, chunked);
    SPRS_HIP_TRY(c, hipGetLastError());
    return SPRS_OK;
}

// (first_thread(), BicgK1 and BicgK3 live in bicg_fuse.hpp: the plane-streaming chain SpMV runs their prologues and their
// element-wise updates inside its own launch — "fused SpMV input" below)

// K5  bicg_stab.rs:178-196:  w = (t.t > 0) ? t.r / t.t : 0 ; x -= alpha*y ; x -= w*s ; r -= w*t
//     + partials of norm2(r)^2 and r0.r for the next iteration's K1 (:123,:128)
// SV (fused SpMV input, below): K4 formed s = r + v (-alpha) on the fly and did NOT store it — a store of s costs the fused K4
// 80 of its 294 us (scripts/micro/fused_window.hip, profiles/r04_tuning.md §3) — so s is formed again here from r and v with the
// same expression (K3's, bicg_stab.rs:172): one more read, which finds r and v in the Infinity Cache behind K4's reads; r' = s - w t
// is written over r.
template <class T, bool PC, bool SV = false>
struct BicgK5 {
    BicgState<T> *S; const T *partTT; const T *partTR; int P;
    const T *y; const T *z; const T *t; const T *r0; T *x; T *r; Real<T> *partN; T *partRho;
    Fin fin;                    // distributed: the last workgroup reduces (partN, partRho) for the all-reduce
    T na, nw, w;
    Real<T> accN; T accR;
    unsigned int tag = 0; unsigned long long mb_timeout = 0;     // peer-to-peer hand-off (see BicgK1): partTT = this rank's mailbox entries
    const T *sv = nullptr;                                       // SV: the v of s = r + v (-alpha)
    __device__ __forceinline__ bool prologue() {
        __shared__ T smT[NWAVE];
        __shared__ T smT2[NWAVE];
        const int status = S->status;                               // requested together with the partials
        const T alpha = S->alpha;
        T tt, tr;
        if (tag != 0) {
            if (status != ST_RUNNING) { fin_idle(fin, true); return false; }
            if (!mbox_sum2(MboxSrc{reinterpret_cast<const unsigned long long *>(partTT), P, tag, mb_timeout}, tt, tr)) {
                if (first_thread()) S->status = ST_COMM_TIMEOUT;
                return false;
            }
        } else {
            reduce_partials2(partTT, partTR, P, smT, smT2, tt, tr); // :178, :183
        }
        if (status != ST_RUNNING) { fin_idle(fin, true); return false; }
        w = (sre(tt) > 0.0) ? sdiv(tr, tt) : szero<T>();            // :179-186
        na = sneg(alpha); nw = sneg(w);
        accN = 0.0; accR = szero<T>();
        return true;
    }
    template <int PK, bool NT> __device__ __forceinline__ void run(int64_t i) {
        auto xv = ldp<T, PK, NT>(x, i); auto yv = ldp<T, PK, NT>(y, i); auto rv = ldp<T, PK, NT>(r, i);
        auto tv = ldp<T, PK, NT>(t, i); auto qv = ldp<T, PK, NT>(r0, i);
        [[maybe_unused]] Pack<T, PK> zv;
        if (PC) zv = ldp<T, PK, NT>(z, i);
        if constexpr (SV) {
            const auto vv = ldp<T, PK, NT>(sv, i);
#pragma unroll
            for (int e = 0; e < PK; ++e) rv.v[e] = sadd(rv.v[e], smul(vv.v[e], na));    // :172  s = r - alpha v (K3's expression)
        }
#pragma unroll
        for (int e = 0; e < PK; ++e) {
            T xx = sadd(xv.v[e], smul(yv.v[e], na));                // :188
            xx = sadd(xx, smul(PC ? zv.v[e] : rv.v[e], nw));        // :191 / :357
            xv.v[e] = xx;
            const T rr = sadd(rv.v[e], smul(tv.v[e], nw));          // :196
            rv.v[e] = rr;
            accN = accN + ssq(rr);
            accR = sadd(accR, smul(sconj(qv.v[e]), rr));
        }
        stp<T, PK, NT>(x, i, xv);
        stp<T, PK, NT>(r, i, rv);
    }
    __device__ __forceinline__ void epilogue() {
        __shared__ Real<T> smD[NWAVE];
        __shared__ T smT[NWAVE];
        const Real<T> sN = block_sum(accN, smD);
        const T sR = block_sum(accR, smT);
        if (threadIdx.x == 0) { st_partial(fin, partN + blockIdx.x, sN); st_partial(fin, partRho + blockIdx.x, sR); }
        if (first_thread()) { S->w = w; S->rho_old = S->rho; S->its = S->its + 1; }
        if (fin.counter) finalize_last_block<Real<T>, T>(fin, true, smD, smT);
    }
};

// ======================================================================= MINRES kernels
// M2  minres.rs:117-120 (+ :276-278):  v_new -= beta*v_old ; v_new -= alpha*v ;
//     partials of |v_new|^2   or, preconditioned,  w_new = M^-1 v_new and conj(v_new).w_new
template <class T, class V, bool PC>
struct MinresM2 {
    MinresDev<T> *D; int par; const T *partAlpha; int P;
    const T *v_old; const T *v; T *v_new; const V *dinv; T *w_new; Real<T> *partBeta; T *partBeta2;
    Fin fin;                    // distributed: the last workgroup reduces partBeta / partBeta2 for the all-reduce
    T nb, na; Real<T> accD; T accT;
    unsigned int tag = 0; unsigned long long mb_timeout = 0;     // peer-to-peer hand-off (see BicgK1): partAlpha = this rank's mailbox entries
    __device__ __forceinline__ bool prologue() {
        __shared__ T smT[NWAVE];
        const int status = D->status;                               // requested together with the partials
        const Real<T> beta = D->st[par].beta;
        T alpha;
        if (tag != 0) {
            if (status != ST_RUNNING) { fin_idle(fin, false); return false; }
            if (!mbox_sum1(MboxSrc{reinterpret_cast<const unsigned long long *>(partAlpha), P, tag, mb_timeout}, alpha)) {
                if (first_thread()) D->status = ST_COMM_TIMEOUT;
                return false;
            }
        } else {
            alpha = reduce_partials(partAlpha, P, smT);             // :116
        }
        if (status != ST_RUNNING) { fin_idle(fin, false); return false; }
        nb = sfromr<T>(-beta);                                      // :117 T::from_real(-beta)
        na = sneg(alpha);                                           // :118
        accD = 0.0; accT = szero<T>();
        if (first_thread()) D->st[par].alpha = alpha;
        return true;
    }
    template <int PK, bool NT> __device__ __forceinline__ void run(int64_t i) {
        auto nv = ldp<T, PK, NT>(v_new, i); auto ov = ldp<T, PK, NT>(v_old, i); auto cv = ldp<T, PK, NT>(v, i);
        Pack<T, PK> wv;
        [[maybe_unused]] Pack<V, PK> dv;
        if (PC) dv = ldp<V, PK, NT>(dinv, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) {
            T t = sadd(nv.v[e], smul(ov.v[e], nb));                 // :117
            t = sadd(t, smul(cv.v[e], na));                         // :118
            nv.v[e] = t;
            if (PC) {
                wv.v[e] = smulv(t, dv.v[e]);                        // :276
                accT = sadd(accT, smul(sconj(t), wv.v[e]));         // :278
            } else {
                accD = accD + ssq(t);                               // :120
            }
        }
        stp<T, PK, NT>(v_new, i, nv);
        if (PC) stp<T, PK, NT>(w_new, i, wv);
    }
    __device__ __forceinline__ void epilogue() {
        __shared__ Real<T> smD[NWAVE];
        __shared__ T smT[NWAVE];
        if (PC) {
            const T s = block_sum(accT, smT);
            if (threadIdx.x == 0) st_partial(fin, partBeta2 + blockIdx.x, s);
            if (fin.counter) finalize_last_block<T, T>(fin, false, smT, smT);
        } else {
            const Real<T> s = block_sum(accD, smD);
            if (threadIdx.x == 0) st_partial(fin, partBeta + blockIdx.x, s);
            if (fin.counter) finalize_last_block<Real<T>, Real<T>>(fin, false, smD, smD);
        }
    }
};

// (MinresM3 lives in minres_fuse.hpp: the lane-per-row SpMV of the compressed streams can run it inside its own launch —
// "M3 inside M1" below)

// ======================================================================= KrylovBase
template <class T>
int KrylovBase<T>::init(const sprs_csr *A_, size_t size, int nvec_) {
    A = A_; ctx = A_->ctx; n = size; nvec = nvec_;
    // distributed: every work vector carries the halo tail (sparse halo) or is padded to the all-gather slice
    const size_t nx = !A->dist ? n : (A->dist->ag_slice > 0 ? std::max<size_t>(n, (size_t)A->dist->ag_slice) : (size_t)A->ncols);
    stride = (nx + 31) & ~(size_t)31;
    if (stride == 0) stride = 32;
    SPRS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    SPRS_HIP_TRY(ctx, hipMalloc((void **)&work, sizeof(T) * stride * (size_t)nvec));
    SPRS_HIP_TRY(ctx, hipMemsetAsync(work, 0, sizeof(T) * stride * (size_t)nvec, ctx->stream));   // vec![T::zero(); size*7]
    SPRS_HIP_TRY(ctx, hipMalloc((void **)&part, sizeof(T) * MAX_GRID * 8));
    SPRS_HIP_TRY(ctx, hipMalloc((void **)&partD, sizeof(Real<T>) * MAX_GRID * 4));
    SPRS_HIP_TRY(ctx, hipMemsetAsync(part, 0, sizeof(T) * MAX_GRID * 8, ctx->stream));
    SPRS_HIP_TRY(ctx, hipMemsetAsync(partD, 0, sizeof(Real<T>) * MAX_GRID * 4, ctx->stream));
    if (A->dist) {
        SPRS_HIP_TRY(ctx, hipMalloc((void **)&red, sizeof(double) * 32));
        SPRS_HIP_TRY(ctx, hipMemsetAsync(red, 0, sizeof(double) * 32, ctx->stream));
        SPRS_HIP_TRY(ctx, hipMalloc((void **)&fin_counter, sizeof(unsigned int) * 64));      // one arrival counter per hand-off slot, 64 B apart
        SPRS_HIP_TRY(ctx, hipMemsetAsync(fin_counter, 0, sizeof(unsigned int) * 64, ctx->stream));
        SPRS_HIP_TRY(ctx, hipMalloc((void **)&xext, sizeof(T) * stride));
    }
    SPRS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SPRS_OK;
}

template <class T>
void KrylovBase<T>::destroy() {
    if (work) (void)hipFree(work);
    if (rhs_buf) (void)hipFree(rhs_buf);
    if (x_buf) (void)hipFree(x_buf);
    if (part) (void)hipFree(part);
    if (partD) (void)hipFree(partD);
    if (red) (void)hipFree(red);
    if (fin_counter) (void)hipFree(fin_counter);
    if (xext) (void)hipFree(xext);
    red = nullptr; xext = nullptr; fin_counter = nullptr;
    for (auto e : ev) (void)hipEventDestroy(e);
    ev.clear();
    work = rhs_buf = x_buf = part = nullptr; partD = nullptr;
}

template <class T>
int KrylovBase<T>::ew_grid() const {
    constexpr int PKW = pack_width<T>::value;
    int64_t workb = ((int64_t)n / PKW + BLOCK - 1) / BLOCK;
    return balanced_grid(ctx, workb);
}

// Hand-offs of the distributed case.  The producing launch's last-arriving workgroup has already reduced the partials
// into `red` (struct Fin / finalize_last_block): all that is left per hand-off is ONE stream operation, the all-reduce — or
// NONE where the communicator has peer-to-peer mailboxes (knob "p2p_allreduce"): that workgroup also posts the
// values into every rank's mailbox and the consumer kernels of all ranks sum the `world` entries in rank order (device.hpp).
template <class T>
Fin KrylovBase<T>::fin_for(int slot, const void *base0, const void *base1, int P) const {
    if (!A->dist) return Fin{};
    Fin f;
    f.counter = fin_counter + 16 * slot;
    f.base0 = base0; f.base1 = base1;
    f.out0 = red + 2 * slot; f.out1 = red + 2 * slot + 2;
    f.P = P;
    if (use_p2p()) {
        // one more hand-off on this slot: its tag and the half of the slot it uses.  A fast rank can be at most one hand-off of
        // a slot ahead of a slow one (to post hand-off h + 2 it must have consumed h + 1, which the slow rank posts only after
        // all its workgroups consumed h), so two halves suffice.
        sprs_comm *cm = A->dist->comm;
        const unsigned int h = ++cm->seq[slot];
        f.box = cm->d_box; f.tag = h; f.mb_off = (unsigned int)mb_offset(slot, (int)(h & 1u));
    }
    return f;
}
template <class T>
const void *KrylovBase<T>::mbox_entries(int slot) const {
    const sprs_comm *cm = A->dist->comm;
    return reinterpret_cast<const char *>(cm->mbox) + mb_offset(slot, (int)(cm->seq[slot] & 1u));
}
template <class T>
int KrylovBase<T>::red1(const T *a, int P, int slot, PartT *oa) {
    if (!A->dist) { *oa = PartT{a, P}; return SPRS_OK; }
    if (use_p2p()) { *oa = PartT{reinterpret_cast<const T *>(mbox_entries(slot)), comm()->world, comm()->seq[slot]}; return SPRS_OK; }
    double *ra = red + 2 * slot;
    SPRS_TRY(allreduce_sum(comm(), ra, 16 / sizeof(Real<T>), sizeof(Real<T>) == 4));
    *oa = PartT{reinterpret_cast<const T *>(ra), 1};
    return SPRS_OK;
}
template <class T>
int KrylovBase<T>::red2(const T *a, const T *b, int P, int slot, PartT *oa, PartT *ob) {
    if (!A->dist) { *oa = PartT{a, P}; *ob = PartT{b, P}; return SPRS_OK; }
    if (use_p2p()) {
        *oa = PartT{reinterpret_cast<const T *>(mbox_entries(slot)), comm()->world, comm()->seq[slot]};
        *ob = *oa;
        return SPRS_OK;
    }
    double *ra = red + 2 * slot, *rb = red + 2 * slot + 2;
    SPRS_TRY(allreduce_sum(comm(), ra, 32 / sizeof(Real<T>), sizeof(Real<T>) == 4));
    *oa = PartT{reinterpret_cast<const T *>(ra), 1};
    *ob = PartT{reinterpret_cast<const T *>(rb), 1};
    return SPRS_OK;
}
template <class T>
int KrylovBase<T>::redD1(const Real<T> *a, int P, int slot, PartD *oa) {
    if (!A->dist) { *oa = PartD{a, P}; return SPRS_OK; }
    if (use_p2p()) { *oa = PartD{reinterpret_cast<const Real<T> *>(mbox_entries(slot)), comm()->world, comm()->seq[slot]}; return SPRS_OK; }
    double *ra = red + 2 * slot;
    SPRS_TRY(allreduce_sum(comm(), ra, 16 / sizeof(Real<T>), sizeof(Real<T>) == 4));
    *oa = PartD{reinterpret_cast<const Real<T> *>(ra), 1};
    return SPRS_OK;
}
template <class T>
int KrylovBase<T>::redDT(const Real<T> *a, const T *b, int P, int slot, PartD *oa, PartT *ob) {
    if (!A->dist) { *oa = PartD{a, P}; *ob = PartT{b, P}; return SPRS_OK; }
    if (use_p2p()) {
        *oa = PartD{reinterpret_cast<const Real<T> *>(mbox_entries(slot)), comm()->world, comm()->seq[slot]};
        *ob = PartT{reinterpret_cast<const T *>(mbox_entries(slot)), comm()->world, comm()->seq[slot]};
        return SPRS_OK;
    }
    double *ra = red + 2 * slot, *rb = red + 2 * slot + 2;
    SPRS_TRY(allreduce_sum(comm(), ra, 32 / sizeof(Real<T>), sizeof(Real<T>) == 4));
    *oa = PartD{reinterpret_cast<const Real<T> *>(ra), 1};
    *ob = PartT{reinterpret_cast<const T *>(rb), 1};
    return SPRS_OK;
}

template <class T>
int KrylovBase<T>::spmv(const T *x, T *y, int dot, const T *u, T *p0, T *p1, const int *status, bool conj_x, const Fin *fin) {
    if (A->dist) {
        // the SpMV input needs its halo tail filled: work vectors have room for it, a caller's
        // vector (initial residual, restart) is staged through `xext`
        T *xe = const_cast<T *>(x);
        const bool is_work = x >= work && x < work + stride * (size_t)nvec;
        if (!is_work) {
            SPRS_HIP_TRY(ctx, hipMemcpyAsync(xext, x, sizeof(T) * n, hipMemcpyDeviceToDevice, ctx->stream));
            xe = xext;
        }
        x = xe;
    }
    const T *x_caller = x;
    const int st = profiled([&]() -> int {
        if (A->dist) return dist_spmv<T>(A, const_cast<T *>(x), y, dot, u, p0, p1, status, conj_x, fin);
        return launch_spmv<T>(A, x, y, dot, u, p0, p1, status, conj_x, fin);
    }, !A->dist);
    if (profile && dot != 0 && u != x_caller) mark_step(1);
    return st;
}

template <class T>
template <class F>
int KrylovBase<T>::profiled(F &&run, bool one_kernel) {
    if (!profile) return run();
    // The events are not free: a launch that carries them costs ~6 us more (completion signal + timestamps; measured on the
    // 30-50 us iterations of cfg 2 / 3 / 4, and 11-13 us per cfg-5 iteration = 1 %).  profile = k >= 2 brackets one PAIR of
    // consecutive SpMV-class steps in k (a pair: BiCGStab's K2 and K4 are sampled equally often).
    const size_t call = prof_calls++;
    last_pair = -1;
    if (profile >= 2 && (call >> 1) % (size_t)profile != 0) return run();
    if (ev_used + 2 > ev.size()) {
        for (int k = 0; k < 2; ++k) {
            hipEvent_t e;
            SPRS_HIP_TRY(ctx, hipEventCreate(&e));
            ev.push_back(e);
        }
    }
    int st;
    if (one_kernel) {
        // one kernel per SpMV: the launch records its own begin / end (what rocprofv3 reports as the kernel's duration)
        ctx->prof_start = ev[ev_used]; ctx->prof_stop = ev[ev_used + 1];
        st = run();
        ctx->prof_start = nullptr; ctx->prof_stop = nullptr;
    } else {
        // exchange + two launches: bracket the whole thing (includes the wait for the halo)
        SPRS_HIP_TRY(ctx, hipEventRecord(ev[ev_used], ctx->stream));
        st = run();
        SPRS_HIP_TRY(ctx, hipEventRecord(ev[ev_used + 1], ctx->stream));
    }
    ev_noop.resize(ev_used / 2 + 1, 0);
    ev_noop[ev_used / 2] = 0;
    ev_call.resize(ev_used / 2 + 1, 0);
    ev_call[ev_used / 2] = call;
    ev_kind.resize(ev_used / 2 + 1, 0);
    ev_kind[ev_used / 2] = 0;
    last_pair = (long)(ev_used / 2);
    ev_used += 2;
    return st;
}

template <class T>
void KrylovBase<T>::profile_discard_last(size_t launches) {
    if (!profile) return;
    // the last `launches` STEPS were no-ops: the event pairs among them (all of them, or the sampled ones)
    const size_t first_noop = prof_calls > launches ? prof_calls - launches : 0;
    for (size_t k = ev_used / 2; k > 1 && ev_call[k - 1] >= first_noop; --k) ev_noop[k - 1] = 1;     // pair 0 brackets the solve
}

template <class T>
int KrylovBase<T>::begin_solve() {
    SPRS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    trace_rows = 0;
    stats = SolverStats();
    if (profile) {
        if (ev.size() < 2) {
            for (int k = 0; k < 2; ++k) {
                hipEvent_t e;
                SPRS_HIP_TRY(ctx, hipEventCreate(&e));
                ev.push_back(e);
            }
        }
        ev_used = 2;  // ev[0], ev[1] bracket the whole solve
        prof_calls = 0;
        ev_call.assign(1, 0);
        ev_kind.assign(1, 0);
        last_pair = -1;
        SPRS_HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
    }
    return SPRS_OK;
}

template <class T>
int KrylovBase<T>::end_solve() {
    if (!profile) return SPRS_OK;
    SPRS_HIP_TRY(ctx, hipEventRecord(ev[1], ctx->stream));
    SPRS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    SPRS_HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[0], ev[1]));
    stats.solve_ms = ms;
    for (size_t k = 2; k + 1 < ev_used; k += 2) {
        if (k / 2 < ev_noop.size() && ev_noop[k / 2]) continue;
        SPRS_HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
        stats.spmv_ms += ms;
        stats.spmv_launches += 1;
        const unsigned char kind = k / 2 < ev_kind.size() ? ev_kind[k / 2] : 0;
        stats.timed_dot_other += (kind & 1) != 0;
        stats.timed_fused_k2 += (kind & 2) != 0;
        stats.timed_fused_k4 += (kind & 4) != 0;
    }
    stats.steps = (int64_t)prof_calls;
    return SPRS_OK;
}

template <class T>
void KrylovBase<T>::trace_row(double a0, double a1, T b, T c, T d) {
    if (!trace || trace_rows >= trace_cap) return;
    double *t = trace + 8 * trace_rows;
    t[0] = a0; t[1] = a1;
    t[2] = sre(b); t[3] = sim(b); t[4] = sre(c); t[5] = sim(c); t[6] = sre(d); t[7] = sim(d);
    ++trace_rows;
}

// small helpers on the context stream
template <class T>
static int dcopy(sprs_ctx *c, T *dst, const T *src, size_t n) {
    SPRS_HIP_TRY(c, hipMemcpyAsync(dst, src, sizeof(T) * n, hipMemcpyDeviceToDevice, c->stream));
    return SPRS_OK;
}
template <class T>
static int dzero(sprs_ctx *c, T *dst, size_t n) {
    SPRS_HIP_TRY(c, hipMemsetAsync(dst, 0, sizeof(T) * n, c->stream));
    return SPRS_OK;
}

// ======================================================================= BiCGStab host
template <class T>
int BicgStab<T>::create(const sprs_csr *A, size_t size) {
    SPRS_TRY(this->init(A, size, 7));   // bicg_stab.rs:28 workspace 7n
    SPRS_HIP_TRY(this->ctx, hipMalloc((void **)&d_state, sizeof(BicgState<T>)));
    SPRS_HIP_TRY(this->ctx, hipHostMalloc((void **)&h_state, sizeof(BicgState<T>), hipHostMallocDefault));
    return SPRS_OK;
}
template <class T>
void BicgStab<T>::destroy() {
    if (d_state) (void)hipFree(d_state);
    if (h_state) (void)hipHostFree(h_state);
    d_state = h_state = nullptr;
    KrylovBase<T>::destroy();
}

template <class T>
template <class V>
int BicgStab<T>::run(const V *dinv, const T *rhs, T *x, size_t max_iter, Real<T> tol, size_t *its_out, Real<T> *res_out) {
    sprs_ctx *c = this->ctx;
    const size_t n = this->n;
    const bool pc = dinv != nullptr;
    *its_out = 0; *res_out = 0.0;

    Real<T> rhs_norm = 0.0;
    SPRS_TRY(this->norm2(rhs, &rhs_norm));                          // :55
    if (rhs_norm <= seps<Real<T>>()) {                                          // :56-60
        SPRS_TRY(dzero(c, x, n));
        *its_out = 0; *res_out = rhs_norm;
        return SPRS_OK;
    }
    const Real<T> tol2 = tol * rhs_norm;                             // :61

    // :64-69 / :234-241
    T *r = this->vec(0), *r0 = this->vec(1), *y = this->vec(2);
    T *p = pc ? this->vec(3) : y;
    T *v = pc ? this->vec(4) : this->vec(3);
    T *t = pc ? this->vec(5) : this->vec(4);
    T *z = pc ? this->vec(6) : nullptr;
    // ---- fused SpMV input (f64, no preconditioner, one GPU, plane-streaming chains; knob "spmv_fuse"): the two vector updates
    // whose results are SpMV inputs are formed INSIDE those SpMVs (spmv_chain.hip, FUSE) — K3 (r -= alpha v, :172) in K4, K1
    // (p = (v (-beta w) + p beta) + r, :155-156) in K2 — with K3's / K1's own prologues (bicg_fuse.hpp) and rounding sequence, so
    // every scalar, every element and every dot partial is bit-identical to the five-launch iteration; an iteration is three
    // launches.  K4 does not even store s: K5 forms it again from r and v (BicgK5<SV>) and writes r' = s - w t over r.  A tile reads
    // its operands' windows while other tiles are still reading them, so K2 writes p' to ANOTHER buffer: p alternates with a work
    // vector the unpreconditioned solve leaves unused (:28 allocates seven), and v and t swap roles every iteration (t is dead when K2
    // writes v', v when K4 writes t).
    bool fuse = false;
    if constexpr (std::is_same<T, double>::value && std::is_same<V, double>::value)
        fuse = !pc && !this->A->dist && c->spmv_fuse != 0 && chain_plan_used(this->A);
    T *palt = fuse ? this->vec(5) : nullptr;
    int pend_k1 = -1, pend_k3 = -1;          // fused: the mode / breakdown flag of the update that the next SpMV forms
    bool s_pending = false;                  // fused: K4 formed s without storing it (K5 forms it again)

    SPRS_TRY(this->spmv(x, r, 0, nullptr, nullptr, nullptr, nullptr));      // :73
    SPRS_TRY((launch_axpy<T, T>(c, n, sneg(sone<T>()), rhs, r)));           // :75
    SPRS_TRY(dcopy(c, r0, r, n));                                           // :78
    Real<T> r0_norm = 0.0;
    SPRS_TRY(this->norm2(r0, &r0_norm));                                    // :80
    if (r0_norm <= tol2) {                                                  // :81-83
        *its_out = 0; *res_out = r0_norm / rhs_norm;
        return SPRS_OK;
    }
    Real<T> r0_norm_tol = r0_norm * seps<Real<T>>();                                     // :84
    r0_norm_tol = r0_norm_tol * r0_norm_tol;                                // :85

    BicgState<T> &H = *h_state;
    memset(&H, 0, sizeof(H));
    H.rho = sfromr<T>(r0_norm * r0_norm);                                   // :88
    H.rho_old = H.rho;
    H.r_norm = r0_norm; H.r0_norm_tol = r0_norm_tol; H.tol2 = tol2;
    H.its = 0; H.status = ST_RUNNING;
    SPRS_HIP_TRY(c, hipMemcpyAsync(d_state, &H, sizeof(H), hipMemcpyHostToDevice, c->stream));
    const int *d_status = &d_state->status;

    const int G = this->ew_grid();
    const int cw = fused_chunked(this->A) ? 1 : 0;      // XCD-chunked walk of the vector kernels (spmv.hip)
    const int GS = spmv_num_partials(this->A);
    Real<T> *partN = this->dslot(0);
    T *partRho = this->pslot(0), *partB = this->pslot(1), *partTT = this->pslot(2), *partTR = this->pslot(3);

    typename KrylovBase<T>::PartT qB{partB, GS}, qTT{partTT, GS}, qTR{partTR, GS}, qRho{partRho, G};
    typename KrylovBase<T>::PartD qN{partN, G};
    auto K2 = [&]() -> int {                                                                 // :93/:160  v = A y ; r0.v
        const Fin f = this->fin_for(0, partB, nullptr, GS);
        if constexpr (std::is_same<T, double>::value && std::is_same<V, double>::value) {
            if (fuse && pend_k1 >= 0) {
                const BicgK1<double, double, false> k1{d_state, qN.p, qRho.p, qN.P, pend_k1, v, r, p, nullptr, y, 0.0, 0.0};
                pend_k1 = -1;
                SPRS_TRY(this->profiled([&]() -> int { return launch_chain_k2f(this->A, GS, k1, v, p, r, palt, t, r0, partB, d_status); }, true));
                this->stats.fused_k2 += 1;
                this->mark_step(2 | 1);      // (its dot operand is r0)
                std::swap(p, palt); y = p;   // p' lives in the other buffer
                std::swap(v, t);             // v' was written where t was
                return this->red1(partB, GS, 0, &qB);
            }
        }
        SPRS_TRY(this->spmv(y, v, 1, r0, partB, nullptr, d_status, false, &f));
        return this->red1(partB, GS, 0, &qB);
    };
    auto K3 = [&](int check) -> int {
        if (fuse) { pend_k3 = check; return (int)SPRS_OK; }     // formed by the next K4
        if (pc) return launch_fused<T>(c, n, G, cw, BicgK3<T, V, true>{d_state, qB.p, qB.P, check, v, r, dinv, z, T(), qB.tag, this->mb_timeout()});
        return launch_fused<T>(c, n, G, cw, BicgK3<T, V, false>{d_state, qB.p, qB.P, check, v, r, dinv, z, T(), qB.tag, this->mb_timeout()});
    };
    auto K4 = [&]() -> int {                                                                 // :104/:175 t = A s ; t.t, t.r
        const Fin f = this->fin_for(1, partTT, partTR, GS);
        if constexpr (std::is_same<T, double>::value && std::is_same<V, double>::value) {
            if (fuse && pend_k3 >= 0) {
                const BicgK3<double, double, false> k3{d_state, qB.p, qB.P, pend_k3, v, r, nullptr, nullptr, 0.0};
                pend_k3 = -1;
                SPRS_TRY(this->profiled([&]() -> int { return launch_chain_k4f(this->A, GS, k3, r, v, nullptr, t, partTT, partTR, d_status); }, true));
                this->stats.fused_k4 += 1;
                this->mark_step(4);
                s_pending = true;            // s was formed on the fly and not stored: K5 forms it again from r and v
                return this->red2(partTT, partTR, GS, 1, &qTT, &qTR);
            }
        }
        SPRS_TRY(this->spmv(pc ? z : r, t, 2, r, partTT, partTR, d_status, false, &f));
        return this->red2(partTT, partTR, GS, 1, &qTT, &qTR);
    };
    auto K5 = [&]() -> int {
        const Fin f = this->fin_for(3, partN, partRho, G);
        if (s_pending) {
            s_pending = false;
            SPRS_TRY(launch_fused<T>(c, n, G, cw, BicgK5<T, false, true>{d_state, qTT.p, qTR.p, qTT.P, y, z, t, r0, x, r, partN, partRho, f, T(), T(), T(), 0.0, T(), qTT.tag, this->mb_timeout(), v}));
            return this->redDT(partN, partRho, G, 3, &qN, &qRho);
        }
        if (pc) SPRS_TRY(launch_fused<T>(c, n, G, cw, BicgK5<T, true>{d_state, qTT.p, qTR.p, qTT.P, y, z, t, r0, x, r, partN, partRho, f, T(), T(), T(), 0.0, T(), qTT.tag, this->mb_timeout()}));
        else SPRS_TRY(launch_fused<T>(c, n, G, cw, BicgK5<T, false>{d_state, qTT.p, qTR.p, qTT.P, y, z, t, r0, x, r, partN, partRho, f, T(), T(), T(), 0.0, T(), qTT.tag, this->mb_timeout()}));
        return this->redDT(partN, partRho, G, 3, &qN, &qRho);
    };
    auto K1 = [&](int mode) -> int {
        if (fuse) { pend_k1 = mode; return (int)SPRS_OK; }      // formed by the next K2
        if (pc) return launch_fused<T>(c, n, G, cw, BicgK1<T, V, true>{d_state, qN.p, qRho.p, qN.P, mode, v, r, p, dinv, y, T(), T(), qN.tag, this->mb_timeout()});
        return launch_fused<T>(c, n, G, cw, BicgK1<T, V, false>{d_state, qN.p, qRho.p, qN.P, mode, v, r, p, dinv, y, T(), T(), qN.tag, this->mb_timeout()});
    };
    auto fetch = [&]() -> int {
        SPRS_HIP_TRY(c, hipMemcpyAsync(&H, d_state, sizeof(H), hipMemcpyDeviceToHost, c->stream));
        SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        return (int)SPRS_OK;
    };

    // ---- unrolled first iteration (:87-120 / :258-293)
    if (pc) {
        SPRS_TRY(dcopy(c, p, r, n));                                        // :261
        SPRS_TRY((launch_diag_apply<T, V>(c, n, dinv, p, y)));              // :262
    } else {
        SPRS_TRY(dcopy(c, y, r, n));                                        // :91
    }
    SPRS_TRY(K2()); SPRS_TRY(K3(0)); SPRS_TRY(K4()); SPRS_TRY(K5());
    const bool tracing = this->trace != nullptr;
    if (tracing) {
        SPRS_TRY(fetch());
        this->trace_row(0.0, r0_norm, H.rho, H.alpha, H.w);
    }

    // ---- main loop (:122-197)
    const size_t poll = tracing ? 1 : (size_t)(c->poll < 1 ? 1 : c->poll);
    size_t its = 1, since_poll = 0;
    int resume_mode = 0;
    while (true) {
        const bool done_enqueue = its >= max_iter;
        if (!done_enqueue) {
            SPRS_TRY(K1(resume_mode)); resume_mode = 0;
            SPRS_TRY(K2()); SPRS_TRY(K3(1)); SPRS_TRY(K4()); SPRS_TRY(K5());
            ++its; ++since_poll;
        }
        if (done_enqueue || since_poll >= poll) {
            since_poll = 0;
            SPRS_TRY(fetch());
            if (H.status == ST_CONVERGED) {                                 // :124-126
                *its_out = (size_t)H.its; *res_out = H.r_norm / rhs_norm;
                return SPRS_OK;
            }
            if (H.status == ST_BREAKDOWN) {                                 // :164-167
                *its_out = (size_t)H.its;
                return SPRS_BREAKDOWN;
            }
            if (H.status == ST_COMM_TIMEOUT) {
                snprintf(c->err, sizeof(c->err), "a peer's hand-off did not reach this rank's mailbox within %d ms (p2p_timeout_ms)", c->p2p_timeout_ms);
                return SPRS_ERR_RCCL;
            }
            if (H.status == ST_RESTART) {                                   // :131-145, executed at iteration H.its
                // K2 / K4 of the iterations enqueued from the requesting one on returned at their first instruction
                this->profile_discard_last(2 * (its - (size_t)H.its));
                if (fuse && ((its - (size_t)H.its) & 1)) {
                    // ... but the host rotated the buffers once for each of them as it enqueued: an odd number of idle iterations
                    // leaves every pair of names exchanged against what the last EXECUTED launches wrote
                    std::swap(p, palt); y = p; std::swap(v, t);
                }
                SPRS_TRY(this->spmv(x, r, 0, nullptr, nullptr, nullptr, nullptr));  // :134
                SPRS_TRY((launch_axpy<T, T>(c, n, sneg(sone<T>()), rhs, r)));       // :137
                SPRS_TRY(dcopy(c, r0, r, n));                                       // :140
                Real<T> rn = 0.0;
                SPRS_TRY(this->norm2(r, &rn));                                      // :142
                H.rho = sfromr<T>(rn * rn);                                         // :143
                H.r0_norm_tol = sre(H.rho) * seps<Real<T>>() * seps<Real<T>>();                             // :144
                H.status = ST_RUNNING;
                SPRS_HIP_TRY(c, hipMemcpyAsync(d_state, &H, sizeof(H), hipMemcpyHostToDevice, c->stream));
                its = (size_t)H.its;       // every kernel after the request was a no-op: redo from here
                resume_mode = 1;
                continue;
            }
            if (tracing && !done_enqueue) this->trace_row((double)(H.its - 1), H.r_norm, H.rho, H.alpha, H.w);
            if (done_enqueue) break;
        }
    }
    *its_out = max_iter;                                                    // :199
    return SPRS_INSUFFICIENT_ITER;
}

// literal mode: the reference's op list, one kernel per op, host-consumed scalars
template <class T>
template <class V>
int BicgStab<T>::run_literal(const V *dinv, const T *rhs, T *x, size_t max_iter, Real<T> tol, size_t *its_out,
                             Real<T> *res_out) {
    sprs_ctx *c = this->ctx;
    const size_t n = this->n;
    const bool pc = dinv != nullptr;
    *its_out = 0; *res_out = 0.0;
    Real<T> rhs_norm = 0.0;
    SPRS_TRY(this->norm2(rhs, &rhs_norm));
    if (rhs_norm <= seps<Real<T>>()) { SPRS_TRY(dzero(c, x, n)); *res_out = rhs_norm; return SPRS_OK; }
    const Real<T> tol2 = tol * rhs_norm;
    T *r = this->vec(0), *r0 = this->vec(1), *y = this->vec(2);
    T *p = pc ? this->vec(3) : y;
    T *v = pc ? this->vec(4) : this->vec(3);
    T *t = pc ? this->vec(5) : this->vec(4);
    T *z = pc ? this->vec(6) : nullptr;
    const T *sz = pc ? z : r;
    auto mv = [&](const T *in, T *out) { return this->spmv(in, out, 0, nullptr, nullptr, nullptr, nullptr); };
    auto cdot = [&](const T *a, const T *b, T *o) { return this->cdot(a, b, o); };
    auto axpy = [&](T a, const T *xx, T *yy) { return launch_axpy<T, T>(c, n, a, xx, yy); };

    SPRS_TRY(mv(x, r));
    SPRS_TRY(axpy(sneg(sone<T>()), rhs, r));
    SPRS_TRY(dcopy(c, r0, r, n));
    Real<T> r0_norm = 0.0;
    SPRS_TRY(this->norm2(r0, &r0_norm));
    if (r0_norm <= tol2) { *res_out = r0_norm / rhs_norm; return SPRS_OK; }
    Real<T> r0_norm_tol = r0_norm * seps<Real<T>>();
    r0_norm_tol = r0_norm_tol * r0_norm_tol;
    T rho = sfromr<T>(r0_norm * r0_norm);
    if (pc) { SPRS_TRY(dcopy(c, p, r, n)); SPRS_TRY((launch_diag_apply<T, V>(c, n, dinv, p, y))); }
    else SPRS_TRY(dcopy(c, y, r, n));
    SPRS_TRY(mv(y, v));
    T tmp;
    SPRS_TRY(cdot(r0, v, &tmp));
    T alpha = sdiv(rho, tmp);
    SPRS_TRY(axpy(sneg(alpha), v, r));
    if (pc) SPRS_TRY((launch_diag_apply<T, V>(c, n, dinv, r, z)));
    SPRS_TRY(mv(sz, t));
    SPRS_TRY(cdot(t, t, &tmp));
    T w = szero<T>();
    if (sre(tmp) > 0.0) { T tr; SPRS_TRY(cdot(t, r, &tr)); w = sdiv(tr, tmp); }
    SPRS_TRY(axpy(sneg(alpha), y, x));
    SPRS_TRY(axpy(sneg(w), sz, x));
    SPRS_TRY(axpy(sneg(w), t, r));
    this->trace_row(0.0, r0_norm, rho, alpha, w);
    for (size_t its = 1; its < max_iter; ++its) {
        Real<T> r_norm = 0.0;
        SPRS_TRY(this->norm2(r, &r_norm));
        if (r_norm <= tol2) { *its_out = its; *res_out = r_norm / rhs_norm; return SPRS_OK; }
        const T rho_old = rho;
        SPRS_TRY(cdot(r0, r, &rho));
        if (sabs(rho) < r0_norm_tol) {
            SPRS_TRY(mv(x, r));
            SPRS_TRY(axpy(sneg(sone<T>()), rhs, r));
            SPRS_TRY(dcopy(c, r0, r, n));
            Real<T> rn = 0.0;
            SPRS_TRY(this->norm2(r, &rn));
            rho = sfromr<T>(rn * rn);
            r0_norm_tol = sre(rho) * seps<Real<T>>() * seps<Real<T>>();
        }
        const T beta = smul(sdiv(rho, rho_old), sdiv(alpha, w));
        SPRS_TRY(launch_axpby<T>(c, n, smul(sneg(beta), w), v, beta, p));
        SPRS_TRY(axpy(sone<T>(), r, p));
        if (pc) SPRS_TRY((launch_diag_apply<T, V>(c, n, dinv, p, y)));
        SPRS_TRY(mv(y, v));
        SPRS_TRY(cdot(r0, v, &tmp));
        if (sabs(tmp) <= 0.0) { *its_out = its; return SPRS_BREAKDOWN; }
        alpha = sdiv(rho, tmp);
        SPRS_TRY(axpy(sneg(alpha), v, r));
        if (pc) SPRS_TRY((launch_diag_apply<T, V>(c, n, dinv, r, z)));
        SPRS_TRY(mv(sz, t));
        SPRS_TRY(cdot(t, t, &tmp));
        if (sre(tmp) > 0.0) { T tr; SPRS_TRY(cdot(t, r, &tr)); w = sdiv(tr, tmp); }
        else w = szero<T>();
        SPRS_TRY(axpy(sneg(alpha), y, x));
        SPRS_TRY(axpy(sneg(w), sz, x));
        SPRS_TRY(axpy(sneg(w), t, r));
        this->trace_row((double)its, r_norm, rho, alpha, w);
    }
    *its_out = max_iter;
    return SPRS_INSUFFICIENT_ITER;
}

template <class T>
static int check_diag(const sprs_diag *P, size_t n) {
    if (!P) return SPRS_OK;
    if (P->n != n) return SPRS_DIM_MISMATCH;
    if (P->t_dtype != dtype_of<T>::value) return SPRS_INVALID_ARGUMENT;
    return SPRS_OK;
}

template <class T>
int BicgStab<T>::solve_dev(const sprs_diag *P, const T *rhs, size_t rhs_len, T *x, size_t x_len, size_t max_iter,
                           Real<T> tol, size_t *its_out, Real<T> *res_out) {
    size_t its_dummy; Real<T> res_dummy;
    if (!its_out) its_out = &its_dummy;
    if (!res_out) res_out = &res_dummy;
    if (rhs_len != this->n) return SPRS_INCOMPATIBLE_RHS_SIZE;             // :44-48
    if (x_len != this->n) return SPRS_INCOMPATIBLE_X_SIZE;                 // :49-53
    SPRS_TRY(check_diag<T>(P, this->n));
    SPRS_TRY(this->begin_solve());
    int st;
    const bool lit = this->mode == 1;
    if (P && P->v_complex) {
        if constexpr (is_complex<T>::value) {
            const T *d = (const T *)P->dinv;
            st = lit ? run_literal<T>(d, rhs, x, max_iter, tol, its_out, res_out)
                     : run<T>(d, rhs, x, max_iter, tol, its_out, res_out);
        } else {
            return SPRS_INVALID_ARGUMENT;
        }
    } else {
        const Real<T> *d = P ? (const Real<T> *)P->dinv : nullptr;
        st = lit ? run_literal<Real<T>>(d, rhs, x, max_iter, tol, its_out, res_out)
                 : run<Real<T>>(d, rhs, x, max_iter, tol, its_out, res_out);
    }
    if (st >= SPRS_ERR_HIP) return st;
    SPRS_TRY(this->end_solve());
    return st;
}

// ======================================================================= MINRES / CSMINRES host
template <class T>
int MinRes<T>::create(const sprs_csr *A, size_t size, bool saunders_) {
    saunders = saunders_;
    SPRS_TRY(this->init(A, size, 8));   // minres.rs:24 workspace 8n (cs_minres.rs:22 uses 7n)
    SPRS_HIP_TRY(this->ctx, hipMalloc((void **)&d_state, sizeof(MinresDev<T>)));
    SPRS_HIP_TRY(this->ctx, hipHostMalloc((void **)&h_state, sizeof(MinresDev<T>), hipHostMallocDefault));
    return SPRS_OK;
}
template <class T>
void MinRes<T>::destroy() {
    if (d_state) (void)hipFree(d_state);
    if (h_state) (void)hipHostFree(h_state);
    d_state = h_state = nullptr;
    KrylovBase<T>::destroy();
}

template <class T>
template <class V>
int MinRes<T>::run(const V *dinv, const T *rhs, T *x, size_t max_iter, Real<T> tol, size_t *its_out, Real<T> *res_out) {
    sprs_ctx *c = this->ctx;
    const size_t n = this->n;
    const bool pc = dinv != nullptr;
    const bool sau = saunders && is_complex<T>::value;   // conj() is the identity on real data
    *its_out = 0; *res_out = 0.0;

    Real<T> rhs_norm = 0.0;
    SPRS_TRY(this->norm2(rhs, &rhs_norm));                          // :51
    if (rhs_norm <= seps<Real<T>>()) { SPRS_TRY(dzero(c, x, n)); *res_out = rhs_norm; return SPRS_OK; }   // :52-56
    const Real<T> threshold = tol * rhs_norm;                                // :57

    T *v_old = this->vec(0), *v_new = this->vec(1), *v = this->vec(2);      // :68-70
    T *p_old = this->vec(3), *p_oold = this->vec(4), *p = this->vec(5);     // :71-73
    T *w = this->vec(6), *w_new = this->vec(7);                             // :222-223

    SPRS_TRY(dcopy(c, v_new, rhs, n));                                      // :77
    SPRS_TRY(this->spmv(x, v_old, 0, nullptr, nullptr, nullptr, nullptr));  // :78
    SPRS_TRY((launch_axpy<T, T>(c, n, sneg(sone<T>()), v_old, v_new)));     // :80
    Real<T> res_norm = 0.0;
    SPRS_TRY(this->norm2(v_new, &res_norm));                        // :81
    Real<T> beta_new;
    if (pc) {
        SPRS_TRY((launch_diag_apply<T, V>(c, n, dinv, v_new, w_new)));      // :233
        T b2;
        SPRS_TRY(this->cdot(v_new, w_new, &b2));               // :235
        if (sre(b2) < seps<Real<T>>() || sim(b2) > seps<Real<T>>() * sre(b2)) {                     // :236-244
            *its_out = 0; *res_out = sre(b2);
            return SPRS_INVALID_PRECOND;
        }
        beta_new = ssqrt(sre(b2));                                           // :245
        const Real<T> ts = Real<T>(1) / beta_new;                                   // :248
        SPRS_TRY(launch_rscale<T>(c, n, ts, v_new));                        // :249
        SPRS_TRY(launch_rscale<T>(c, n, ts, w_new));                        // :250
    } else {
        beta_new = res_norm;                                                // :82
        SPRS_TRY(launch_rscale<T>(c, n, Real<T>(1) / beta_new, v_new));            // :84
    }
    SPRS_TRY(dzero(c, v, n)); SPRS_TRY(dzero(c, p_old, n)); SPRS_TRY(dzero(c, p, n));   // :86-88

    MinresDev<T> &H = *h_state;
    memset(&H, 0, sizeof(H));
    MinresState<T> &S0 = H.st[0];
    S0.c = sone<T>(); S0.c_old = sone<T>(); S0.eta = sone<T>(); S0.alpha = szero<T>();   // :60-64
    S0.s = 0.0; S0.s_old = 0.0;
    S0.beta = beta_new; S0.beta_one = beta_new;                             // :82-83
    S0.res_norm = res_norm; S0.threshold = threshold;
    H.st[1] = S0;
    H.its = 0; H.status = ST_RUNNING;
    SPRS_HIP_TRY(c, hipMemcpyAsync(d_state, &H, sizeof(H), hipMemcpyHostToDevice, c->stream));
    const int *d_status = &d_state->status;

    const int G = this->ew_grid();
    const int cw = fused_chunked(this->A) ? 1 : 0;      // XCD-chunked walk of the vector kernels (spmv.hip)
    const int GS = spmv_num_partials(this->A);
    T *partAlpha = this->pslot(0), *partBeta2 = this->pslot(1);

    auto fetch = [&]() -> int {
        SPRS_HIP_TRY(c, hipMemcpyAsync(&H, d_state, sizeof(H), hipMemcpyDeviceToHost, c->stream));
        SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        return (int)SPRS_OK;
    };
    const bool tracing = this->trace != nullptr;
    const size_t poll = tracing ? 1 : (size_t)(c->poll < 1 ? 1 : c->poll);
    size_t since_poll = 0;

    // ---- M3 deferred (no preconditioner, one GPU, the lane-per-row kernels of the compressed streams — cfg 3, cfg 4; knob
    // "spmv_fuse"; minres_fuse.hpp): M3 of iteration k — beta_new, the normalisation of v_new, the Givens rotation, p, x, the
    // convergence event — is not launched after M2.  Iteration k + 1 then runs TWO launches: the SpMV on the un-normalised v_new
    // (its prologue gets 1 / beta_new, its gathers multiply by it: spmv_dict_scaled_kernel) and MinresM23 = M3 (k) + M2 (k + 1) in
    // one pass (9 vector passes instead of 8 + 4).  The normalised v_new is never stored: the vector that is "v" of iteration
    // k + 1 and "v_old" of k + 2 stays raw in memory and every reader applies the factor (v_raw / vold_raw below).  Every scalar and
    // element is bit-identical to the three-launch iteration.  M3 is launched on its own where nothing follows it in time — on the
    // last iteration, before a poll of the status word (a convergence is seen as early as without the deferral), while tracing —
    // and then also writes back the normalised form of a raw v, so that the plain kernels find what they expect.
    bool m3_fusable = false;
    if constexpr (std::is_same<T, double>::value || std::is_same<T, cplx>::value)
        m3_fusable = !pc && !this->A->dist && spmv_scaled_available(this->A);
    Real<T> *pbeta[2] = {this->dslot(0), this->dslot(1)};     // |v_new|^2 partials: MinresM23 reads one array while it writes the other
    int cur_pb = 0;
    bool deferred = false, v_raw = false, vold_raw = false;
    const T *m3_p_old = nullptr, *m3_p_oold = nullptr; T *m3_p = nullptr;

    for (size_t its = 0;; ++its) {                                          // :90
        const bool done_enqueue = its >= max_iter;
        if (!done_enqueue) {
            const int par = (int)(its & 1);
            { T *tp = v_old; v_old = v; v = v_new; v_new = tp; }             // :92-96
            vold_raw = v_raw; v_raw = deferred;
            const bool will_defer = m3_fusable && !tracing && its + 1 < max_iter && since_poll + 1 < poll;     // M3 of THIS iteration
            if (pc) { T *tp = w; w = w_new; w_new = tp; }                    // :259,264-265
            const T *q = pc ? w : v;                                         // operand of A and source of p
            // M1: v_new = A q (CSMINRES: A conj(q)) ; alpha = conj(q).v_new   (:116 / :271 / cs:99-103)
            const Fin fA = this->fin_for(0, partAlpha, nullptr, GS);
            typename KrylovBase<T>::PartT qA, qB2{partBeta2, G};
            typename KrylovBase<T>::PartD qBt{pbeta[cur_pb], G};
            bool iteration_done = false;
            if constexpr (std::is_same<T, double>::value || std::is_same<T, cplx>::value) {
                if (deferred) {
                    // v is the raw v_new of iteration its - 1, v_old = v (its - 1) (raw too unless that iteration began after a flush)
                    deferred = false;
                    auto two_launches = [&](auto sau_tag) -> int {
                        constexpr bool SAU = decltype(sau_tag)::value;
                        MinresM3<T, false, SAU> m3{d_state, par ^ 1, (long long)its - 1, pbeta[cur_pb], partBeta2, G, nullptr, nullptr, v_old, m3_p_old, m3_p_oold, m3_p, x,
                                                   0.0, 0.0, 0.0, 0.0, T(), T(), T(), T()};
                        m3.q_raw = vold_raw ? 1 : 0;
                        SPRS_TRY(this->profiled([&]() -> int { return launch_spmv_scaled<T, SAU>(this->A, m3, v, v_new, partAlpha); }, true));
                        this->stats.fused_k2 += 1;                  // (counted with BiCGStab's fused K2: an SpMV launch that formed its input)
                        this->mark_step(2);
                        return launch_fused<T>(c, n, G, cw, MinresM23<T, SAU>{m3, partAlpha, GS, v, v_new, pbeta[cur_pb ^ 1], T(), T(), T(), 0.0});
                    };
                    int st;
                    if (sau) {
                        if constexpr (is_complex<T>::value) st = two_launches(std::true_type{});
                        else st = SPRS_INVALID_ARGUMENT;
                    } else st = two_launches(std::false_type{});
                    SPRS_TRY(st);
                    cur_pb ^= 1;
                    qBt = typename KrylovBase<T>::PartD{pbeta[cur_pb], G};
                    iteration_done = true;
                }
            }
            if (!iteration_done) {
                SPRS_TRY(this->spmv(q, v_new, 1, q, partAlpha, nullptr, d_status, sau, &fA));
                SPRS_TRY(this->red1(partAlpha, GS, 0, &qA));
                if (pc) {
                    SPRS_TRY(launch_fused<T>(c, n, G, cw, MinresM2<T, V, true>{d_state, par, qA.p, qA.P, v_old, v, v_new, dinv, w_new, pbeta[cur_pb], partBeta2, this->fin_for(1, partBeta2, nullptr, G), T(), T(), 0.0, T(), qA.tag, this->mb_timeout()}));
                    SPRS_TRY(this->red1(partBeta2, G, 1, &qB2));
                } else {
                    SPRS_TRY(launch_fused<T>(c, n, G, cw, MinresM2<T, V, false>{d_state, par, qA.p, qA.P, v_old, v, v_new, dinv, w_new, pbeta[cur_pb], partBeta2, this->fin_for(1, pbeta[cur_pb], nullptr, G), T(), T(), 0.0, T(), qA.tag, this->mb_timeout()}));
                    SPRS_TRY(this->redD1(pbeta[cur_pb], G, 1, &qBt));
                }
            }
            { T *tp = p_oold; p_oold = p_old; p_old = p; p = tp; }           // :151-154
            if (will_defer) {
                // done by the next iteration's two launches (above); the p names as they are now
                deferred = true;
                m3_p_old = p_old; m3_p_oold = p_oold; m3_p = p;
            } else {
                auto m3_alone = [&](auto pc_tag, auto sau_tag) -> int {
                    constexpr bool PCF = decltype(pc_tag)::value, SAF = decltype(sau_tag)::value;
                    MinresM3<T, PCF, SAF> m3{d_state, par, (long long)its, qBt.p, qB2.p, pc ? qB2.P : qBt.P, v_new, w_new, q, p_old, p_oold, p, x,
                                             0.0, 0.0, 0.0, 0.0, T(), T(), T(), T(), pc ? qB2.tag : qBt.tag, this->mb_timeout()};
                    if (!PCF && v_raw) { m3.q_raw = 1; m3.q_back = v; }     // a raw v: used scaled, and left normalised for the plain kernels
                    return launch_fused<T>(c, n, G, cw, m3);
                };
                if (pc) SPRS_TRY(m3_alone(std::true_type{}, std::false_type{}));
                else if (sau) SPRS_TRY(m3_alone(std::false_type{}, std::true_type{}));
                else SPRS_TRY(m3_alone(std::false_type{}, std::false_type{}));
                v_raw = false;
            }
            ++since_poll;
        }
        if (done_enqueue || since_poll >= poll) {
            since_poll = 0;
            SPRS_TRY(fetch());
            if ((H.status & 15) == ST_CONVERGED) {                          // :165-167 (0-based its; the word carries the iteration, MinresM3)
                *its_out = (size_t)H.its;
                *res_out = H.st[(H.its + 1) & 1].res_norm / rhs_norm;
                if (tracing) {
                    const MinresState<T> &N = H.st[(H.its + 1) & 1];
                    this->trace_row((double)H.its, N.beta, H.st[H.its & 1].alpha, N.c, sfromr<T>(N.s));
                    if (this->trace_rows) this->trace[8 * (this->trace_rows - 1) + 7] = N.res_norm;
                }
                return SPRS_OK;
            }
            if (H.status == ST_INVALID_PC) {                                // :279-287
                *its_out = (size_t)H.its; *res_out = H.st[H.its & 1].pc_re;
                return SPRS_INVALID_PRECOND;
            }
            if (H.status == ST_COMM_TIMEOUT) {
                snprintf(c->err, sizeof(c->err), "a peer's hand-off did not reach this rank's mailbox within %d ms (p2p_timeout_ms)", c->p2p_timeout_ms);
                return SPRS_ERR_RCCL;
            }
            if (done_enqueue) break;
            if (tracing) {
                const MinresState<T> &N = H.st[(its + 1) & 1];
                this->trace_row((double)its, N.beta, H.st[its & 1].alpha, N.c, sfromr<T>(N.s));
                if (this->trace_rows) this->trace[8 * (this->trace_rows - 1) + 7] = N.res_norm;
            }
        }
    }
    *its_out = max_iter;                                                    // :171
    return SPRS_INSUFFICIENT_ITER;
}

template <class T>
template <class V>
int MinRes<T>::run_literal(const V *dinv, const T *rhs, T *x, size_t max_iter, Real<T> tol, size_t *its_out,
                           Real<T> *res_out) {
    sprs_ctx *c = this->ctx;
    const size_t n = this->n;
    const bool pc = dinv != nullptr;
    const bool sau = saunders;
    *its_out = 0; *res_out = 0.0;
    Real<T> rhs_norm = 0.0;
    SPRS_TRY(this->norm2(rhs, &rhs_norm));
    if (rhs_norm <= seps<Real<T>>()) { SPRS_TRY(dzero(c, x, n)); *res_out = rhs_norm; return SPRS_OK; }
    const Real<T> threshold = tol * rhs_norm;
    T cc = sone<T>(), c_old = sone<T>(), eta = sone<T>();
    Real<T> s = 0.0, s_old = 0.0;
    T *v_old = this->vec(0), *v_new = this->vec(1), *v = this->vec(2);
    T *p_old = this->vec(3), *p_oold = this->vec(4), *p = this->vec(5);
    T *w = this->vec(6), *w_new = this->vec(7), *tvec = this->vec(6);
    auto mv = [&](const T *in, T *out) { return this->spmv(in, out, 0, nullptr, nullptr, nullptr, nullptr); };
    auto axpy = [&](T a, const T *xx, T *yy) { return launch_axpy<T, T>(c, n, a, xx, yy); };
    SPRS_TRY(dcopy(c, v_new, rhs, n));
    SPRS_TRY(mv(x, v_old));
    SPRS_TRY(axpy(sneg(sone<T>()), v_old, v_new));
    Real<T> res_norm = 0.0;
    SPRS_TRY(this->norm2(v_new, &res_norm));
    Real<T> beta_new, beta_one;
    if (pc) {
        SPRS_TRY((launch_diag_apply<T, V>(c, n, dinv, v_new, w_new)));
        T b2;
        SPRS_TRY(this->cdot(v_new, w_new, &b2));
        if (sre(b2) < seps<Real<T>>() || sim(b2) > seps<Real<T>>() * sre(b2)) { *res_out = sre(b2); return SPRS_INVALID_PRECOND; }
        beta_new = ssqrt(sre(b2)); beta_one = beta_new;
        SPRS_TRY(launch_rscale<T>(c, n, Real<T>(1) / beta_new, v_new));
        SPRS_TRY(launch_rscale<T>(c, n, Real<T>(1) / beta_new, w_new));
    } else {
        beta_new = res_norm; beta_one = beta_new;
        SPRS_TRY(launch_rscale<T>(c, n, Real<T>(1) / beta_new, v_new));
    }
    SPRS_TRY(dzero(c, v, n)); SPRS_TRY(dzero(c, p_old, n)); SPRS_TRY(dzero(c, p, n));
    for (size_t its = 0; its < max_iter; ++its) {
        const Real<T> beta = beta_new;
        { T *tp = v_old; v_old = v; v = v_new; v_new = tp; }
        T alpha;
        const T *q;
        if (pc) {
            { T *tp = w; w = w_new; w_new = tp; }
            SPRS_TRY(mv(w, v_new));
            SPRS_TRY(this->cdot(w, v_new, &alpha));
            q = w;
        } else if (sau) {
            SPRS_TRY(launch_conj<T>(c, n, v, tvec));
            SPRS_TRY(mv(tvec, v_new));
            SPRS_TRY(this->cdot(v, v_new, &alpha));
            q = tvec;
        } else {
            SPRS_TRY(mv(v, v_new));
            SPRS_TRY(this->cdot(v, v_new, &alpha));
            q = v;
        }
        SPRS_TRY(axpy(sfromr<T>(-beta), v_old, v_new));
        SPRS_TRY(axpy(sneg(alpha), v, v_new));
        if (pc) {
            SPRS_TRY((launch_diag_apply<T, V>(c, n, dinv, v_new, w_new)));
            T b2;
            SPRS_TRY(this->cdot(v_new, w_new, &b2));
            if (sre(b2) < seps<Real<T>>() || sim(b2) > seps<Real<T>>() * sre(b2)) { *its_out = its; *res_out = sre(b2); return SPRS_INVALID_PRECOND; }
            beta_new = ssqrt(sre(b2));
            SPRS_TRY(launch_rscale<T>(c, n, Real<T>(1) / beta_new, v_new));
            SPRS_TRY(launch_rscale<T>(c, n, Real<T>(1) / beta_new, w_new));
        } else {
            SPRS_TRY(this->norm2(v_new, &beta_new));
            SPRS_TRY(launch_rscale<T>(c, n, Real<T>(1) / beta_new, v_new));
        }
        const Real<T> r3 = s_old * beta;
        const T tr = smulr(sau ? sconj(c_old) : c_old, beta);
        const T r2 = sadd(smulr(alpha, s), smul(cc, tr));
        const T r1_hat = ssub(smul(sau ? sconj(cc) : cc, alpha), smulr(tr, s));
        const Real<T> r1_inv = Real<T>(1) / ssqrt(ssq(r1_hat) + beta_new * beta_new);
        c_old = cc; s_old = s;
        cc = smulr(sau ? sconj(r1_hat) : r1_hat, r1_inv);
        s = beta_new * r1_inv;
        { T *tp = p_oold; p_oold = p_old; p_old = p; p = tp; }
        SPRS_TRY(dcopy(c, p, q, n));
        SPRS_TRY(axpy(sneg(r2), p_old, p));
        SPRS_TRY(axpy(sfromr<T>(-r3), p_oold, p));
        SPRS_TRY(launch_rscale<T>(c, n, r1_inv, p));
        SPRS_TRY(axpy(smulr(smul(cc, eta), beta_one), p, x));
        res_norm *= sabs(s);
        this->trace_row((double)its, beta_new, alpha, cc, sfromr<T>(s));
        if (this->trace && this->trace_rows) this->trace[8 * (this->trace_rows - 1) + 7] = res_norm;
        if (res_norm < threshold) { *its_out = its; *res_out = res_norm / rhs_norm; return SPRS_OK; }
        eta = smulr(eta, -s);
    }
    *its_out = max_iter;
    return SPRS_INSUFFICIENT_ITER;
}

template <class T>
int MinRes<T>::solve_dev(const sprs_diag *P, const T *rhs, size_t rhs_len, T *x, size_t x_len, size_t max_iter,
                         Real<T> tol, size_t *its_out, Real<T> *res_out) {
    size_t its_dummy; Real<T> res_dummy;
    if (!its_out) its_out = &its_dummy;
    if (!res_out) res_out = &res_dummy;
    if (rhs_len != this->n) return SPRS_INCOMPATIBLE_RHS_SIZE;             // minres.rs:40-44
    if (x_len != this->n) return SPRS_INCOMPATIBLE_X_SIZE;                 // :45-49
    if (saunders && P) return SPRS_INVALID_ARGUMENT;                       // CSMinRes has no precond_solve
    SPRS_TRY(check_diag<T>(P, this->n));
    SPRS_TRY(this->begin_solve());
    int st;
    const bool lit = this->mode == 1;
    if (P && P->v_complex) {
        if constexpr (is_complex<T>::value) {
            const T *d = (const T *)P->dinv;
            st = lit ? run_literal<T>(d, rhs, x, max_iter, tol, its_out, res_out)
                     : run<T>(d, rhs, x, max_iter, tol, its_out, res_out);
        } else {
            return SPRS_INVALID_ARGUMENT;
        }
    } else {
        const Real<T> *d = P ? (const Real<T> *)P->dinv : nullptr;
        st = lit ? run_literal<Real<T>>(d, rhs, x, max_iter, tol, its_out, res_out)
                 : run<Real<T>>(d, rhs, x, max_iter, tol, its_out, res_out);
    }
    if (st >= SPRS_ERR_HIP) return st;
    SPRS_TRY(this->end_solve());
    return st;
}

template class KrylovBase<double>;
template class KrylovBase<float>;
template class KrylovBase<cplxf>;
template class KrylovBase<cplx>;
template class BicgStab<double>;
template class BicgStab<float>;
template class BicgStab<cplxf>;
template class BicgStab<cplx>;
template class MinRes<double>;
template class MinRes<float>;
template class MinRes<cplxf>;
template class MinRes<cplx>;

}  // namespace sprs
