// MINRES' third kernel (krylov.hip) as a structure other translation units can run, and the DEFERRED form of it ("M3 deferred",
// krylov.hip): M3 of iteration k — beta_new, the normalisation of v_new, the Givens rotation, p, x, the convergence test — is not
// launched.  The SpMV of iteration k + 1 (spmv_dict.hip, the lane-per-row kernels of the compressed streams) runs M3's PROLOGUE to
// get 1 / beta_new and multiplies by v_new[c] * (1 / beta_new) formed in its gathers; the normalised vector is never stored.
// M3's element-wise work then rides in the same launch as M2 of iteration k + 1 (MinresM23 below), which reads v_new(k) and
// v(k) anyway: 9 vector passes instead of M3's 8 + M2's 4, two launches per iteration instead of three.
#pragma once
#include "bicg_fuse.hpp"

namespace sprs {

// M3  minres.rs:120-168 (cs_minres.rs:106-154 with SAUNDERS):  beta_new, normalise v_new
//     [and w_new], Givens rotation, p = q - r2*p_old - r3*p_oold, p *= 1/r1, x += c*eta*beta_1*p,
//     res_norm *= |s| ; converged?  eta *= -s
template <class T, bool PC, bool SAUNDERS>
struct MinresM3 {
    MinresDev<T> *D; int par; long long its; const Real<T> *partBeta; const T *partBeta2; int P;
    T *v_new; T *w_new; const T *q; const T *p_old; const T *p_oold; T *p; T *x;
    Real<T> inv, r1_inv, beta_new, s_new; T nr2, nr3, coef, c_new;
    unsigned int tag = 0; unsigned long long mb_timeout = 0;     // peer-to-peer hand-off (see BicgK1): partBeta / partBeta2 = this rank's mailbox entries
    // M3 deferred: q may be a RAW vector (the un-normalised v_new of the iteration before, never normalised in memory): q_raw != 0 =>
    // its elements are multiplied by 1 / S.beta — the factor that iteration's M3 computed — and, q_back != nullptr, written back
    // normalised (the flush before a poll: afterwards every vector is what the plain kernels expect)
    int q_raw = 0; T *q_back = nullptr; Real<T> qs = 1;
    // This launch's own epilogue sets the status to "converged at `its`" when workgroup 0 is done — possibly before another
    // workgroup of the SAME launch has read the status word.  That workgroup must still do its share of this iteration (the
    // reference updates x, then tests: minres.rs:162-167), so the event carries its iteration and this launch does not
    // stop for its own.  (Found by the solver fuzz: one solve in ~10^5 returned x with only some tiles updated.)
    __device__ __forceinline__ int converged_word() const { return ST_CONVERGED | (int)((its & 0x7ffffff) << 4); }
    __device__ __forceinline__ bool stopped(int status) const { return status != ST_RUNNING && status != converged_word(); }
    __device__ __forceinline__ bool prologue() {
        __shared__ Real<T> smD[NWAVE];
        __shared__ T smT[NWAVE];
        const int status = D->status;                               // state words requested together with the partials
        const MinresState<T> S = D->st[par];                        // (a copy: st[par] is not written by this launch)
        if (tag != 0 && stopped(status)) return false;              // (status first: a stopped solve's producers posted nothing)
        if (PC) {
            T b2;
            if (tag != 0) {
                if (!mbox_sum1(MboxSrc{reinterpret_cast<const unsigned long long *>(partBeta2), P, tag, mb_timeout}, b2)) {
                    if (first_thread()) D->status = ST_COMM_TIMEOUT;
                    return false;
                }
            } else {
                b2 = reduce_partials(partBeta2, P, smT);            // :278
            }
            if (stopped(status)) return false;
            if (sre(b2) < seps<Real<T>>() || sim(b2) > seps<Real<T>>() * sre(b2)) {         // :279-287
                if (first_thread()) { D->st[par].pc_re = sre(b2); D->its = its; D->status = ST_INVALID_PC; }
                return false;
            }
            beta_new = ssqrt(sre(b2));                               // :288
        } else {
            Real<T> bsq;
            if (tag != 0) {
                if (!mbox_sum1(MboxSrc{reinterpret_cast<const unsigned long long *>(partBeta), P, tag, mb_timeout}, bsq)) {
                    if (first_thread()) D->status = ST_COMM_TIMEOUT;
                    return false;
                }
            } else {
                bsq = reduce_partials(partBeta, P, smD);
            }
            beta_new = ssqrt(bsq);                                   // :120
            if (stopped(status)) return false;
        }
        inv = Real<T>(1) / beta_new;                                       // :121 / :289
        qs = q_raw ? Real<T>(1) / S.beta : Real<T>(1);                     // (S.beta is the beta_new of the iteration before: the same quotient)
        const Real<T> beta = S.beta;
        const T c = S.c, c_old = S.c_old, alpha = S.alpha;
        const Real<T> s = S.s, s_old = S.s_old;
        const Real<T> r3 = s_old * beta;                                                    // :132
        const T tr = smulr(SAUNDERS ? sconj(c_old) : c_old, beta);                         // :133 / cs:120
        const T r2 = sadd(smulr(alpha, s), smul(c, tr));                                   // :134
        const T r1_hat = ssub(smul(SAUNDERS ? sconj(c) : c, alpha), smulr(tr, s));         // :136 / cs:122
        r1_inv = Real<T>(1) / ssqrt(ssq(r1_hat) + beta_new * beta_new);                            // :139-140
        c_new = smulr(SAUNDERS ? sconj(r1_hat) : r1_hat, r1_inv);                          // :147 / cs:133
        s_new = beta_new * r1_inv;                                                         // :148
        nr2 = sneg(r2); nr3 = sfromr<T>(-r3);
        coef = smulr(smul(c_new, S.eta), S.beta_one);                                      // :162
        return true;
    }
    template <int PK, bool NT> __device__ __forceinline__ void run(int64_t i) const {
        auto nv = ldp<T, PK, NT>(v_new, i);
        auto qv = ldp<T, PK, NT>(q, i); auto po = ldp<T, PK, NT>(p_old, i); auto poo = ldp<T, PK, NT>(p_oold, i);
        auto xv = ldp<T, PK, NT>(x, i);
        [[maybe_unused]] Pack<T, PK> wv;
        if (PC) wv = ldp<T, PK, NT>(w_new, i);
        Pack<T, PK> pv;
#pragma unroll
        for (int e = 0; e < PK; ++e) {
            nv.v[e] = smulr(nv.v[e], inv);                          // :121 / :290
            if (PC) wv.v[e] = smulr(wv.v[e], inv);                  // :291
            if (!PC) qv.v[e] = smulr(qv.v[e], qs);                  // (a raw q: its normalisation, :121 of the iteration before; else * 1, exact)
            T t = SAUNDERS ? sconj(qv.v[e]) : qv.v[e];              // :156 p = v  (cs:142 p = conj(q))
            t = sadd(t, smul(po.v[e], nr2));                        // :158
            t = sadd(t, smul(poo.v[e], nr3));                       // :159
            t = smulr(t, r1_inv);                                   // :160
            pv.v[e] = t;
            xv.v[e] = sadd(xv.v[e], smul(t, coef));                 // :162
        }
        stp<T, PK, NT>(v_new, i, nv);
        if (!PC && q_back) stp<T, PK, NT>(q_back, i, qv);
        if (PC) stp<T, PK, NT>(w_new, i, wv);
        stp<T, PK, NT>(p, i, pv);
        stp<T, PK, NT>(x, i, xv);
    }
    __device__ __forceinline__ void epilogue() const {
        if (!first_thread()) return;
        const MinresState<T> &S = D->st[par];
        MinresState<T> N;
        N.c_old = S.c; N.s_old = S.s;                               // :142-143
        N.c = c_new; N.s = s_new;                                   // :147-148
        N.alpha = S.alpha;
        N.beta = beta_new; N.beta_one = S.beta_one; N.threshold = S.threshold;
        N.res_norm = S.res_norm * sabs(s_new);                      // :164
        N.eta = smulr(S.eta, -s_new);                               // :168
        N.pc_re = 0.0; N.pad0 = 0.0;
        D->st[par ^ 1] = N;
        if (N.res_norm < S.threshold) { D->its = its; D->status = converged_word(); }   // :165-167
    }
};

// M3 of iteration k together with M2 of iteration k + 1 (minres.rs:117-120 after :120-168): one pass over
//   q = v(k) (raw or normalised, as above) — M3's p source AND M2's v_old —, p_old, p_oold, x, raw = v_new(k) un-normalised — M2's v
//   after * 1 / beta_new —, v_new = A v(k + 1) from the SpMV just before;  writes p, x, v_new and the |v_new|^2 partials.
// Same expressions, same order per element as the two kernels; the partials are summed by the same threads in the same order
// (fused_kernel's walk), so beta_new of the next iteration is bit-identical.  partBeta (read by m3's prologue) and partBetaOut are
// different arrays: a workgroup may write its new partial while another is still reducing the old ones.
template <class T, bool SAUNDERS>
struct MinresM23 {
    MinresM3<T, false, SAUNDERS> m3;        // its v_new / w_new are unused here
    const T *partAlpha; int PA; const T *raw; T *v_new; Real<T> *partBetaOut;
    T nb, na, alpha; Real<T> accD;
    __device__ __forceinline__ bool prologue() {
        __shared__ T smA[NWAVE];
        if (!m3.prologue()) return false;                           // M3's stopping rule: its own "converged at k" word does not stop this launch
        alpha = reduce_partials(partAlpha, PA, smA);                // :116 of iteration k + 1
        nb = sfromr<T>(-m3.beta_new);                               // :117 (beta of iteration k + 1 = beta_new of k)
        na = sneg(alpha);                                           // :118
        accD = 0.0;
        return true;
    }
    template <int PK, bool NT> __device__ __forceinline__ void run(int64_t i) {
        auto qv = ldp<T, PK, NT>(m3.q, i); auto po = ldp<T, PK, NT>(m3.p_old, i); auto poo = ldp<T, PK, NT>(m3.p_oold, i);
        auto xv = ldp<T, PK, NT>(m3.x, i);
        auto rv = ldp<T, PK, NT>(raw, i); auto nv = ldp<T, PK, NT>(v_new, i);
        Pack<T, PK> pv;
#pragma unroll
        for (int e = 0; e < PK; ++e) {
            qv.v[e] = smulr(qv.v[e], m3.qs);                        // v(k): :121 of iteration k - 1 (or * 1)
            T t = SAUNDERS ? sconj(qv.v[e]) : qv.v[e];              // :156 (cs:142)
            t = sadd(t, smul(po.v[e], m3.nr2));                     // :158
            t = sadd(t, smul(poo.v[e], m3.nr3));                    // :159
            t = smulr(t, m3.r1_inv);                                // :160
            pv.v[e] = t;
            xv.v[e] = sadd(xv.v[e], smul(t, m3.coef));              // :162
            const T vn = smulr(rv.v[e], m3.inv);                    // v(k + 1): :121
            T w = sadd(nv.v[e], smul(qv.v[e], nb));                 // :117
            w = sadd(w, smul(vn, na));                              // :118
            nv.v[e] = w;
            accD = accD + ssq(w);                                   // :120
        }
        stp<T, PK, NT>(m3.p, i, pv);
        stp<T, PK, NT>(m3.x, i, xv);
        stp<T, PK, NT>(v_new, i, nv);
    }
    __device__ __forceinline__ void epilogue() {
        __shared__ Real<T> smD[NWAVE];
        const Real<T> s = block_sum(accD, smD);
        if (threadIdx.x == 0) partBetaOut[blockIdx.x] = s;
        m3.epilogue();                                              // st[par ^ 1] = the state after iteration k, the convergence event
        if (first_thread()) m3.D->st[m3.par ^ 1].alpha = alpha;     // ... and M2's record of alpha (k + 1)
    }
};

// ---- spmv_dict.hip: M1 of iteration k + 1 on the un-normalised v_new of iteration k ("M3 deferred"): y = A [conj] (raw / beta_new),
// partials of conj(raw / beta_new) . y in partAlpha; beta_new from m3's prologue (only D, par, its, partBeta, P of m3 are used;
// nothing of the solver's state is written).
template <class T, bool SAUNDERS>
int launch_spmv_scaled(const sprs_csr *A, const MinresM3<T, false, SAUNDERS> &m3, const T *raw, T *y, T *partAlpha);
bool spmv_scaled_available(const sprs_csr *A);      // the handle's SpMV is one of those kernels (and the knob "spmv_fuse" allows it)

}  // namespace sprs
