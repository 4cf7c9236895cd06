// MINRES' third kernel (krylov.hip) as a structure other translation units can run: the lane-per-row SpMV of the compressed
// streams (spmv_dict.hip) executes its PROLOGUE (beta_new from the |v_new|^2 partials, the Givens rotation, the convergence
// bookkeeping) at the top of the NEXT iteration's SpMV launch, normalises the SpMV's operand on the fly (x[c] = v_new[c] * (1 / beta_new))
// and does M3's element-wise work for the rows each lane owns — "M3 inside M1", krylov.hip.
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
            T t = SAUNDERS ? sconj(qv.v[e]) : qv.v[e];              // :156 p = v  (cs:142 p = conj(q))
            t = sadd(t, smul(po.v[e], nr2));                        // :158
            t = sadd(t, smul(poo.v[e], nr3));                       // :159
            t = smulr(t, r1_inv);                                   // :160
            pv.v[e] = t;
            xv.v[e] = sadd(xv.v[e], smul(t, coef));                 // :162
        }
        stp<T, PK, NT>(v_new, i, nv);
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

// ---- spmv_dict.hip: M1 of iteration k + 1 with M3 of iteration k inside (no preconditioner, one GPU, the lane-per-row kernels of the
// compressed streams).  raw = v_new of iteration k as M2 left it (not normalised; NOT modified: the normalised vector goes to
// vn_out), y = A [conj] (raw / beta_new), partials of conj(raw / beta_new) . y in partAlpha.  m3.v_new is ignored.
template <class T, bool SAUNDERS>
int launch_spmv_m3(const sprs_csr *A, const MinresM3<T, false, SAUNDERS> &m3, const T *raw, T *vn_out, T *y, T *partAlpha);
bool spmv_m3_available(const sprs_csr *A);      // the handle's SpMV is one of those kernels (and the knob "spmv_fuse" allows it)

}  // namespace sprs
