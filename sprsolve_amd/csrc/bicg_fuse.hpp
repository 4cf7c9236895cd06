// BiCGStab's vector-update kernels K1 and K3 (krylov.hip) as structures other translation units can run: the plane-streaming
// chain SpMV (spmv_chain.hip) executes their PROLOGUES (the scalar logic of the recurrence: convergence / restart / breakdown
// tests, beta, alpha) at the top of its own launch and forms their element-wise results while it stages its x windows — the
// updated vector is consumed without a pass of its own ("fused SpMV input", krylov.hip).
#pragma once
#include "device.hpp"
#include "krylov.hpp"

namespace sprs {

__device__ __forceinline__ bool first_thread() { return blockIdx.x == 0 && threadIdx.x == 0; }

// ======================================================================= BiCGStab kernels
// K1  bicg_stab.rs:123-156 (+ :321-328 with a preconditioner)
//   r_norm = norm2(r); converged?  rho = r0.r; restart?  beta = (rho/rho_old)*(alpha/w)
//   p = v*(-beta*w) + p*beta ;  p += r*1 ;  [y = M^-1 p]
template <class T, class V, bool PC>
struct BicgK1 {
    BicgState<T> *S; const Real<T> *partN; const T *partRho; int P; int mode;
    const T *v; const T *r; T *p; const V *dinv; T *y;
    T a, beta;
    unsigned int tag = 0; unsigned long long mb_timeout = 0;     // peer-to-peer hand-off: partN = this rank's mailbox entries, P = world (device.hpp, mbox_sum2)
    __device__ __forceinline__ bool prologue() {
        __shared__ Real<T> smD[NWAVE];
        __shared__ T smT[NWAVE];
        // every load of the prologue is issued before any is consumed (state words, then both partial arrays): one
        // memory round trip where the literal order (status -> |r| partials -> tol -> rho partials -> ...) paid four.
        // Only fields that no workgroup of THIS launch writes are read (rho / r_norm / beta are written below).
        const int status = S->status;
        const Real<T> tol2 = S->tol2, r0_norm_tol = S->r0_norm_tol;
        const T w = S->w, rho_old = S->rho_old, alpha = S->alpha;
        T rho; Real<T> r_norm;
        if (mode == 0) {
            Real<T> sN; T sR;
            if (tag != 0) {                                          // (status first: a stopped solve's producers posted nothing)
                if (status != ST_RUNNING) return false;
                if (!mbox_sum2(MboxSrc{reinterpret_cast<const unsigned long long *>(partN), P, tag, mb_timeout}, sN, sR)) {
                    if (first_thread()) S->status = ST_COMM_TIMEOUT;
                    return false;
                }
            } else {
                reduce_partials2(partN, partRho, P, smD, smT, sN, sR);  // :123 |r|^2, :128 r0.r
                if (status != ST_RUNNING) return false;
            }
            r_norm = ssqrt(sN);                                      // :123
            if (r_norm <= tol2) {                                    // :124
                if (first_thread()) { S->r_norm = r_norm; S->status = ST_CONVERGED; }
                return false;
            }
            rho = sR;                                                // :128
            if (sabs(rho) < r0_norm_tol) {                           // :131 -> host runs :132-145
                if (first_thread()) { S->r_norm = r_norm; S->status = ST_RESTART; }
                return false;
            }
        } else {  // resumed after the host-side restart: rho, r0_norm_tol already updated
            if (status != ST_RUNNING) return false;
            rho = S->rho; r_norm = S->r_norm;
        }
        beta = smul(sdiv(rho, rho_old), sdiv(alpha, w));             // :146
        a = smul(sneg(beta), w);                                     // :155  -beta * w
        if (first_thread()) { S->rho = rho; S->r_norm = r_norm; S->beta = beta; }
        return true;
    }
    template <int PK, bool NT> __device__ __forceinline__ void run(int64_t i) const {
        auto vv = ldp<T, PK, NT>(v, i); auto pv = ldp<T, PK, NT>(p, i); auto rv = ldp<T, PK, NT>(r, i);
        Pack<T, PK> yv;
        [[maybe_unused]] Pack<V, PK> dv;
        if (PC) dv = ldp<V, PK, NT>(dinv, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) {
            T t = sadd(smul(vv.v[e], a), smul(pv.v[e], beta));      // :155 axpby
            t = sadd(t, smul(rv.v[e], sone<T>()));                  // :156 axpy(one, r, p)
            pv.v[e] = t;
            if (PC) yv.v[e] = smulv(t, dv.v[e]);                    // :328
        }
        stp<T, PK, NT>(p, i, pv);
        if (PC) stp<T, PK, NT>(y, i, yv);
    }
    __device__ __forceinline__ void epilogue() const {}
};

// K3  bicg_stab.rs:163-172 (+ :343):  alpha = rho / (r0.v) ; r -= alpha*v ; [z = M^-1 r]
template <class T, class V, bool PC>
struct BicgK3 {
    BicgState<T> *S; const T *partB; int P; int check_breakdown;
    const T *v; T *r; const V *dinv; T *z;
    T na;
    unsigned int tag = 0; unsigned long long mb_timeout = 0;     // peer-to-peer hand-off (see BicgK1)
    __device__ __forceinline__ bool prologue() {
        __shared__ T smT[NWAVE];
        const int status = S->status;                               // requested together with the partials
        const T rho = S->rho;
        T tmp;
        if (tag != 0) {
            if (status != ST_RUNNING) return false;
            if (!mbox_sum1(MboxSrc{reinterpret_cast<const unsigned long long *>(partB), P, tag, mb_timeout}, tmp)) {
                if (first_thread()) S->status = ST_COMM_TIMEOUT;
                return false;
            }
        } else {
            tmp = reduce_partials(partB, P, smT);                   // :163
            if (status != ST_RUNNING) return false;
        }
        if (check_breakdown && sabs(tmp) <= 0.0) {                  // :164-167
            if (first_thread()) S->status = ST_BREAKDOWN;
            return false;
        }
        const T alpha = sdiv(rho, tmp);                             // :169
        na = sneg(alpha);
        if (first_thread()) S->alpha = alpha;
        return true;
    }
    template <int PK, bool NT> __device__ __forceinline__ void run(int64_t i) const {
        auto vv = ldp<T, PK, NT>(v, i); auto rv = ldp<T, PK, NT>(r, i);
        Pack<T, PK> zv;
        [[maybe_unused]] Pack<V, PK> dv;
        if (PC) dv = ldp<V, PK, NT>(dinv, i);
#pragma unroll
        for (int e = 0; e < PK; ++e) {
            rv.v[e] = sadd(rv.v[e], smul(vv.v[e], na));             // :172
            if (PC) zv.v[e] = smulv(rv.v[e], dv.v[e]);              // :343
        }
        stp<T, PK, NT>(r, i, rv);
        if (PC) stp<T, PK, NT>(z, i, zv);
    }
    __device__ __forceinline__ void epilogue() const {}
};

// ---- spmv_chain.hip: the chain SpMV with the preceding vector update formed on the fly (f64, no preconditioner, single GPU)
// K3 into K4 (bicg_stab.rs:163-178):  alpha from partB (pro's prologue) ; s = r + v * (-alpha), formed on the fly and — s_out ==
//   nullptr — not stored (the pair2 walk of the blocks outside the chains honours a non-null s_out) ; t = A s ; partials of t.t and t.s
int launch_chain_k4f(const sprs_csr *A, int g, const BicgK3<double, double, false> &pro, const double *r, const double *v, double *s_out,
                     double *t, double *partTT, double *partTR, const int *status);
// K1 into K2 (bicg_stab.rs:123-163):  norm / convergence / restart / beta (pro's prologue) ; p' = (v * (-beta w) + p * beta) + r -> p_out ;
//   v' = A p' -> v_out ; partials of r0.v'
int launch_chain_k2f(const sprs_csr *A, int g, const BicgK1<double, double, false> &pro, const double *v_old, const double *p, const double *r,
                     double *p_out, double *v_out, const double *r0, double *partB, const int *status);

}  // namespace sprs
