// Solver objects behind sprs_bicgstab / sprs_minres / sprs_csminres.
#pragma once
#include "internal.hpp"

namespace sprs {

// Device-resident scalar state of a BiCGStab solve (bicg_stab.rs:84-88,127-186 locals).
template <class T>
struct BicgState {
    T rho, rho_old, alpha, w, beta;
    Real<T> r_norm, r0_norm_tol, tol2, pad0;
    long long its;
    int status, pad1;
};

// Device-resident scalar state of a MINRES / CSMINRES solve (minres.rs:60-64,81-83 locals).
// Two copies, indexed by iteration parity: kernels of iteration k read st[k&1], workgroup 0 of
// the last kernel writes st[(k+1)&1] — no workgroup ever reads a word another one is writing.
template <class T>
struct MinresState {
    T c, c_old, eta, alpha;
    Real<T> s, s_old, beta, beta_one, res_norm, threshold, pc_re, pad0;
};
template <class T>
struct MinresDev {
    MinresState<T> st[2];
    long long its;   // 0-based iteration index of the event recorded in `status`
    int status, pad;
};

struct SolverStats {
    double spmv_ms = 0.0, solve_ms = 0.0;
    int64_t spmv_launches = 0;
    int64_t fused_k2 = 0, fused_k4 = 0;   // of those (enqueued, profiled or not): chain launches that formed their input on the fly (K1 into K2 / K3 into K4)
    // among the TIMED launches (spmv_launches): how many read a dot operand that is not their input vector, how many were fused K2 / K4
    int64_t timed_dot_other = 0, timed_fused_k2 = 0, timed_fused_k4 = 0;
    int64_t steps = 0;                    // SpMV-class steps of the solve, timed or not
};

template <class T>
class KrylovBase {
   public:
    sprs_ctx *ctx = nullptr;
    const sprs_csr *A = nullptr;
    size_t n = 0, stride = 0;
    int nvec = 0;
    T *work = nullptr;           // nvec * stride
    T *rhs_buf = nullptr, *x_buf = nullptr;
    T *part = nullptr;           // NSLOT * MAX_GRID partials of T
    Real<T> *partD = nullptr;    // NSLOT * MAX_GRID real partials
    int mode = 0;
    double *trace = nullptr;
    size_t trace_cap = 0, trace_rows = 0;
    int profile = 0;             // 0 off; 1: every SpMV launch between HIP events; k >= 2: one pair of consecutive launches in k (a sample:
                                 // the events cost ~6 us per launch, krylov.hip profiled())
    size_t prof_calls = 0;       // SpMV-class steps of this solve so far (sampled or not)
    std::vector<size_t> ev_call; // per event pair: the step it brackets
    std::vector<unsigned char> ev_kind;   // per event pair: 1 = dot operand is not the input vector, 2 = fused K2, 4 = fused K4
    long last_pair = -1;         // the event pair of the last step (-1: it carried none)
    void mark_step(unsigned char kind) { if (last_pair >= 0) ev_kind[(size_t)last_pair] |= kind; }
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    std::vector<char> ev_noop;   // per profiled SpMV (event pair): launched after a restart request, i.e. returned at once — not a measurement
    SolverStats stats;
    // distributed operator (A->dist): all-reduced scalars live in `red`, 16-byte slots
    double *red = nullptr;       // device, 32 doubles
    unsigned int *fin_counter = nullptr;   // device: arrival counters of the in-launch finalizes (struct Fin), one per slot
    T *xext = nullptr;           // extended copy of a caller vector that has no halo tail

    int init(const sprs_csr *A_, size_t size, int nvec_);
    void destroy();
    T *vec(int i) { return work + (size_t)i * stride; }
    T *pslot(int s) { return part + (size_t)s * MAX_GRID; }
    Real<T> *dslot(int s) { return partD + (size_t)s * MAX_GRID; }
    int spmv(const T *x, T *y, int dot, const T *u, T *p0, T *p1, const int *status, bool conj_x = false, const sprs::Fin *fin = nullptr);
    // distributed: descriptor that makes the producing launch reduce its partials into red[2*slot ..] (empty otherwise)
    sprs::Fin fin_for(int slot, const void *base0, const void *base1, int P) const;
    template <class F> int profiled(F &&run, bool one_kernel);   // run() = one SpMV-class step, bracketed by the profile's events when a profile is taken
    void profile_discard_last(size_t launches);   // the last `launches` profiled SpMVs were no-ops (status word set): keep them out of the mean
    int begin_solve();
    int end_solve();
    void trace_row(double a0, double a1, T b, T c, T d);
    // host-slice wrapper around a device solve
    template <class F>
    int solve_host(const T *rhs, size_t rhs_len, T *x, size_t x_len, F &&dev_solve);
    int ew_grid() const;  // workgroups used by the fused element-wise kernels for this n
    sprs_comm *comm() const { return A->dist ? A->dist->comm : nullptr; }
    // Hand a producer's partials to its consumer kernel.  Single GPU: the consumer re-reduces the
    // P partials itself.  Distributed: the producer's last workgroup has reduced them into `red`
    // (fixed order; fin_for), these all-reduce over the ranks, and the consumer reads one value.
    // `slot` picks a 16-byte cell of `red`.
    struct PartT { const T *p; int P; unsigned int tag = 0; };            // tag != 0: p = this rank's mailbox entries of the hand-off, P = world
    struct PartD { const Real<T> *p; int P; unsigned int tag = 0; };
    bool use_p2p() const { return A->dist && A->dist->comm->p2p && ctx->p2p_allreduce != 0; }
    unsigned long long mb_timeout() const { return (unsigned long long)(ctx->p2p_timeout_ms < 1 ? 1 : ctx->p2p_timeout_ms) * 100000ull; }   // ticks of the 100 MHz wall clock
    const void *mbox_entries(int slot) const;    // this rank's mailbox at the CURRENT hand-off of `slot`
    int red1(const T *a, int P, int slot, PartT *oa);
    int red2(const T *a, const T *b, int P, int slot, PartT *oa, PartT *ob);
    int redD1(const Real<T> *a, int P, int slot, PartD *oa);
    int redDT(const Real<T> *a, const T *b, int P, int slot, PartD *oa, PartT *ob);
    int norm2(const T *x, Real<T> *out) { return norm2_host<T>(ctx, n, x, out, comm()); }
    int cdot(const T *x, const T *y, T *out) { return dot_host<T>(ctx, n, x, y, true, out, comm()); }
};

template <class T>
class BicgStab : public KrylovBase<T> {
   public:
    BicgState<T> *d_state = nullptr, *h_state = nullptr;
    int create(const sprs_csr *A, size_t size);
    void destroy();
    int solve_dev(const sprs_diag *P, const T *rhs, size_t rhs_len, T *x, size_t x_len, size_t max_iter, Real<T> tol,
                  size_t *its_out, Real<T> *res_out);

   private:
    template <class V>
    int run(const V *dinv, const T *rhs, T *x, size_t max_iter, Real<T> tol, size_t *its_out, Real<T> *res_out);
    template <class V>
    int run_literal(const V *dinv, const T *rhs, T *x, size_t max_iter, Real<T> tol, size_t *its_out, Real<T> *res_out);
};

template <class T>
class MinRes : public KrylovBase<T> {
   public:
    MinresDev<T> *d_state = nullptr, *h_state = nullptr;
    bool saunders = false;  // CSMINRES
    int create(const sprs_csr *A, size_t size, bool saunders_);
    void destroy();
    int solve_dev(const sprs_diag *P, const T *rhs, size_t rhs_len, T *x, size_t x_len, size_t max_iter, Real<T> tol,
                  size_t *its_out, Real<T> *res_out);

   private:
    template <class V>
    int run(const V *dinv, const T *rhs, T *x, size_t max_iter, Real<T> tol, size_t *its_out, Real<T> *res_out);
    template <class V>
    int run_literal(const V *dinv, const T *rhs, T *x, size_t max_iter, Real<T> tol, size_t *its_out, Real<T> *res_out);
};

template <class T>
template <class F>
int KrylovBase<T>::solve_host(const T *rhs, size_t rhs_len, T *x, size_t x_len, F &&dev_solve) {
    // size checks first (bicg_stab.rs:44-53): nothing is copied on a mismatch
    if (rhs_len != n) return SPRS_INCOMPATIBLE_RHS_SIZE;
    if (x_len != n) return SPRS_INCOMPATIBLE_X_SIZE;
    SPRS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!rhs_buf) SPRS_HIP_TRY(ctx, hipMalloc((void **)&rhs_buf, sizeof(T) * stride));
    if (!x_buf) SPRS_HIP_TRY(ctx, hipMalloc((void **)&x_buf, sizeof(T) * stride));
    SPRS_HIP_TRY(ctx, hipMemcpyAsync(rhs_buf, rhs, sizeof(T) * n, hipMemcpyHostToDevice, ctx->stream));
    SPRS_HIP_TRY(ctx, hipMemcpyAsync(x_buf, x, sizeof(T) * n, hipMemcpyHostToDevice, ctx->stream));
    int st = dev_solve(rhs_buf, x_buf);
    if (st >= SPRS_ERR_HIP) return st;
    // x is in/out in the reference and is left modified on Err as well
    SPRS_HIP_TRY(ctx, hipMemcpyAsync(x, x_buf, sizeof(T) * n, hipMemcpyDeviceToHost, ctx->stream));
    SPRS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return st;
}

}  // namespace sprs

// opaque C handles: type-erased over T
struct sprs_bicgstab {
    int dtype;
    void *impl;
};
struct sprs_minres {
    int dtype;
    void *impl;
};
struct sprs_csminres {
    int dtype;
    void *impl;
};
