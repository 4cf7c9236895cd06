/* TEST INFRASTRUCTURE ONLY — CPU oracle, template body.
 *
 * Included twice by sprs_oracle.c:  SFX=d (T=R)  and  SFX=z (T=orc_c64).
 * A statement-for-statement restatement of the reference's hot path; every function cites
 * the reference file:line it follows (paths relative to the reference root).  Sums are
 * strict left folds, no fma (compile with -ffp-contract=off).
 *
 * R = T::Real (double or float).  Index type is int64 (the reference's sprs::CsMat<T> uses usize, mat.rs:199).
 */

#define FN(name) ORC_CAT3(orc_, name, SFX_US)
#define S(op) ORC_CAT2(SP, op)

/* ------------------------------------------------------------------ vecalg.rs:556-605 */

/* vecalg.rs:556-561  dot_fallback: fold(zero, acc + x*y), no conjugate */
T FN(dot)(int64_t n, const T *x, const T *y) {
    T acc = S(zero)();
    for (int64_t i = 0; i < n; ++i) acc = S(add)(acc, S(mul)(x[i], y[i]));
    return acc;
}

/* vecalg.rs:563-568  conj_dot_fallback: fold(zero, acc + conj(x)*y) */
T FN(conj_dot)(int64_t n, const T *x, const T *y) {
    T acc = S(zero)();
    for (int64_t i = 0; i < n; ++i) acc = S(add)(acc, S(mul)(S(conj)(x[i]), y[i]));
    return acc;
}

/* vecalg.rs:570-575  axpy_fallback: y += x * a   (a: T) */
void FN(axpy)(int64_t n, T a, const T *x, T *y) {
    for (int64_t i = 0; i < n; ++i) y[i] = S(add)(y[i], S(mul)(x[i], a));
}

/* vecalg.rs:577-583  conj_fallback: out = conj(in) */
void FN(conj)(int64_t n, const T *in, T *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = S(conj)(in[i]);
}

/* vecalg.rs:585-590  axpby_fallback: y = x*a + y*b */
void FN(axpby)(int64_t n, T a, const T *x, T b, T *y) {
    for (int64_t i = 0; i < n; ++i) y[i] = S(add)(S(mul)(x[i], a), S(mul)(y[i], b));
}

/* vecalg.rs:592-595  scale_fallback: v *= a */
void FN(scale)(int64_t n, T a, T *v) {
    for (int64_t i = 0; i < n; ++i) v[i] = S(mul)(v[i], a);
}

/* vecalg.rs:596-599  rscale_fallback: v = v.mul_real(a) */
void FN(rscale)(int64_t n, R a, T *v) {
    for (int64_t i = 0; i < n; ++i) v[i] = S(mulr)(v[i], a);
}

/* vecalg.rs:601-605  norm2_fallback: R_SQRT(fold(0, acc + x.square())), unscaled */
R FN(norm2)(int64_t n, const T *v) {
    R acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc = acc + S(sq)(v[i]);
    return R_SQRT(acc);
}

/* ------------------------------------------------------------------ mat.rs:68-152 */

/* mat.rs:68-129 CSR branch. y is zero-filled (:71) then each row is the strict left fold
 * acc + x[col]*val starting from zero (:100-105).  Row-parallel exactly where rayon is
 * (:91-95, chunks >= 128 rows); per-row bits do not depend on the thread split. */
void FN(spmv_csr)(int64_t nrows, const int64_t *indptr, const int64_t *indices, const T *data,
                  const T *x, T *y, int parallel) {
    for (int64_t i = 0; i < nrows; ++i) y[i] = S(zero)();
    if (parallel) {
#pragma omp parallel for schedule(dynamic, 4096) if (nrows >= 256)
        for (int64_t i = 0; i < nrows; ++i) {
            int64_t st = indptr[i], nn = indptr[i + 1] - st;
            const int64_t *li = indices + st;
            const T *ld = data + st;
            T acc = S(zero)();
            for (int64_t k = 0; k < nn; ++k) acc = S(add)(acc, S(mul)(x[li[k]], ld[k]));
            y[i] = acc;
        }
    } else {
        for (int64_t i = 0; i < nrows; ++i) {
            int64_t st = indptr[i], nn = indptr[i + 1] - st;
            T acc = S(zero)();
            for (int64_t k = 0; k < nn; ++k)
                acc = S(add)(acc, S(mul)(x[indices[st + k]], data[st + k]));
            y[i] = acc;
        }
    }
}

/* mat.rs:130-142 CSC branch: serial scatter y[row] += x[col] * value after zero fill */
void FN(spmv_csc)(int64_t nrows, int64_t ncols, const int64_t *indptr, const int64_t *indices,
                  const T *data, const T *x, T *y) {
    for (int64_t i = 0; i < nrows; ++i) y[i] = S(zero)();
    for (int64_t c = 0; c < ncols; ++c) {
        T m = x[c];
        for (int64_t k = indptr[c]; k < indptr[c + 1]; ++k)
            y[indices[k]] = S(add)(y[indices[k]], S(mul)(m, data[k]));
    }
}

typedef struct {
    int64_t n;
    const int64_t *indptr, *indices;
    const T *data;
    int parallel;
} FN(csr);

static void FN(mv)(const FN(csr) * A, const T *x, T *y) {
    FN(spmv_csr)(A->n, A->indptr, A->indices, A->data, x, y, A->parallel);
}

/* mat.rs:145-152 mul_vec_dot_unchecked: y = A x ; return conj_dot(x, y) */
T FN(spmv_csr_dot)(int64_t nrows, const int64_t *indptr, const int64_t *indices, const T *data,
                   const T *x, T *y, int parallel) {
    FN(spmv_csr)(nrows, indptr, indices, data, x, y, parallel);
    return FN(conj_dot)(nrows, x, y);
}

/* ------------------------------------------------------------------ precond.rs:20-52 */

/* precond.rs:20-29 DiagPrecond::new: diag_inv = V::one() / v  (V real) */
void FN(diag_inv_real)(int64_t n, const R *diag, R *dinv) {
    for (int64_t i = 0; i < n; ++i) dinv[i] = (R)1 / diag[i];
}
/* precond.rs:48-52 apply: out = in * diag_inv   (T: Mul<V>; V real => mul_real) */
static void FN(pc_apply)(int64_t n, const void *dinv, int dinv_complex, const T *in, T *out) {
#if SFX_IS_COMPLEX
    if (dinv_complex) {
        const CT *d = (const CT *)dinv;
        for (int64_t i = 0; i < n; ++i) out[i] = ORC_CAT2(CSP, mul)(in[i], d[i]);
        return;
    }
#else
    (void)dinv_complex;
#endif
    const R *d = (const R *)dinv;
    for (int64_t i = 0; i < n; ++i) out[i] = S(mulr)(in[i], d[i]);
}
void FN(diag_apply)(int64_t n, const void *dinv, int dinv_complex, const T *in, T *out) {
    FN(pc_apply)(n, dinv, dinv_complex, in, out);
}

/* ------------------------------------------------------------------ reductions as the SOLVERS call them
 *
 * orc_red_mode 0 (default): the reference's serial left folds above (vecalg.rs:563-568, 601-605).
 * orc_red_mode 1: the SAME products, added in the order of libsprsolve_hip's stand-alone reduction kernels
 * (sprsolve_amd/csrc/blas1.hip dot_kernel / nrm2sq_kernel / finalize_kernel, csrc/device.hpp block_sum): per thread
 * a grid-stride fold over 16-byte packs, a 64-lane butterfly per wavefront (v[l] += v[l + off], off = 32 .. 1), the
 * four wavefronts of a workgroup left to right, one partial per workgroup, the partials folded the same way by one
 * workgroup.  Only the ORDER of additions differs from the reference; every product and every element-wise statement
 * is unchanged.  Purpose (test infrastructure): with SpMV and the element-wise kernels bit-exact, the library's
 * "literal" solver mode then has to reproduce this oracle's whole recurrence BIT FOR BIT — every scalar of every
 * iteration, every restart / breakdown / convergence decision — which pins the host recurrences statement for
 * statement instead of to a tolerance (tests/test_gpu_parity.py::test_literal_mode_is_the_oracle_bit_for_bit). */
static T FN(tree256)(T *v) {                         /* block_sum over 256 threads; v is clobbered */
    T w[4];
    for (int wv = 0; wv < 4; ++wv) {
        T *q = v + 64 * wv;
        for (int off = 32; off > 0; off >>= 1)
            for (int l = 0; l < off; ++l) q[l] = S(add)(q[l], q[l + off]);
        w[wv] = q[0];
    }
    T acc = w[0];
    for (int wv = 1; wv < 4; ++wv) acc = S(add)(acc, w[wv]);
    return acc;
}
static R FN(tree256r)(R *v) {
    R w[4];
    for (int wv = 0; wv < 4; ++wv) {
        R *q = v + 64 * wv;
        for (int off = 32; off > 0; off >>= 1)
            for (int l = 0; l < off; ++l) q[l] = q[l] + q[l + off];
        w[wv] = q[0];
    }
    return ((w[0] + w[1]) + w[2]) + w[3];
}
static int64_t FN(red_grid)(int64_t n, int pk) {     /* blas1.hip red_grid -> internal.hpp balanced_grid */
    const int64_t work = (n / pk + 255) / 256, g0 = orc_red_grid;
    if (work <= g0) return work < 1 ? 1 : work;
    const int64_t trips = (work + g0 - 1) / g0;
    int64_t g = (work + trips - 1) / trips;
    g = (g + 7) & ~(int64_t)7;
    return g > g0 ? g0 : g;
}
static T FN(gpu_order_dot)(int64_t n, const T *x, const T *y, int conj) {
    const int pk = sizeof(T) >= 16 ? 1 : (int)(16 / sizeof(T));
    const int64_t g = FN(red_grid)(n, pk), np = n / pk, st = g * 256;
    T *part = (T *)malloc(sizeof(T) * (size_t)g);
    T v[256];
    for (int64_t b = 0; b < g; ++b) {
        for (int t = 0; t < 256; ++t) {
            T acc = S(zero)();
            for (int64_t i = b * 256 + t; i < np; i += st)
                for (int e = 0; e < pk; ++e) {
                    const int64_t k = i * pk + e;
                    acc = S(add)(acc, S(mul)(conj ? S(conj)(x[k]) : x[k], y[k]));
                }
            if (pk > 1) {
                const int64_t k = np * pk + b * 256 + t;
                if (k < n) acc = S(add)(acc, S(mul)(conj ? S(conj)(x[k]) : x[k], y[k]));
            }
            v[t] = acc;
        }
        part[b] = FN(tree256)(v);
    }
    for (int t = 0; t < 256; ++t) {
        T acc = S(zero)();
        for (int64_t j = t; j < g; j += 256) acc = S(add)(acc, part[j]);
        v[t] = acc;
    }
    free(part);
    return FN(tree256)(v);
}
static R FN(gpu_order_norm2)(int64_t n, const T *x) {
    const int pk = sizeof(T) >= 16 ? 1 : (int)(16 / sizeof(T));
    const int64_t g = FN(red_grid)(n, pk), np = n / pk, st = g * 256;
    R *part = (R *)malloc(sizeof(R) * (size_t)g);
    R v[256];
    for (int64_t b = 0; b < g; ++b) {
        for (int t = 0; t < 256; ++t) {
            R acc = 0;
            for (int64_t i = b * 256 + t; i < np; i += st)
                for (int e = 0; e < pk; ++e) acc = acc + S(sq)(x[i * pk + e]);
            if (pk > 1) {
                const int64_t k = np * pk + b * 256 + t;
                if (k < n) acc = acc + S(sq)(x[k]);
            }
            v[t] = acc;
        }
        part[b] = FN(tree256r)(v);
    }
    for (int t = 0; t < 256; ++t) {
        R acc = 0;
        for (int64_t j = t; j < g; j += 256) acc = acc + part[j];
        v[t] = acc;
    }
    free(part);
    return R_SQRT(FN(tree256r)(v));                   /* vecalg.rs:604 */
}
static T FN(s_cdot)(int64_t n, const T *x, const T *y) {
    return orc_red_mode == 1 ? FN(gpu_order_dot)(n, x, y, 1) : FN(conj_dot)(n, x, y);
}
static R FN(s_norm2)(int64_t n, const T *x) {
    return orc_red_mode == 1 ? FN(gpu_order_norm2)(n, x) : FN(norm2)(n, x);
}
/* for tests of the emulation itself */
T FN(conj_dot_gpu_order)(int64_t n, const T *x, const T *y) { return FN(gpu_order_dot)(n, x, y, 1); }
R FN(norm2_gpu_order)(int64_t n, const T *x) { return FN(gpu_order_norm2)(n, x); }

/* ------------------------------------------------------------------ bicg_stab.rs */

static void FN(trace8)(double *trace, int64_t cap, int64_t *cnt, double a0, double a1, T b, T c, T d) {
    if (!trace || *cnt >= cap) return;
    double *t = trace + 8 * (*cnt);
    t[0] = a0; t[1] = a1;
    t[2] = S(re)(b); t[3] = S(im)(b);
    t[4] = S(re)(c); t[5] = S(im)(c);
    t[6] = S(re)(d); t[7] = S(im)(d);
    ++*cnt;
}

/* bicg_stab.rs:35-200 (solve, pc==NULL) and :204-366 (precond_solve, pc!=NULL).
 * work: 7*n scalars (bicg_stab.rs:28).  trace rows: [its, r_norm, rho, alpha, w]. */
int FN(bicgstab)(int64_t size, const int64_t *indptr, const int64_t *indices, const T *data,
                 int parallel, const void *pc, int pc_complex, const T *rhs, int64_t rhs_len, T *x,
                 int64_t x_len, int64_t max_iter, R tol, T *work, int64_t *its_out,
                 R *res_out, double *trace, int64_t trace_cap, int64_t *trace_rows) {
    FN(csr) A = {size, indptr, indices, data, parallel};
    int64_t n = rhs_len, tr = 0;
    *its_out = 0; *res_out = 0.0;
    if (trace_rows) *trace_rows = 0;
    if (n != size) return ORC_INCOMPATIBLE_RHS;          /* :44-48 */
    if (n != x_len) return ORC_INCOMPATIBLE_X;           /* :49-53 */

    R rhs_norm = FN(s_norm2)(n, rhs);                 /* :55 */
    if (rhs_norm <= R_EPS) {                       /* :56-60 */
        for (int64_t i = 0; i < n; ++i) x[i] = S(zero)();
        *its_out = 0; *res_out = rhs_norm;
        return ORC_OK;
    }
    R tol2 = tol * rhs_norm;                        /* :61 */

    /* :64-69 / :234-241 workspace layout */
    T *r = work, *r0 = work + n, *y = work + 2 * n, *p = NULL, *v, *t, *z = NULL;
    if (pc) { p = work + 3 * n; v = work + 4 * n; t = work + 5 * n; z = work + 6 * n; }
    else    { v = work + 3 * n; t = work + 4 * n; }

    FN(mv)(&A, x, r);                                    /* :73 */
    FN(axpy)(n, S(neg)(S(one)()), rhs, r);               /* :75  r = A x - rhs */
    memcpy(r0, r, (size_t)n * sizeof(T));                /* :78 */
    R r0_norm = FN(s_norm2)(n, r0);                   /* :80 */
    if (r0_norm <= tol2) {                               /* :81-83 */
        *its_out = 0; *res_out = r0_norm / rhs_norm;
        return ORC_OK;
    }
    R r0_norm_tol = r0_norm * R_EPS;          /* :84 */
    r0_norm_tol = r0_norm_tol * r0_norm_tol;             /* :85 */

    T rho = S(fromr)(r0_norm * r0_norm);                 /* :88 */
    if (pc) {
        memcpy(p, r, (size_t)n * sizeof(T));             /* :261 */
        FN(pc_apply)(n, pc, pc_complex, p, y);           /* :262 */
    } else {
        memcpy(y, r, (size_t)n * sizeof(T));             /* :91 */
    }
    FN(mv)(&A, y, v);                                    /* :93 / :263 */
    T alpha = S(div)(rho, FN(s_cdot)(n, r0, v));       /* :96 */
    FN(axpy)(n, S(neg)(alpha), v, r);                    /* :100 */
    const T *sz = r;                                     /* \hat s: r (no precond) or z */
    if (pc) { FN(pc_apply)(n, pc, pc_complex, r, z); sz = z; }   /* :273 */
    FN(mv)(&A, sz, t);                                   /* :104 / :275 */
    T tmp = FN(s_cdot)(n, t, t);                       /* :107 */
    T w;
    if (S(re)(tmp) > 0.0) w = S(div)(FN(s_cdot)(n, t, r), tmp);   /* :108-110 */
    else w = S(zero)();                                  /* :112 */
    FN(axpy)(n, S(neg)(alpha), y, x);                    /* :115 */
    FN(axpy)(n, S(neg)(w), sz, x);                       /* :117 / :290 */
    FN(axpy)(n, S(neg)(w), t, r);                        /* :120 */
    FN(trace8)(trace, trace_cap, &tr, 0.0, r0_norm, rho, alpha, w);

    for (int64_t its = 1; its < max_iter; ++its) {       /* :122 */
        R r_norm = FN(s_norm2)(n, r);                 /* :123 */
        if (r_norm <= tol2) {                            /* :124-126 */
            *its_out = its; *res_out = r_norm / rhs_norm;
            if (trace_rows) *trace_rows = tr;
            return ORC_OK;
        }
        T rho_old = rho;                                 /* :127 */
        rho = FN(s_cdot)(n, r0, r);                    /* :128 */
        if (S(abs)(rho) < r0_norm_tol) {                 /* :131 restart */
            FN(mv)(&A, x, r);                            /* :134 */
            FN(axpy)(n, S(neg)(S(one)()), rhs, r);       /* :137 */
            memcpy(r0, r, (size_t)n * sizeof(T));        /* :140 */
            R rn = FN(s_norm2)(n, r);                 /* :142 */
            rho = S(fromr)(rn * rn);                     /* :143 */
            r0_norm_tol = S(re)(rho) * R_EPS * R_EPS;   /* :144 */
        }
        T beta = S(mul)(S(div)(rho, rho_old), S(div)(alpha, w));    /* :146 */
        T *yp = pc ? p : y;
        FN(axpby)(n, S(mul)(S(neg)(beta), w), v, beta, yp);         /* :155 / :324 */
        FN(axpy)(n, S(one)(), r, yp);                               /* :156 / :325 */
        if (pc) FN(pc_apply)(n, pc, pc_complex, p, y);              /* :328 */
        FN(mv)(&A, y, v);                                           /* :160 / :329 */
        tmp = FN(s_cdot)(n, r0, v);                               /* :163 */
        if (S(abs)(tmp) <= 0.0) {                                   /* :164-167 */
            *its_out = its;
            if (trace_rows) *trace_rows = tr;
            return ORC_BREAKDOWN;
        }
        alpha = S(div)(rho, tmp);                                   /* :169 */
        FN(axpy)(n, S(neg)(alpha), v, r);                           /* :172 */
        if (pc) FN(pc_apply)(n, pc, pc_complex, r, z);              /* :343 */
        FN(mv)(&A, sz, t);                                          /* :175 / :344 */
        tmp = FN(s_cdot)(n, t, t);                                /* :178 */
        if (S(re)(tmp) > 0.0) w = S(div)(FN(s_cdot)(n, t, r), tmp);   /* :179-183 */
        else w = S(zero)();
        FN(axpy)(n, S(neg)(alpha), y, x);                           /* :188 */
        FN(axpy)(n, S(neg)(w), sz, x);                              /* :191 / :357 */
        FN(axpy)(n, S(neg)(w), t, r);                               /* :196 */
        FN(trace8)(trace, trace_cap, &tr, (double)its, r_norm, rho, alpha, w);
    }
    *its_out = max_iter;                                            /* :199 */
    if (trace_rows) *trace_rows = tr;
    return ORC_INSUFFICIENT_ITER;
}

/* ------------------------------------------------------------------ minres.rs / cs_minres.rs */

/* minres.rs:31-172 (solve), :178-341 (precond_solve, pc!=NULL), and cs_minres.rs:29-158
 * (saunders!=0: complex-symmetric variant; pc must be NULL).  work: 8*n scalars.
 * trace rows: [its, beta_new, alpha, c, (s, res_norm)]. */
int FN(minres)(int saunders, int64_t size, const int64_t *indptr, const int64_t *indices,
               const T *data, int parallel, const void *pc, int pc_complex, const T *rhs,
               int64_t rhs_len, T *x, int64_t x_len, int64_t max_iter, R tol, T *work,
               int64_t *its_out, R *res_out, double *trace, int64_t trace_cap,
               int64_t *trace_rows) {
    FN(csr) A = {size, indptr, indices, data, parallel};
    int64_t n = rhs_len, trn = 0;
    *its_out = 0; *res_out = 0.0;
    if (trace_rows) *trace_rows = 0;
    if (n != size) return ORC_INCOMPATIBLE_RHS;          /* minres.rs:40-44 */
    if (n != x_len) return ORC_INCOMPATIBLE_X;           /* :45-49 */

    R rhs_norm = FN(s_norm2)(n, rhs);                 /* :51 */
    if (rhs_norm <= R_EPS) {                       /* :52-56 */
        for (int64_t i = 0; i < n; ++i) x[i] = S(zero)();
        *its_out = 0; *res_out = rhs_norm;
        return ORC_OK;
    }
    R threshold = tol * rhs_norm;                   /* :57 */

    T c = S(one)(), c_old = S(one)();                    /* :60-61 */
    R s = 0.0, s_old = 0.0;                         /* :62-63 */
    T eta = S(one)();                                    /* :64 */

    T *v_old = work, *v_new = work + n, *v = work + 2 * n;          /* :68-70 */
    T *p_old = work + 3 * n, *p_oold = work + 4 * n, *p = work + 5 * n;   /* :71-73 */
    T *w = work + 6 * n, *w_new = work + 7 * n;          /* :222-223; cs_minres.rs:72 tvec = 6n */
    T *tvec = work + 6 * n;

    memcpy(v_new, rhs, (size_t)n * sizeof(T));           /* :77 */
    FN(mv)(&A, x, v_old);                                /* :78 */
    FN(axpy)(n, S(neg)(S(one)()), v_old, v_new);         /* :80  v_new = rhs - A x */
    R res_norm = FN(s_norm2)(n, v_new);               /* :81 */
    R beta_new, beta_one;
    if (pc) {
        FN(pc_apply)(n, pc, pc_complex, v_new, w_new);   /* :233 */
        T b2 = FN(s_cdot)(n, v_new, w_new);            /* :235 */
        if (S(re)(b2) < R_EPS || S(im)(b2) > R_EPS * S(re)(b2)) {   /* :236-244 */
            *its_out = 0; *res_out = S(re)(b2);
            return ORC_INVALID_PRECOND;
        }
        beta_new = R_SQRT(S(re)(b2));                      /* :245 */
        beta_one = beta_new;                             /* :246 */
        R ts = (R)1 / beta_new;                      /* :248 */
        FN(rscale)(n, ts, v_new);                        /* :249 */
        FN(rscale)(n, ts, w_new);                        /* :250 */
    } else {
        beta_new = res_norm;                             /* :82 */
        beta_one = beta_new;                             /* :83 */
        FN(rscale)(n, (R)1 / beta_new, v_new);            /* :84 */
    }
    for (int64_t i = 0; i < n; ++i) v[i] = S(zero)();     /* :86 */
    for (int64_t i = 0; i < n; ++i) p_old[i] = S(zero)(); /* :87 */
    for (int64_t i = 0; i < n; ++i) p[i] = S(zero)();     /* :88 */

    for (int64_t its = 0; its < max_iter; ++its) {       /* :90 */
        R beta = beta_new;                          /* :91 */
        T *vt = v_old; v_old = v; v = v_new; v_new = vt; /* :92-96 pointer rotation */
        T alpha;
        const T *q;                                      /* the vector copied into p */
        if (pc) {
            T *wt = w; w = w_new; w_new = wt;            /* :259,264-265 */
            FN(mv)(&A, w, v_new);                        /* :271 mul_vec_dot(w, v_new) */
            alpha = FN(s_cdot)(n, w, v_new);
            q = w;
        } else if (saunders) {
            FN(conj)(n, v, tvec);                        /* cs_minres.rs:99 */
            FN(mv)(&A, tvec, v_new);                     /* cs_minres.rs:101 */
            alpha = FN(s_cdot)(n, v, v_new);           /* cs_minres.rs:103 */
            q = tvec;
        } else {
            FN(mv)(&A, v, v_new);                        /* :116 mul_vec_dot(v, v_new) */
            alpha = FN(s_cdot)(n, v, v_new);
            q = v;
        }
        FN(axpy)(n, S(fromr)(-beta), v_old, v_new);      /* :117 */
        FN(axpy)(n, S(neg)(alpha), v, v_new);            /* :118 */
        if (pc) {
            FN(pc_apply)(n, pc, pc_complex, v_new, w_new);   /* :276 */
            T b2 = FN(s_cdot)(n, v_new, w_new);            /* :278 */
            if (S(re)(b2) < R_EPS || S(im)(b2) > R_EPS * S(re)(b2)) {   /* :279-287 */
                *its_out = its; *res_out = S(re)(b2);
                if (trace_rows) *trace_rows = trn;
                return ORC_INVALID_PRECOND;
            }
            beta_new = R_SQRT(S(re)(b2));                  /* :288 */
            R ts = (R)1 / beta_new;                  /* :289 */
            FN(rscale)(n, ts, v_new);                    /* :290 */
            FN(rscale)(n, ts, w_new);                    /* :291 */
        } else {
            beta_new = FN(s_norm2)(n, v_new);              /* :120 */
            FN(rscale)(n, (R)1 / beta_new, v_new);        /* :121 */
        }

        /* Givens: minres.rs:132-148 ; cs_minres.rs:119-134 adds the conj() calls */
        R r3 = s_old * beta;                        /* :132 */
        T trv = saunders ? S(mulr)(S(conj)(c_old), beta) : S(mulr)(c_old, beta);   /* :133 / cs:120 */
        T r2 = S(add)(S(mulr)(alpha, s), S(mul)(c, trv));                          /* :134 */
        T r1_hat = saunders ? S(sub)(S(mul)(S(conj)(c), alpha), S(mulr)(trv, s))   /* cs:122 */
                            : S(sub)(S(mul)(c, alpha), S(mulr)(trv, s));           /* :136 */
        R r1_inv = (R)1 / R_SQRT(S(sq)(r1_hat) + beta_new * beta_new);           /* :139-140 */
        c_old = c;                                       /* :142 */
        s_old = s;                                       /* :143 */
        c = saunders ? S(mulr)(S(conj)(r1_hat), r1_inv) : S(mulr)(r1_hat, r1_inv); /* :147 / cs:133 */
        s = beta_new * r1_inv;                           /* :148 */

        T *pt = p_oold; p_oold = p_old; p_old = p; p = pt;   /* :151-154 */
        memcpy(p, q, (size_t)n * sizeof(T));             /* :156 / :325 / cs:142 */
        FN(axpy)(n, S(neg)(r2), p_old, p);               /* :158 */
        FN(axpy)(n, S(fromr)(-r3), p_oold, p);           /* :159 */
        FN(rscale)(n, r1_inv, p);                        /* :160 */
        FN(axpy)(n, S(mulr)(S(mul)(c, eta), beta_one), p, x);   /* :162 */

        res_norm *= R_ABS(s);                             /* :164 */
        FN(trace8)(trace, trace_cap, &trn, (double)its, beta_new, alpha, c, S(fromr)(s));
        if (trace && trn <= trace_cap && trn > 0) trace[8 * (trn - 1) + 7] = res_norm;
        if (res_norm < threshold) {                      /* :165-167 */
            *its_out = its; *res_out = res_norm / rhs_norm;
            if (trace_rows) *trace_rows = trn;
            return ORC_OK;
        }
        eta = S(mulr)(eta, -s);                          /* :168 */
    }
    *its_out = max_iter;                                 /* :171 */
    if (trace_rows) *trace_rows = trn;
    return ORC_INSUFFICIENT_ITER;
}


/* ------------------------------------------------------------------ gauss_seidel.rs:33-140
 * (the reference bounds T: PartialOrd, i.e. real scalars; instantiated for all T here, used for f64/f32)
 * work: 2*n scalars (:29).  Returns ORC_ZERO_DIAG with *its_out = row for ZeorDiagonalElem(row). */
#if !SFX_IS_COMPLEX
int FN(gauss_seidel)(int64_t size, const int64_t *indptr, const int64_t *indices, const T *data, const T *rhs,
                     int64_t rhs_len, T *x, int64_t x_len, int64_t max_iter, R eps, T *work, int64_t *its_out,
                     R *res_out) {
    FN(csr) A = {size, indptr, indices, data, 0};
    *its_out = 0; *res_out = 0.0;
    if (rhs_len != size) return ORC_INCOMPATIBLE_RHS;            /* :41-45 */
    if (rhs_len != x_len) return ORC_INCOMPATIBLE_X;             /* :46-50 */
    if (max_iter == 0) { *its_out = 0; return ORC_INSUFFICIENT_ITER; }   /* :52-54 */
    const int64_t n = rhs_len;
    R b_norm = 0;                                                /* :57 */
    for (int64_t row = 0; row < n; ++row) {                      /* :60 unrolled first sweep */
        T sigma = S(zero)();
        int have_diag = 0; T diag = S(zero)();
        for (int64_t k = indptr[row]; k < indptr[row + 1]; ++k) {
            if (indices[k] != row) sigma = S(add)(sigma, S(mul)(data[k], x[indices[k]]));   /* :66 */
            else { diag = data[k]; have_diag = 1; }              /* :69 */
        }
        if (!have_diag) { *its_out = row; return ORC_ZERO_DIAG; }            /* :72-74 */
        if (S(sq)(diag) < R_EPS) { *its_out = row; return ORC_ZERO_DIAG; }   /* :76-78 */
        work[n + row] = diag;                                    /* :81 */
        b_norm = b_norm + S(sq)(rhs[row]);                       /* :83 */
        x[row] = S(div)(S(sub)(rhs[row], sigma), diag);          /* :84 */
    }
    const R tol2 = eps * R_SQRT(b_norm);                         /* :87 */
    FN(mv)(&A, x, work);                                         /* :90 */
    FN(axpy)(n, S(neg)(S(one)()), rhs, work);                    /* :97 */
    R res = FN(s_norm2)(n, work);                                  /* :104 */
    if (res <= tol2) { *its_out = 1; *res_out = res; return ORC_OK; }   /* :106-108 */
    for (int64_t it = 1; it < max_iter; ++it) {                  /* :110 */
        for (int64_t row = 0; row < n; ++row) {
            T sigma = S(zero)();
            for (int64_t k = indptr[row]; k < indptr[row + 1]; ++k)
                if (indices[k] != row) sigma = S(add)(sigma, S(mul)(data[k], x[indices[k]]));   /* :116 */
            x[row] = S(div)(S(sub)(rhs[row], sigma), work[n + row]);         /* :123 */
        }
        FN(mv)(&A, x, work);                                     /* :128 */
        FN(axpy)(n, S(neg)(S(one)()), rhs, work);                /* :131 */
        res = FN(s_norm2)(n, work);                                /* :133 */
        if (res <= tol2) { *its_out = it; *res_out = res; return ORC_OK; }   /* :135-137 */
    }
    *its_out = max_iter;                                         /* :139 */
    return ORC_INSUFFICIENT_ITER;
}
#endif

#undef FN
#undef S
