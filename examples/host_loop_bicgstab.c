/* north_star's literal structure from plain C: the HOST owns the Krylov recurrence and calls the HIP kernels one
 * reference op at a time through the C ABI — exactly what `BiCGStab::solve` (src/bicg_stab.rs:35-200) would do if its
 * `MatVecMul` / `vecalg` calls were bound to include/sprsolve_hip.h: sprs_mul_vec_dev_d for
 * `A.mul_vec_unchecked`, sprs_axpy_d / sprs_axpby_d / sprs_conj_dot_d / sprs_norm2_d for `vecalg::*`,
 * sprs_memcpy_d2d for `ptr::copy_nonoverlapping`.  Vectors stay in HBM; five scalars per iteration cross PCIe.
 * The same system is then solved by the library's own recurrence (`sprs_bicgstab_solve_d`, device-resident scalars)
 * in its "literal" mode — same kernels in the same order, so iteration count and x must agree BIT FOR BIT — and in
 * the default fused mode (same arithmetic per element, different reduction order: agreement to rounding).
 *   gcc -std=c99 -I include examples/host_loop_bicgstab.c -o host_loop -L sprsolve_amd -l:libsprsolve_hip.so \
 *       -Wl,-rpath,$PWD/sprsolve_amd -Wl,-rpath,/opt/rocm/lib -lm && ./host_loop 96 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sprsolve_hip.h"

static sprs_ctx *ctx = NULL;
#define CHECK(call) do { int st__ = (call); if (st__ != SPRS_OK) { fprintf(stderr, "%s -> %d (%s) %s\n", #call, st__, \
    sprs_status_str(st__), ctx ? sprs_last_error(ctx) : ""); return 100 + st__; } } while (0)

/* src/bicg_stab.rs:35-200, statement for statement; r = A x - b (negative residual), x -= ... */
static int host_bicgstab(const sprs_csr *A, size_t n, const double *d_rhs, double *d_x, double *work /* 5 n */,
                         size_t max_iter, double tol, size_t *its_out, double *res_out) {
    double *r = work, *r0 = work + n, *y = work + 2 * n, *v = work + 3 * n, *t = work + 4 * n;
    double rhs_norm, r0_norm, tmp, tt, tr, rho, rho_old, alpha, w, beta;
    const double eps = 2.220446049250313e-16;
    CHECK(sprs_norm2_d(ctx, n, d_rhs, &rhs_norm));                                   /* :55 */
    if (rhs_norm <= eps) { CHECK(sprs_memset_zero(ctx, d_x, n * sizeof(double))); *its_out = 0; *res_out = rhs_norm; return SPRS_OK; }
    const double tol2 = tol * rhs_norm;
    CHECK(sprs_mul_vec_dev_d(A, d_x, r));                                            /* :73 */
    CHECK(sprs_axpy_d(ctx, n, -1.0, d_rhs, r));                                      /* :75 */
    CHECK(sprs_memcpy_d2d(ctx, r0, r, n * sizeof(double)));                          /* :78 */
    CHECK(sprs_norm2_d(ctx, n, r0, &r0_norm));                                       /* :79 */
    if (r0_norm <= tol2) { *its_out = 0; *res_out = r0_norm / rhs_norm; return SPRS_OK; }
    double r0_norm_tol = r0_norm * eps; r0_norm_tol *= r0_norm_tol;                  /* :84-86 */
    rho = r0_norm * r0_norm;                                                         /* :88 */
    CHECK(sprs_memcpy_d2d(ctx, y, r, n * sizeof(double)));                           /* :91 */
    CHECK(sprs_mul_vec_dev_d(A, y, v));                                              /* :93 */
    CHECK(sprs_conj_dot_d(ctx, n, r0, v, &tmp));                                     /* :96 (no breakdown test here) */
    alpha = rho / tmp;
    CHECK(sprs_axpy_d(ctx, n, -alpha, v, r));                                        /* :101 */
    CHECK(sprs_mul_vec_dev_d(A, r, t));                                              /* :104 */
    CHECK(sprs_conj_dot_d(ctx, n, t, t, &tt));                                       /* :107 */
    if (tt > 0.0) { CHECK(sprs_conj_dot_d(ctx, n, t, r, &tr)); w = tr / tt; } else w = 0.0;
    CHECK(sprs_axpy_d(ctx, n, -alpha, y, d_x));                                      /* :115 */
    CHECK(sprs_axpy_d(ctx, n, -w, r, d_x));                                          /* :117 */
    CHECK(sprs_axpy_d(ctx, n, -w, t, r));                                            /* :120 */
    for (size_t its = 1; its < max_iter; ++its) {                                    /* :122 */
        double r_norm;
        CHECK(sprs_norm2_d(ctx, n, r, &r_norm));                                     /* :123 */
        if (r_norm <= tol2) { *its_out = its; *res_out = r_norm / rhs_norm; return SPRS_OK; }
        rho_old = rho;
        CHECK(sprs_conj_dot_d(ctx, n, r0, r, &rho));                                 /* :128 */
        if (fabs(rho) < r0_norm_tol) {                                               /* :131-145 restart */
            double rn;
            CHECK(sprs_mul_vec_dev_d(A, d_x, r));
            CHECK(sprs_axpy_d(ctx, n, -1.0, d_rhs, r));
            CHECK(sprs_memcpy_d2d(ctx, r0, r, n * sizeof(double)));
            CHECK(sprs_norm2_d(ctx, n, r, &rn));
            rho = rn * rn; r0_norm_tol = rho * eps * eps;
        }
        beta = (rho / rho_old) * (alpha / w);                                        /* :146 */
        CHECK(sprs_axpby_d(ctx, n, -beta * w, v, beta, y));                          /* :155 */
        CHECK(sprs_axpy_d(ctx, n, 1.0, r, y));                                       /* :156 */
        CHECK(sprs_mul_vec_dev_d(A, y, v));                                          /* :160 */
        CHECK(sprs_conj_dot_d(ctx, n, r0, v, &tmp));                                 /* :163 */
        if (fabs(tmp) <= 0.0) { *its_out = its; return SPRS_BREAKDOWN; }             /* :164-167 */
        alpha = rho / tmp;
        CHECK(sprs_axpy_d(ctx, n, -alpha, v, r));                                    /* :172 */
        CHECK(sprs_mul_vec_dev_d(A, r, t));                                          /* :175 */
        CHECK(sprs_conj_dot_d(ctx, n, t, t, &tt));                                   /* :178 */
        if (tt > 0.0) { CHECK(sprs_conj_dot_d(ctx, n, t, r, &tr)); w = tr / tt; } else w = 0.0;
        CHECK(sprs_axpy_d(ctx, n, -alpha, y, d_x));                                  /* :188 */
        CHECK(sprs_axpy_d(ctx, n, -w, r, d_x));                                      /* :191 */
        CHECK(sprs_axpy_d(ctx, n, -w, t, r));                                        /* :196 */
    }
    *its_out = max_iter;
    return SPRS_INSUFFICIENT_ITER;                                                   /* :199 */
}

int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 64;
    const size_t n = (size_t)R * (size_t)R;
    int32_t *rp = malloc(sizeof(int32_t) * (n + 1)), *ci = malloc(sizeof(int32_t) * n * 5);
    double *val = malloc(sizeof(double) * n * 5), *rhs = calloc(n, sizeof(double));
    double *xa = calloc(n, sizeof(double)), *xb = calloc(n, sizeof(double)), *xc = calloc(n, sizeof(double));
    int64_t nnz = 0;
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) {                       /* benches/bicgstab.rs:54-104 */
            const int r = i * R + j;
            rp[r] = (int32_t)nnz;
            if (i == 0 || j == 0 || i == R - 1 || j == R - 1) { ci[nnz] = r; val[nnz++] = 1.0; rhs[r] = (double)(i + j); }
            else {
                const int c[5] = {r - R, r - 1, r, r + 1, r + R};
                const double v[5] = {1.0, 1.0, -4.0, 1.0, 1.0};
                for (int k = 0; k < 5; ++k) { ci[nnz] = c[k]; val[nnz++] = v[k]; }
            }
        }
    rp[n] = (int32_t)nnz;
    sprs_csr *A = NULL; sprs_bicgstab *S = NULL;
    void *d_rhs = NULL, *d_x = NULL, *d_work = NULL;
    CHECK(sprs_ctx_create(0, NULL, &ctx));
    CHECK(sprs_csr_create_d(ctx, (int64_t)n, (int64_t)n, nnz, rp, ci, val, 0, &A));
    CHECK(sprs_malloc(ctx, n * sizeof(double), &d_rhs));
    CHECK(sprs_malloc(ctx, n * sizeof(double), &d_x));
    CHECK(sprs_malloc(ctx, 5 * n * sizeof(double), &d_work));
    CHECK(sprs_memcpy_h2d(ctx, d_rhs, rhs, n * sizeof(double)));
    CHECK(sprs_memset_zero(ctx, d_x, n * sizeof(double)));
    const size_t max_iter = 20000; const double tol = 1e-10;
    size_t its_a = 0, its_b = 0, its_c = 0; double res_a = 0, res_b = 0, res_c = 0;
    /* (a) the host-owned loop */
    int st = host_bicgstab(A, n, d_rhs, d_x, d_work, max_iter, tol, &its_a, &res_a);
    if (st != SPRS_OK) { fprintf(stderr, "host loop: status %d\n", st); return 3; }
    CHECK(sprs_memcpy_d2h(ctx, xa, d_x, n * sizeof(double)));
    /* (b) the library's recurrence, literal mode; (c) default fused mode */
    CHECK(sprs_bicgstab_create_d(A, n, &S));
    CHECK(sprs_solver_set_mode(S, SPRS_SOLVER_BICGSTAB, 1));
    CHECK(sprs_bicgstab_solve_d(S, rhs, n, xb, n, max_iter, tol, &its_b, &res_b));
    CHECK(sprs_solver_set_mode(S, SPRS_SOLVER_BICGSTAB, 0));
    CHECK(sprs_bicgstab_solve_d(S, rhs, n, xc, n, max_iter, tol, &its_c, &res_c));
    double err = 0.0, dfused = 0.0;
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) {
            err = fmax(err, fabs(xa[i * R + j] - (double)(i + j)));
            dfused = fmax(dfused, fabs(xa[i * R + j] - xc[i * R + j]));
        }
    const int same_bits = memcmp(xa, xb, n * sizeof(double)) == 0 && its_a == its_b && res_a == res_b;
    printf("n=%zu host-loop its=%zu res=%.3e | library literal its=%zu (bit-identical: %s) | fused its=%zu max|dx|=%.2e | max_err=%.3e\n",
           n, its_a, res_a, its_b, same_bits ? "yes" : "NO", its_c, dfused, err);
    sprs_bicgstab_destroy(S); sprs_free(ctx, d_rhs); sprs_free(ctx, d_x); sprs_free(ctx, d_work); sprs_csr_destroy(A); sprs_ctx_destroy(ctx);
    free(rp); free(ci); free(val); free(rhs); free(xa); free(xb); free(xc);
    return (same_bits && err < 1e-6 && dfused < 1e-6) ? 0 : 2;
}
