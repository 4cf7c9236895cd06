/* TEST INFRASTRUCTURE ONLY — CPU oracle for the sprsolve Krylov hot path.
 *
 * This is a plain-C restatement of the reference's algorithm (cxzheng/sprsolve v0.1.4,
 * `parallel` feature without `mkl`): CSR/CSC SpMV (src/mat.rs:68-152), the eight BLAS-1
 * fallbacks (src/vecalg.rs:556-605), the Jacobi preconditioner (src/precond.rs:20-52) and the
 * BiCGStab / MINRES / CSMINRES recurrences (src/bicg_stab.rs, src/minres.rs, src/cs_minres.rs).
 *
 * It is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing in the product path (sprsolve_amd/, include/) may link, import or call it.
 *
 * Parity pin: the reference itself cannot be built here (Rust nightly + un-vendored git
 * dependencies, no toolchain), so this oracle is pinned against every known-answer test the
 * reference's own test modules hold for this path (src/mat.rs:207-280, src/mkl_mat.rs:341-463,
 * src/vecalg.rs:612-841, and the exact solutions implied by the tests/ directory) — see
 * tests/golden/ and tests/test_oracle_golden.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "scalar.h"

/* status codes — identical to include/sprsolve_hip.h (which maps 1:1 onto
 * SolverError, src/error.rs:7-22) */
#define ORC_OK 0
#define ORC_INCOMPATIBLE_RHS 1
#define ORC_INCOMPATIBLE_X 2
#define ORC_INSUFFICIENT_ITER 3
#define ORC_BREAKDOWN 4
#define ORC_INVALID_PRECOND 5
#define ORC_DIM_MISMATCH 6
#define ORC_ZERO_DIAG 8

#define ORC_CAT2_(a, b) a##b
#define ORC_CAT2(a, b) ORC_CAT2_(a, b)
#define ORC_CAT3_(a, b, c) a##b##c
#define ORC_CAT3(a, b, c) ORC_CAT3_(a, b, c)

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* summation order of the reductions inside the solver recurrences (krylov_tmpl.h): 0 = the reference's serial folds,
 * 1 = the order of libsprsolve_hip's stand-alone reduction kernels with at most `grid` workgroups (ctx knob "grid") */
static int orc_red_mode = 0;
static int64_t orc_red_grid = 512;
void orc_set_reduction_order(int mode, int grid) { orc_red_mode = mode; if (grid > 0) orc_red_grid = grid; }

#define ORC_UNDEF_ALL
/* f64 */
#define T double
#define R double
#define CT orc_c64
#define R_EPS DBL_EPSILON
#define R_SQRT sqrt
#define R_ABS fabs
#define SP d_
#define CSP z_
#define SFX_US _d
#define SFX_IS_COMPLEX 0
#include "krylov_tmpl.h"
#undef T
#undef SP
#undef SFX_US
#undef SFX_IS_COMPLEX
/* Complex<f64> */
#define T orc_c64
#define SP z_
#define SFX_US _z
#define SFX_IS_COMPLEX 1
#include "krylov_tmpl.h"
#undef T
#undef R
#undef CT
#undef R_EPS
#undef R_SQRT
#undef R_ABS
#undef SP
#undef CSP
#undef SFX_US
#undef SFX_IS_COMPLEX
/* f32 (the reference is generic over cauchy::Scalar; its tests use f32/c32 at vecalg.rs:647-658,669-677,771-830) */
#define T float
#define R float
#define CT orc_c32
#define R_EPS FLT_EPSILON
#define R_SQRT sqrtf
#define R_ABS fabsf
#define SP f_
#define CSP c_
#define SFX_US _s
#define SFX_IS_COMPLEX 0
#include "krylov_tmpl.h"
#undef T
#undef SP
#undef SFX_US
#undef SFX_IS_COMPLEX
/* Complex<f32> */
#define T orc_c32
#define SP c_
#define SFX_US _c
#define SFX_IS_COMPLEX 1
#include "krylov_tmpl.h"
#undef T
#undef R
#undef CT
#undef R_EPS
#undef R_SQRT
#undef R_ABS
#undef SP
#undef CSP
#undef SFX_US
#undef SFX_IS_COMPLEX

/* precond.rs:20-29 with V = Complex<f64>: one()/v through the complex division formula */
void orc_diag_inv_complex(int64_t n, const orc_c64 *diag, orc_c64 *dinv) {
    for (int64_t i = 0; i < n; ++i) dinv[i] = z_div(z_one(), diag[i]);
}

/* vecalg.rs:570-575 with S = f64, T = Complex<f64> (`Complex * f64`, used by
 * vecalg.rs:746-757): y += x * a with a real */
void orc_axpy_zd(int64_t n, double a, const orc_c64 *x, orc_c64 *y) {
    for (int64_t i = 0; i < n; ++i) y[i] = z_add(y[i], z_mulr(x[i], a));
}
void orc_diag_inv_complex_f(int64_t n, const orc_c32 *diag, orc_c32 *dinv) {
    for (int64_t i = 0; i < n; ++i) dinv[i] = c_div(c_one(), diag[i]);
}
void orc_axpy_cs(int64_t n, float a, const orc_c32 *x, orc_c32 *y) {
    for (int64_t i = 0; i < n; ++i) y[i] = c_add(y[i], c_mulr(x[i], a));
}
