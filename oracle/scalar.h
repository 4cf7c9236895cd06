/* TEST INFRASTRUCTURE ONLY — CPU oracle for the sprsolve hot path.
 *
 * Scalar semantics of the reference's `cauchy::Scalar` / `num_complex::Complex<f64>`
 * restated in plain C.  Every operation is a separately rounded IEEE-754 binary64
 * operation (the Rust reference never contracts a*b+c into an fma; this file must be
 * compiled with -ffp-contract=off).
 *
 * Reference call sites that define the semantics (cauchy 0.3.0 / num-complex 0.3.1 are
 * un-vendored crates.io dependencies, Cargo.toml:15-18; their arithmetic is definitional):
 *   conj()      vecalg.rs:567,582      square() -> |z|^2   vecalg.rs:603
 *   mul_real()  vecalg.rs:598          abs() -> modulus    bicg_stab.rs:131,164
 *   from_real() bicg_stab.rs:88        re()/im()           bicg_stab.rs:108, minres.rs:237-238
 * Complex mul:  (a+bi)(c+di) = (ac-bd) + (ad+bc)i          (num-complex Mul)
 * Complex div:  n = c*c+d*d; ((ac+bd)/n) + ((bc-ad)/n)i    (num-complex Div, the naive formula)
 * Complex abs:  hypot(re, im)                              (num-complex norm())
 */
#ifndef SPRS_ORACLE_SCALAR_H
#define SPRS_ORACLE_SCALAR_H
#include <math.h>
#include <float.h>

typedef struct { double re, im; } orc_c64;
typedef struct { float re, im; } orc_c32;

/* ---- f64 ---- */
static inline double d_zero(void) { return 0.0; }
static inline double d_one(void) { return 1.0; }
static inline double d_add(double a, double b) { return a + b; }
static inline double d_sub(double a, double b) { return a - b; }
static inline double d_mul(double a, double b) { return a * b; }
static inline double d_div(double a, double b) { return a / b; }
static inline double d_neg(double a) { return -a; }
static inline double d_conj(double a) { return a; }
static inline double d_mulr(double a, double r) { return a * r; }
static inline double d_fromr(double r) { return r; }
static inline double d_re(double a) { return a; }
static inline double d_im(double a) { (void)a; return 0.0; }
static inline double d_sq(double a) { return a * a; }
static inline double d_abs(double a) { return fabs(a); }

/* ---- Complex<f64> ---- */
static inline orc_c64 z_make(double re, double im) { orc_c64 r; r.re = re; r.im = im; return r; }
static inline orc_c64 z_zero(void) { return z_make(0.0, 0.0); }
static inline orc_c64 z_one(void) { return z_make(1.0, 0.0); }
static inline orc_c64 z_add(orc_c64 a, orc_c64 b) { return z_make(a.re + b.re, a.im + b.im); }
static inline orc_c64 z_sub(orc_c64 a, orc_c64 b) { return z_make(a.re - b.re, a.im - b.im); }
static inline orc_c64 z_mul(orc_c64 a, orc_c64 b) {
    return z_make(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
static inline orc_c64 z_div(orc_c64 a, orc_c64 b) {
    double n = b.re * b.re + b.im * b.im;
    double re = a.re * b.re + a.im * b.im;
    double im = a.im * b.re - a.re * b.im;
    return z_make(re / n, im / n);
}
static inline orc_c64 z_neg(orc_c64 a) { return z_make(-a.re, -a.im); }
static inline orc_c64 z_conj(orc_c64 a) { return z_make(a.re, -a.im); }
static inline orc_c64 z_mulr(orc_c64 a, double r) { return z_make(a.re * r, a.im * r); }
static inline orc_c64 z_fromr(double r) { return z_make(r, 0.0); }
static inline double z_re(orc_c64 a) { return a.re; }
static inline double z_im(orc_c64 a) { return a.im; }
static inline double z_sq(orc_c64 a) { return a.re * a.re + a.im * a.im; }
static inline double z_abs(orc_c64 a) { return hypot(a.re, a.im); }

/* ---- f32 ---- */
static inline float f_zero(void) { return 0.0f; }
static inline float f_one(void) { return 1.0f; }
static inline float f_add(float a, float b) { return a + b; }
static inline float f_sub(float a, float b) { return a - b; }
static inline float f_mul(float a, float b) { return a * b; }
static inline float f_div(float a, float b) { return a / b; }
static inline float f_neg(float a) { return -a; }
static inline float f_conj(float a) { return a; }
static inline float f_mulr(float a, float r) { return a * r; }
static inline float f_fromr(float r) { return r; }
static inline float f_re(float a) { return a; }
static inline float f_im(float a) { (void)a; return 0.0f; }
static inline float f_sq(float a) { return a * a; }
static inline float f_abs(float a) { return fabsf(a); }

/* ---- Complex<f32> ---- */
static inline orc_c32 c_make(float re, float im) { orc_c32 r; r.re = re; r.im = im; return r; }
static inline orc_c32 c_zero(void) { return c_make(0.0f, 0.0f); }
static inline orc_c32 c_one(void) { return c_make(1.0f, 0.0f); }
static inline orc_c32 c_add(orc_c32 a, orc_c32 b) { return c_make(a.re + b.re, a.im + b.im); }
static inline orc_c32 c_sub(orc_c32 a, orc_c32 b) { return c_make(a.re - b.re, a.im - b.im); }
static inline orc_c32 c_mul(orc_c32 a, orc_c32 b) {
    return c_make(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
static inline orc_c32 c_div(orc_c32 a, orc_c32 b) {
    float n = b.re * b.re + b.im * b.im;
    float re = a.re * b.re + a.im * b.im;
    float im = a.im * b.re - a.re * b.im;
    return c_make(re / n, im / n);
}
static inline orc_c32 c_neg(orc_c32 a) { return c_make(-a.re, -a.im); }
static inline orc_c32 c_conj(orc_c32 a) { return c_make(a.re, -a.im); }
static inline orc_c32 c_mulr(orc_c32 a, float r) { return c_make(a.re * r, a.im * r); }
static inline orc_c32 c_fromr(float r) { return c_make(r, 0.0f); }
static inline float c_re(orc_c32 a) { return a.re; }
static inline float c_im(orc_c32 a) { return a.im; }
static inline float c_sq(orc_c32 a) { return a.re * a.re + a.im * a.im; }
static inline float c_abs(orc_c32 a) { return hypotf(a.re, a.im); }


#endif
