// Scalar semantics of the reference's `cauchy::Scalar` for f64 and Complex<f64>, usable on
// host and device.  Every operation is a separately rounded binary64 operation: the library
// is compiled with -ffp-contract=off because the Rust reference never fuses a*b+c
// (src/vecalg.rs:556-605, src/mat.rs:100-105).  Complex mul/div follow num-complex 0.3.1
// (naive formulas), abs() is hypot (cauchy 0.3.0) — see SURVEY.md §2 "third-party crates".
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

#define SPRS_HD __host__ __device__ __forceinline__

namespace sprs {

struct alignas(16) cplx {
    double re, im;
};

template <class T> struct is_complex { static constexpr bool value = false; };
template <> struct is_complex<cplx> { static constexpr bool value = true; };

template <class T> SPRS_HD T szero();
template <> SPRS_HD double szero<double>() { return 0.0; }
template <> SPRS_HD cplx szero<cplx>() { return cplx{0.0, 0.0}; }
template <class T> SPRS_HD T sone();
template <> SPRS_HD double sone<double>() { return 1.0; }
template <> SPRS_HD cplx sone<cplx>() { return cplx{1.0, 0.0}; }
template <class T> SPRS_HD T sfromr(double r);
template <> SPRS_HD double sfromr<double>(double r) { return r; }
template <> SPRS_HD cplx sfromr<cplx>(double r) { return cplx{r, 0.0}; }

SPRS_HD double sadd(double a, double b) { return a + b; }
SPRS_HD double ssub(double a, double b) { return a - b; }
SPRS_HD double smul(double a, double b) { return a * b; }
SPRS_HD double sdiv(double a, double b) { return a / b; }
SPRS_HD double sneg(double a) { return -a; }
SPRS_HD double sconj(double a) { return a; }
SPRS_HD double smulr(double a, double r) { return a * r; }
SPRS_HD double sre(double a) { return a; }
SPRS_HD double sim(double) { return 0.0; }
SPRS_HD double ssq(double a) { return a * a; }
SPRS_HD double sabs(double a) { return fabs(a); }

SPRS_HD cplx sadd(cplx a, cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
SPRS_HD cplx ssub(cplx a, cplx b) { return cplx{a.re - b.re, a.im - b.im}; }
SPRS_HD cplx smul(cplx a, cplx b) { return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
SPRS_HD cplx sdiv(cplx a, cplx b) {
    double n = b.re * b.re + b.im * b.im;
    double re = a.re * b.re + a.im * b.im;
    double im = a.im * b.re - a.re * b.im;
    return cplx{re / n, im / n};
}
SPRS_HD cplx sneg(cplx a) { return cplx{-a.re, -a.im}; }
SPRS_HD cplx sconj(cplx a) { return cplx{a.re, -a.im}; }
SPRS_HD cplx smulr(cplx a, double r) { return cplx{a.re * r, a.im * r}; }
SPRS_HD double sre(cplx a) { return a.re; }
SPRS_HD double sim(cplx a) { return a.im; }
SPRS_HD double ssq(cplx a) { return a.re * a.re + a.im * a.im; }
SPRS_HD double sabs(cplx a) { return hypot(a.re, a.im); }

// `T * V` of DiagPrecond<T,V> (src/precond.rs:48-52): V real => mul_real, V complex => complex mul
SPRS_HD double smulv(double a, double v) { return a * v; }
SPRS_HD cplx smulv(cplx a, double v) { return smulr(a, v); }
SPRS_HD cplx smulv(cplx a, cplx v) { return smul(a, v); }

// V::one() / v  (src/precond.rs:22-24)
SPRS_HD double sinv(double v) { return 1.0 / v; }
SPRS_HD cplx sinv(cplx v) { return sdiv(cplx{1.0, 0.0}, v); }

}  // namespace sprs
