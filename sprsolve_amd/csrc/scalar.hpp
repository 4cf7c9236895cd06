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

struct alignas(8) cplxf {   // Complex<f32>
    float re, im;
};

template <class T> struct is_complex { static constexpr bool value = false; };
template <> struct is_complex<cplx> { static constexpr bool value = true; };
template <> struct is_complex<cplxf> { static constexpr bool value = true; };

// T::Real of cauchy::Scalar
template <class T> struct real_of { using type = double; };
template <> struct real_of<float> { using type = float; };
template <> struct real_of<cplxf> { using type = float; };
template <class T> using Real = typename real_of<T>::type;

// dtype codes stored in the opaque handles
enum : int { DT_D = 0, DT_Z = 1, DT_S = 2, DT_C = 3 };
template <class T> struct dtype_of;
template <> struct dtype_of<double> { static constexpr int value = DT_D; };
template <> struct dtype_of<cplx> { static constexpr int value = DT_Z; };
template <> struct dtype_of<float> { static constexpr int value = DT_S; };
template <> struct dtype_of<cplxf> { static constexpr int value = DT_C; };
inline size_t dtype_size(int dt) { return dt == DT_D ? 8 : dt == DT_Z ? 16 : dt == DT_S ? 4 : 8; }
inline bool dtype_is_complex(int dt) { return dt == DT_Z || dt == DT_C; }

template <class T> SPRS_HD T szero();
template <> SPRS_HD double szero<double>() { return 0.0; }
template <> SPRS_HD cplx szero<cplx>() { return cplx{0.0, 0.0}; }
template <class T> SPRS_HD T sone();
template <> SPRS_HD double sone<double>() { return 1.0; }
template <> SPRS_HD cplx sone<cplx>() { return cplx{1.0, 0.0}; }
template <> SPRS_HD float szero<float>() { return 0.0f; }
template <> SPRS_HD cplxf szero<cplxf>() { return cplxf{0.0f, 0.0f}; }
template <> SPRS_HD float sone<float>() { return 1.0f; }
template <> SPRS_HD cplxf sone<cplxf>() { return cplxf{1.0f, 0.0f}; }
template <class T> SPRS_HD T sfromr(Real<T> r);
template <> SPRS_HD double sfromr<double>(double r) { return r; }
template <> SPRS_HD cplx sfromr<cplx>(double r) { return cplx{r, 0.0}; }
template <> SPRS_HD float sfromr<float>(float r) { return r; }
template <> SPRS_HD cplxf sfromr<cplxf>(float r) { return cplxf{r, 0.0f}; }

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

// ---- f32 / Complex<f32>: the same definitions in single precision
SPRS_HD float sadd(float a, float b) { return a + b; }
SPRS_HD float ssub(float a, float b) { return a - b; }
SPRS_HD float smul(float a, float b) { return a * b; }
SPRS_HD float sdiv(float a, float b) { return a / b; }
SPRS_HD float sneg(float a) { return -a; }
SPRS_HD float sconj(float a) { return a; }
SPRS_HD float smulr(float a, float r) { return a * r; }
SPRS_HD float sre(float a) { return a; }
SPRS_HD float sim(float) { return 0.0f; }
SPRS_HD float ssq(float a) { return a * a; }
SPRS_HD float sabs(float a) { return fabsf(a); }

SPRS_HD cplxf sadd(cplxf a, cplxf b) { return cplxf{a.re + b.re, a.im + b.im}; }
SPRS_HD cplxf ssub(cplxf a, cplxf b) { return cplxf{a.re - b.re, a.im - b.im}; }
SPRS_HD cplxf smul(cplxf a, cplxf b) { return cplxf{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
SPRS_HD cplxf sdiv(cplxf a, cplxf b) {
    float n = b.re * b.re + b.im * b.im;
    float re = a.re * b.re + a.im * b.im;
    float im = a.im * b.re - a.re * b.im;
    return cplxf{re / n, im / n};
}
SPRS_HD cplxf sneg(cplxf a) { return cplxf{-a.re, -a.im}; }
SPRS_HD cplxf sconj(cplxf a) { return cplxf{a.re, -a.im}; }
SPRS_HD cplxf smulr(cplxf a, float r) { return cplxf{a.re * r, a.im * r}; }
SPRS_HD float sre(cplxf a) { return a.re; }
SPRS_HD float sim(cplxf a) { return a.im; }
SPRS_HD float ssq(cplxf a) { return a.re * a.re + a.im * a.im; }
SPRS_HD float sabs(cplxf a) { return hypotf(a.re, a.im); }

SPRS_HD float smulv(float a, float v) { return a * v; }
SPRS_HD cplxf smulv(cplxf a, float v) { return smulr(a, v); }
SPRS_HD cplxf smulv(cplxf a, cplxf v) { return smul(a, v); }
SPRS_HD float sinv(float v) { return 1.0f / v; }
SPRS_HD cplxf sinv(cplxf v) { return sdiv(cplxf{1.0f, 0.0f}, v); }

SPRS_HD double ssqrt(double a) { return sqrt(a); }
SPRS_HD float ssqrt(float a) { return sqrtf(a); }
template <class R> SPRS_HD R seps();     // T::Real::epsilon()
template <> SPRS_HD double seps<double>() { return 2.220446049250313e-16; }
template <> SPRS_HD float seps<float>() { return 1.1920928955078125e-07f; }

// `T * V` of DiagPrecond<T,V> (src/precond.rs:48-52): V real => mul_real, V complex => complex mul
SPRS_HD double smulv(double a, double v) { return a * v; }
SPRS_HD cplx smulv(cplx a, double v) { return smulr(a, v); }
SPRS_HD cplx smulv(cplx a, cplx v) { return smul(a, v); }

// V::one() / v  (src/precond.rs:22-24)
SPRS_HD double sinv(double v) { return 1.0 / v; }
SPRS_HD cplx sinv(cplx v) { return sdiv(cplx{1.0, 0.0}, v); }

}  // namespace sprs
