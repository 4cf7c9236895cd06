//! MI355X (gfx950) backend for sprsolve behind the crate's own `MatVecMul` trait and the
//! `BiCGStab::solve` / `MinRes::solve` signatures.
//!
//! UNVERIFIED SOURCE: written against `include/sprsolve_hip.h` without a Rust compiler (none
//! exists in the build environment).  The same C ABI is exercised end to end by the Python
//! mirror `sprsolve_amd/`; this file shows the exact binding a sprsolve maintainer would add.
//! Structural template: `src/mkl_mat.rs` of the reference (opaque handle created from a
//! `CsMatI<T, i32>`, `impl MatVecMul`, `Drop`).
#![allow(non_camel_case_types)]

use num_complex::Complex64;
use sprs::{CompressedStorage, CsMatI};
use sprsolve::error::{SolveResult, SolverError};
use sprsolve::MatVecMul;
use std::marker::PhantomData;
use std::os::raw::{c_char, c_int, c_void};
use std::ptr;

pub mod sys {
    use super::*;
    #[repr(C)] pub struct sprs_ctx { _p: [u8; 0] }
    #[repr(C)] pub struct sprs_csr { _p: [u8; 0] }
    #[repr(C)] pub struct sprs_diag { _p: [u8; 0] }
    #[repr(C)] pub struct sprs_bicgstab { _p: [u8; 0] }
    #[repr(C)] pub struct sprs_minres { _p: [u8; 0] }
    #[repr(C)] pub struct sprs_csminres { _p: [u8; 0] }
    #[repr(C)] pub struct sprs_gauss_seidel { _p: [u8; 0] }

    pub const SPRS_OK: c_int = 0;
    pub const SPRS_INCOMPATIBLE_RHS_SIZE: c_int = 1;
    pub const SPRS_INCOMPATIBLE_X_SIZE: c_int = 2;
    pub const SPRS_INSUFFICIENT_ITER: c_int = 3;
    pub const SPRS_BREAKDOWN: c_int = 4;
    pub const SPRS_INVALID_PRECOND: c_int = 5;
    pub const SPRS_DIM_MISMATCH: c_int = 6;
    pub const SPRS_ZERO_DIAGONAL: c_int = 8;
    pub const SPRS_NOT_SQUARE: c_int = 9;
    pub const SPRS_NOT_CSR: c_int = 10;

    extern "C" {
        pub fn sprs_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut sprs_ctx) -> c_int;
        pub fn sprs_ctx_destroy(ctx: *mut sprs_ctx) -> c_int;
        pub fn sprs_last_error(ctx: *const sprs_ctx) -> *const c_char;

        pub fn sprs_csr_create_d(ctx: *mut sprs_ctx, nrows: i64, ncols: i64, nnz: i64, row_ptr: *const i32,
            col_idx: *const i32, val: *const f64, storage_csc: c_int, out: *mut *mut sprs_csr) -> c_int;
        pub fn sprs_csr_create_z(ctx: *mut sprs_ctx, nrows: i64, ncols: i64, nnz: i64, row_ptr: *const i32,
            col_idx: *const i32, val: *const Complex64, storage_csc: c_int, out: *mut *mut sprs_csr) -> c_int;
        pub fn sprs_csr_create_i64_d(ctx: *mut sprs_ctx, nrows: i64, ncols: i64, nnz: i64, row_ptr: *const i64,
            col_idx: *const i64, val: *const f64, storage_csc: c_int, out: *mut *mut sprs_csr) -> c_int;
        pub fn sprs_csr_create_i64_z(ctx: *mut sprs_ctx, nrows: i64, ncols: i64, nnz: i64, row_ptr: *const i64,
            col_idx: *const i64, val: *const Complex64, storage_csc: c_int, out: *mut *mut sprs_csr) -> c_int;
        pub fn sprs_csr_destroy(a: *mut sprs_csr) -> c_int;
        pub fn sprs_csr_stream_format(a: *const sprs_csr, n_offsets: *mut c_int, n_pairs: *mut c_int) -> c_int;

        pub fn sprs_mul_vec_d(a: *const sprs_csr, x: *const f64, x_len: usize, y: *mut f64, y_len: usize) -> c_int;
        pub fn sprs_mul_vec_z(a: *const sprs_csr, x: *const Complex64, x_len: usize, y: *mut Complex64, y_len: usize) -> c_int;
        pub fn sprs_mul_vec_dot_d(a: *const sprs_csr, x: *const f64, x_len: usize, y: *mut f64, y_len: usize, dot: *mut f64) -> c_int;
        pub fn sprs_mul_vec_dot_z(a: *const sprs_csr, x: *const Complex64, x_len: usize, y: *mut Complex64, y_len: usize,
            dot: *mut Complex64) -> c_int;

        pub fn sprs_diag_precond_create_d(ctx: *mut sprs_ctx, n: usize, diag: *const f64, out: *mut *mut sprs_diag) -> c_int;
        pub fn sprs_diag_precond_create_zd(ctx: *mut sprs_ctx, n: usize, diag: *const f64, out: *mut *mut sprs_diag) -> c_int;
        pub fn sprs_diag_precond_create_z(ctx: *mut sprs_ctx, n: usize, diag: *const Complex64, out: *mut *mut sprs_diag) -> c_int;
        pub fn sprs_diag_precond_destroy(p: *mut sprs_diag) -> c_int;

        pub fn sprs_bicgstab_create_d(a: *const sprs_csr, size: usize, out: *mut *mut sprs_bicgstab) -> c_int;
        pub fn sprs_bicgstab_create_z(a: *const sprs_csr, size: usize, out: *mut *mut sprs_bicgstab) -> c_int;
        pub fn sprs_bicgstab_destroy(s: *mut sprs_bicgstab) -> c_int;
        pub fn sprs_bicgstab_solve_d(s: *mut sprs_bicgstab, rhs: *const f64, rhs_len: usize, x: *mut f64, x_len: usize,
            max_iter: usize, tol: f64, its: *mut usize, res: *mut f64) -> c_int;
        pub fn sprs_bicgstab_solve_z(s: *mut sprs_bicgstab, rhs: *const Complex64, rhs_len: usize, x: *mut Complex64,
            x_len: usize, max_iter: usize, tol: f64, its: *mut usize, res: *mut f64) -> c_int;
        pub fn sprs_bicgstab_precond_solve_d(s: *mut sprs_bicgstab, p: *const sprs_diag, rhs: *const f64, rhs_len: usize,
            x: *mut f64, x_len: usize, max_iter: usize, tol: f64, its: *mut usize, res: *mut f64) -> c_int;
        pub fn sprs_bicgstab_precond_solve_z(s: *mut sprs_bicgstab, p: *const sprs_diag, rhs: *const Complex64,
            rhs_len: usize, x: *mut Complex64, x_len: usize, max_iter: usize, tol: f64, its: *mut usize,
            res: *mut f64) -> c_int;

        pub fn sprs_minres_create_d(a: *const sprs_csr, size: usize, out: *mut *mut sprs_minres) -> c_int;
        pub fn sprs_minres_create_z(a: *const sprs_csr, size: usize, out: *mut *mut sprs_minres) -> c_int;
        pub fn sprs_minres_destroy(s: *mut sprs_minres) -> c_int;
        pub fn sprs_minres_solve_d(s: *mut sprs_minres, rhs: *const f64, rhs_len: usize, x: *mut f64, x_len: usize,
            max_iter: usize, tol: f64, its: *mut usize, res: *mut f64) -> c_int;
        pub fn sprs_minres_solve_z(s: *mut sprs_minres, rhs: *const Complex64, rhs_len: usize, x: *mut Complex64,
            x_len: usize, max_iter: usize, tol: f64, its: *mut usize, res: *mut f64) -> c_int;
        pub fn sprs_minres_precond_solve_d(s: *mut sprs_minres, p: *const sprs_diag, rhs: *const f64, rhs_len: usize,
            x: *mut f64, x_len: usize, max_iter: usize, tol: f64, its: *mut usize, res: *mut f64) -> c_int;
        pub fn sprs_minres_precond_solve_z(s: *mut sprs_minres, p: *const sprs_diag, rhs: *const Complex64, rhs_len: usize,
            x: *mut Complex64, x_len: usize, max_iter: usize, tol: f64, its: *mut usize, res: *mut f64) -> c_int;

        pub fn sprs_csminres_create_z(a: *const sprs_csr, size: usize, out: *mut *mut sprs_csminres) -> c_int;
        pub fn sprs_csminres_destroy(s: *mut sprs_csminres) -> c_int;
        pub fn sprs_gauss_seidel_create(a: *const sprs_csr, out: *mut *mut sprs_gauss_seidel) -> c_int;
        pub fn sprs_gauss_seidel_destroy(g: *mut sprs_gauss_seidel) -> c_int;
        pub fn sprs_gauss_seidel_solve_d(g: *mut sprs_gauss_seidel, rhs: *const f64, rhs_len: usize, x: *mut f64, x_len: usize,
            max_iter: usize, eps: f64, its: *mut usize, res: *mut f64) -> c_int;
        pub fn sprs_csminres_solve_z(s: *mut sprs_csminres, rhs: *const Complex64, rhs_len: usize, x: *mut Complex64,
            x_len: usize, max_iter: usize, tol: f64, its: *mut usize, res: *mut f64) -> c_int;
    }
}

/// Status -> the reference's `SolveResult` (src/error.rs:7-22).
fn map_status(st: c_int, its: usize, res: f64) -> SolveResult<(usize, f64)> {
    match st {
        sys::SPRS_OK => Ok((its, res)),
        sys::SPRS_INCOMPATIBLE_RHS_SIZE => Err(SolverError::IncompatibleMatrixFormat(String::from(
            "Input vec dimension doesn't match the matrix size"))),
        sys::SPRS_INCOMPATIBLE_X_SIZE => Err(SolverError::IncompatibleMatrixFormat(String::from(
            "Input and output vec dimension do not match"))),
        sys::SPRS_INSUFFICIENT_ITER => Err(SolverError::InsufficientIterNum(its)),
        sys::SPRS_BREAKDOWN => Err(SolverError::BreakDown(its)),
        sys::SPRS_INVALID_PRECOND => Err(SolverError::InvalidPreconditioner(format!("beta_{} [{}] is not positive", its, res))),
        sys::SPRS_ZERO_DIAGONAL => Err(SolverError::ZeorDiagonalElem(its)),
        sys::SPRS_NOT_SQUARE => Err(SolverError::IncompatibleMatrixFormat(String::from("Not a square matrix"))),
        sys::SPRS_NOT_CSR => Err(SolverError::IncompatibleMatrixFormat(String::from("Not in CSR format"))),
        sys::SPRS_DIM_MISMATCH => panic!("Dimension mismatch"),
        e => panic!("sprsolve_hip backend error {}", e), // HIP / RCCL failure (cf. mkl_mat.rs:188-193)
    }
}

/// Scalars the backend implements.
pub trait HipScalar: cauchy::Scalar<Real = f64> {
    unsafe fn csr_create(ctx: *mut sys::sprs_ctx, m: &CsMatI<Self, i32>, out: *mut *mut sys::sprs_csr) -> c_int;
    unsafe fn mul_vec(a: *const sys::sprs_csr, x: &[Self], y: &mut [Self]) -> c_int;
    unsafe fn mul_vec_dot(a: *const sys::sprs_csr, x: &[Self], y: &mut [Self], d: &mut Self) -> c_int;
}

macro_rules! impl_scalar {
    ($t:ty, $create:ident, $mv:ident, $mvd:ident) => {
        impl HipScalar for $t {
            unsafe fn csr_create(ctx: *mut sys::sprs_ctx, m: &CsMatI<Self, i32>, out: *mut *mut sys::sprs_csr) -> c_int {
                sys::$create(ctx, m.rows() as i64, m.cols() as i64, m.nnz() as i64, m.indptr().as_ptr(),
                    m.indices().as_ptr(), m.data().as_ptr(), (m.storage() == CompressedStorage::CSC) as c_int, out)
            }
            unsafe fn mul_vec(a: *const sys::sprs_csr, x: &[Self], y: &mut [Self]) -> c_int {
                sys::$mv(a, x.as_ptr(), x.len(), y.as_mut_ptr(), y.len())
            }
            unsafe fn mul_vec_dot(a: *const sys::sprs_csr, x: &[Self], y: &mut [Self], d: &mut Self) -> c_int {
                sys::$mvd(a, x.as_ptr(), x.len(), y.as_mut_ptr(), y.len(), d)
            }
        }
    };
}
impl_scalar!(f64, sprs_csr_create_d, sprs_mul_vec_d, sprs_mul_vec_dot_d);
impl_scalar!(Complex64, sprs_csr_create_z, sprs_mul_vec_z, sprs_mul_vec_dot_z);

/// Device-resident CSR operator: the MI355X twin of `MklMat<T>` (src/mkl_mat.rs:15-74).
pub struct HipCsr<T: HipScalar> {
    ctx: *mut sys::sprs_ctx,
    handle: *mut sys::sprs_csr,
    size: (usize, usize),
    _marker: PhantomData<T>,
}

impl<T: HipScalar> HipCsr<T> {
    /// Copies the matrix to HBM once (CSR or CSC; CSC is converted at creation).
    pub fn new(m: &CsMatI<T, i32>) -> Result<Self, i32> {
        let (mut ctx, mut handle) = (ptr::null_mut(), ptr::null_mut());
        unsafe {
            let st = sys::sprs_ctx_create(0, ptr::null_mut(), &mut ctx);
            if st != sys::SPRS_OK { return Err(st); }
            let st = T::csr_create(ctx, m, &mut handle);
            if st != sys::SPRS_OK { sys::sprs_ctx_destroy(ctx); return Err(st); }
        }
        Ok(HipCsr { ctx, handle, size: (m.rows(), m.cols()), _marker: PhantomData })
    }
    pub fn rows(&self) -> usize { self.size.0 }
    pub fn cols(&self) -> usize { self.size.1 }
    /// Which stream the SpMV reads: (0 plain CSR | 1 offset codes | 2 pair codes, distinct offsets, distinct pairs).
    /// Backend detail (csrc/spmv_dict.hip); y is bit-identical in all three.
    pub fn stream_format(&self) -> (i32, i32, i32) {
        let (mut no, mut np) = (0 as c_int, 0 as c_int);
        let m = unsafe { sys::sprs_csr_stream_format(self.handle, &mut no, &mut np) };
        (m, no, np)
    }
    pub(crate) fn raw(&self) -> *const sys::sprs_csr { self.handle }
    pub(crate) fn ctx(&self) -> *mut sys::sprs_ctx { self.ctx }
}

impl<T: HipScalar> MatVecMul<T> for HipCsr<T> {
    fn mul_vec(&self, v_in: &[T], v_out: &mut [T]) {
        let st = unsafe { T::mul_vec(self.handle, v_in, v_out) };
        if st == sys::SPRS_DIM_MISMATCH { panic!("Dimension mismatch"); } // src/mat.rs:50-52
        assert_eq!(st, sys::SPRS_OK);
    }
    fn mul_vec_dot(&self, v_in: &[T], v_out: &mut [T]) -> T {
        let mut d = T::zero();
        let st = unsafe { T::mul_vec_dot(self.handle, v_in, v_out, &mut d) };
        if st == sys::SPRS_DIM_MISMATCH { panic!("Dimension mismatch"); }
        assert_eq!(st, sys::SPRS_OK);
        d
    }
    unsafe fn mul_vec_unchecked(&self, v_in: &[T], v_out: &mut [T]) { self.mul_vec(v_in, v_out) }
    unsafe fn mul_vec_dot_unchecked(&self, v_in: &[T], v_out: &mut [T]) -> T { self.mul_vec_dot(v_in, v_out) }
}

impl<T: HipScalar> Drop for HipCsr<T> {
    fn drop(&mut self) { // src/mkl_mat.rs:322-333
        unsafe { sys::sprs_csr_destroy(self.handle); sys::sprs_ctx_destroy(self.ctx); }
    }
}
unsafe impl<T: HipScalar> Send for HipCsr<T> {}
unsafe impl<T: HipScalar> Sync for HipCsr<T> {} // immutable after creation; SpMV is safe for shared &self

/// Jacobi preconditioner resident in HBM (`DiagPrecond<f64, f64>`, src/precond.rs).
pub struct HipDiagPrecond { handle: *mut sys::sprs_diag }
impl HipDiagPrecond {
    pub fn new(a: &HipCsr<f64>, diag: &[f64]) -> Result<Self, i32> {
        let mut h = ptr::null_mut();
        let st = unsafe { sys::sprs_diag_precond_create_d(a.ctx(), diag.len(), diag.as_ptr(), &mut h) };
        if st == sys::SPRS_OK { Ok(HipDiagPrecond { handle: h }) } else { Err(st) }
    }
}
impl Drop for HipDiagPrecond { fn drop(&mut self) { unsafe { sys::sprs_diag_precond_destroy(self.handle); } } }

/// `BiCGStab` whose whole recurrence runs on the device — same `new` / `solve` / `precond_solve`
/// signatures as src/bicg_stab.rs:25,35-41,204-211.
pub struct HipBiCGStab<'data> { _a: &'data HipCsr<f64>, handle: *mut sys::sprs_bicgstab }

impl<'data> HipBiCGStab<'data> {
    pub fn new(a: &'data HipCsr<f64>, size: usize) -> Self {
        let mut h = ptr::null_mut();
        let st = unsafe { sys::sprs_bicgstab_create_d(a.raw(), size, &mut h) };
        if st == sys::SPRS_DIM_MISMATCH { panic!("Dimension mismatch"); }
        assert_eq!(st, sys::SPRS_OK);
        HipBiCGStab { _a: a, handle: h }
    }
    pub fn solve(&mut self, rhs: &[f64], x: &mut [f64], max_iter: usize, tol: f64) -> SolveResult<(usize, f64)> {
        let (mut its, mut res) = (0usize, 0f64);
        let st = unsafe { sys::sprs_bicgstab_solve_d(self.handle, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(),
                                                     max_iter, tol, &mut its, &mut res) };
        map_status(st, its, res)
    }
    pub fn precond_solve(&mut self, precond: &HipDiagPrecond, rhs: &[f64], x: &mut [f64], max_iter: usize, tol: f64)
        -> SolveResult<(usize, f64)> {
        let (mut its, mut res) = (0usize, 0f64);
        let st = unsafe { sys::sprs_bicgstab_precond_solve_d(self.handle, precond.handle, rhs.as_ptr(), rhs.len(),
                                                             x.as_mut_ptr(), x.len(), max_iter, tol, &mut its, &mut res) };
        map_status(st, its, res)
    }
}
impl<'data> Drop for HipBiCGStab<'data> { fn drop(&mut self) { unsafe { sys::sprs_bicgstab_destroy(self.handle); } } }

/// `MinRes` on the device (src/minres.rs:21,31-37).
pub struct HipMinRes<'data> { _a: &'data HipCsr<f64>, handle: *mut sys::sprs_minres }
impl<'data> HipMinRes<'data> {
    pub fn new(a: &'data HipCsr<f64>, size: usize) -> Self {
        let mut h = ptr::null_mut();
        assert_eq!(unsafe { sys::sprs_minres_create_d(a.raw(), size, &mut h) }, sys::SPRS_OK);
        HipMinRes { _a: a, handle: h }
    }
    pub fn solve(&mut self, rhs: &[f64], x: &mut [f64], max_iter: usize, tol: f64) -> SolveResult<(usize, f64)> {
        let (mut its, mut res) = (0usize, 0f64);
        let st = unsafe { sys::sprs_minres_solve_d(self.handle, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(),
                                                   max_iter, tol, &mut its, &mut res) };
        map_status(st, its, res)
    }
}
impl<'data> Drop for HipMinRes<'data> { fn drop(&mut self) { unsafe { sys::sprs_minres_destroy(self.handle); } } }

/// `CSMinRes` on the device (src/cs_minres.rs:19,29-35), complex-symmetric systems.
pub struct HipCSMinRes<'data> { _a: &'data HipCsr<Complex64>, handle: *mut sys::sprs_csminres }
impl<'data> HipCSMinRes<'data> {
    pub fn new(a: &'data HipCsr<Complex64>, size: usize) -> Self {
        let mut h = ptr::null_mut();
        assert_eq!(unsafe { sys::sprs_csminres_create_z(a.raw(), size, &mut h) }, sys::SPRS_OK);
        HipCSMinRes { _a: a, handle: h }
    }
    pub fn solve(&mut self, rhs: &[Complex64], x: &mut [Complex64], max_iter: usize, tol: f64)
        -> SolveResult<(usize, f64)> {
        let (mut its, mut res) = (0usize, 0f64);
        let st = unsafe { sys::sprs_csminres_solve_z(self.handle, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(),
                                                     max_iter, tol, &mut its, &mut res) };
        map_status(st, its, res)
    }
}
impl<'data> Drop for HipCSMinRes<'data> { fn drop(&mut self) { unsafe { sys::sprs_csminres_destroy(self.handle); } } }

/// `GaussSeidel` on the device (src/gauss_seidel.rs:8-141): level-scheduled sweeps, iterates bit-identical to
/// the serial sweep.  `solve` returns the ABSOLUTE residual norm like the reference (:107,136).
pub struct HipGaussSeidel<'data> { _a: &'data HipCsr<f64>, handle: *mut sys::sprs_gauss_seidel }
impl<'data> HipGaussSeidel<'data> {
    pub fn new(a: &'data HipCsr<f64>) -> SolveResult<Self> {
        let mut h = ptr::null_mut();
        let st = unsafe { sys::sprs_gauss_seidel_create(a.raw(), &mut h) };
        map_status(st, 0, 0.0).map(|_| HipGaussSeidel { _a: a, handle: h })
    }
    pub fn solve(&mut self, rhs: &[f64], x: &mut [f64], max_iter: usize, eps: f64) -> SolveResult<(usize, f64)> {
        let (mut its, mut res) = (0usize, 0f64);
        let st = unsafe { sys::sprs_gauss_seidel_solve_d(self.handle, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(),
                                                         max_iter, eps, &mut its, &mut res) };
        map_status(st, its, res)
    }
}
impl<'data> Drop for HipGaussSeidel<'data> { fn drop(&mut self) { unsafe { sys::sprs_gauss_seidel_destroy(self.handle); } } }
