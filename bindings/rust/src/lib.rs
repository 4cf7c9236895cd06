//! MI355X (gfx950) backend for sprsolve behind the crate's own `MatVecMul` trait and the
//! `BiCGStab::solve` / `MinRes::solve` / `precond_solve` signatures.
//!
//! UNVERIFIED SOURCE: written against `include/sprsolve_hip.h` without a Rust compiler (none
//! exists in the build environment).  The `extern "C"` block is checked mechanically against the
//! header (`tests/test_rust_binding_signatures.py`: names, arity, pointer depth, const-ness, scalar
//! class, and that every `sys::` symbol used below is declared); the same C ABI is exercised end to
//! end by the Python mirror `sprsolve_amd/` and by two C programs (`examples/*.c`).
//! Structural template: `src/mkl_mat.rs` of the reference (opaque handle created from a
//! `CsMatI<T, I>`, `impl MatVecMul`, `Drop`).
//!
//! Three levels, as in `INTEGRATION.md`:
//!  1. `HipCsr<T>: MatVecMul<T>` over host slices — the literal trait, drop-in for `&CsMat<T>`;
//!  2. `DevVec<T>` + `vecalg::*` + `HipCsr::mul_vec_dev` — vectors resident in HBM, the host owns the
//!     Krylov recurrence and calls one kernel per reference op (`examples/host_loop_bicgstab.rs`);
//!  3. `HipBiCGStab` / `HipMinRes` / `HipCSMinRes` — the reference's `new` / `solve` / `precond_solve`
//!     signatures with the whole recurrence on the device.
#![allow(non_camel_case_types)]

use num_complex::{Complex32, Complex64};
use sprs::{CompressedStorage, CsMatI};
use sprsolve::error::{SolveResult, SolverError};
use sprsolve::MatVecMul;
use std::marker::PhantomData;
use std::os::raw::{c_char, c_int, c_void};
use std::ptr;
use std::sync::Once;

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
    pub const SPRS_INVALID_ARGUMENT: c_int = 7;
    pub const SPRS_ZERO_DIAGONAL: c_int = 8;
    pub const SPRS_NOT_SQUARE: c_int = 9;
    pub const SPRS_NOT_CSR: c_int = 10;

    extern "C" {
        pub fn sprs_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut sprs_ctx) -> c_int;
        pub fn sprs_ctx_destroy(ctx: *mut sprs_ctx) -> c_int;
        pub fn sprs_ctx_sync(ctx: *mut sprs_ctx) -> c_int;
        pub fn sprs_last_error(ctx: *const sprs_ctx) -> *const c_char;

        pub fn sprs_malloc(ctx: *mut sprs_ctx, bytes: usize, dev_out: *mut *mut c_void) -> c_int;
        pub fn sprs_free(ctx: *mut sprs_ctx, dev: *mut c_void) -> c_int;
        pub fn sprs_memcpy_h2d(ctx: *mut sprs_ctx, dev_dst: *mut c_void, host_src: *const c_void, bytes: usize) -> c_int;
        pub fn sprs_memcpy_d2h(ctx: *mut sprs_ctx, host_dst: *mut c_void, dev_src: *const c_void, bytes: usize) -> c_int;
        pub fn sprs_memcpy_d2d(ctx: *mut sprs_ctx, dev_dst: *mut c_void, dev_src: *const c_void, bytes: usize) -> c_int;
        pub fn sprs_memset_zero(ctx: *mut sprs_ctx, dev: *mut c_void, bytes: usize) -> c_int;

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
        pub fn sprs_csr_tile_plan(a: *const sprs_csr, n_tiles: *mut i64, n_tile_blocks: *mut i64, n_other_blocks: *mut i64) -> c_int;
        pub fn sprs_csr_chain_plan(a: *const sprs_csr, n_tiles: *mut i64, n_segments: *mut i64, n_chains: *mut i64, n_other_blocks: *mut i64) -> c_int;

        pub fn sprs_mul_vec_d(a: *const sprs_csr, x: *const f64, x_len: usize, y: *mut f64, y_len: usize) -> c_int;
        pub fn sprs_mul_vec_z(a: *const sprs_csr, x: *const Complex64, x_len: usize, y: *mut Complex64, y_len: usize) -> c_int;
        pub fn sprs_mul_vec_dot_d(a: *const sprs_csr, x: *const f64, x_len: usize, y: *mut f64, y_len: usize, dot: *mut f64) -> c_int;
        pub fn sprs_mul_vec_dot_z(a: *const sprs_csr, x: *const Complex64, x_len: usize, y: *mut Complex64, y_len: usize,
            dot: *mut Complex64) -> c_int;
        pub fn sprs_mul_vec_dev_d(a: *const sprs_csr, x: *const f64, y: *mut f64) -> c_int;
        pub fn sprs_mul_vec_dev_z(a: *const sprs_csr, x: *const Complex64, y: *mut Complex64) -> c_int;
        pub fn sprs_mul_vec_dot_dev_d(a: *const sprs_csr, x: *const f64, y: *mut f64, dot: *mut f64) -> c_int;
        pub fn sprs_mul_vec_dot_dev_z(a: *const sprs_csr, x: *const Complex64, y: *mut Complex64, dot: *mut Complex64) -> c_int;

        // vecalg on device vectors (src/vecalg.rs:24-144)
        pub fn sprs_dot_d(ctx: *mut sprs_ctx, n: usize, x: *const f64, y: *const f64, out: *mut f64) -> c_int;
        pub fn sprs_dot_z(ctx: *mut sprs_ctx, n: usize, x: *const Complex64, y: *const Complex64, out: *mut Complex64) -> c_int;
        pub fn sprs_conj_dot_d(ctx: *mut sprs_ctx, n: usize, x: *const f64, y: *const f64, out: *mut f64) -> c_int;
        pub fn sprs_conj_dot_z(ctx: *mut sprs_ctx, n: usize, x: *const Complex64, y: *const Complex64, out: *mut Complex64) -> c_int;
        pub fn sprs_norm2_d(ctx: *mut sprs_ctx, n: usize, x: *const f64, out: *mut f64) -> c_int;
        pub fn sprs_norm2_z(ctx: *mut sprs_ctx, n: usize, x: *const Complex64, out: *mut f64) -> c_int;
        pub fn sprs_scale_d(ctx: *mut sprs_ctx, n: usize, a: f64, x: *mut f64) -> c_int;
        pub fn sprs_scale_z(ctx: *mut sprs_ctx, n: usize, a: Complex64, x: *mut Complex64) -> c_int;
        pub fn sprs_rscale_d(ctx: *mut sprs_ctx, n: usize, a: f64, x: *mut f64) -> c_int;
        pub fn sprs_rscale_z(ctx: *mut sprs_ctx, n: usize, a: f64, x: *mut Complex64) -> c_int;
        pub fn sprs_conj_d(ctx: *mut sprs_ctx, n: usize, v_in: *const f64, v_out: *mut f64) -> c_int;
        pub fn sprs_conj_z(ctx: *mut sprs_ctx, n: usize, v_in: *const Complex64, v_out: *mut Complex64) -> c_int;
        pub fn sprs_axpy_d(ctx: *mut sprs_ctx, n: usize, a: f64, x: *const f64, y: *mut f64) -> c_int;
        pub fn sprs_axpy_z(ctx: *mut sprs_ctx, n: usize, a: Complex64, x: *const Complex64, y: *mut Complex64) -> c_int;
        pub fn sprs_axpy_zd(ctx: *mut sprs_ctx, n: usize, a: f64, x: *const Complex64, y: *mut Complex64) -> c_int;
        pub fn sprs_axpby_d(ctx: *mut sprs_ctx, n: usize, a: f64, x: *const f64, b: f64, y: *mut f64) -> c_int;
        pub fn sprs_axpby_z(ctx: *mut sprs_ctx, n: usize, a: Complex64, x: *const Complex64, b: Complex64, y: *mut Complex64) -> c_int;

        pub fn sprs_diag_precond_create_d(ctx: *mut sprs_ctx, n: usize, diag: *const f64, out: *mut *mut sprs_diag) -> c_int;
        pub fn sprs_diag_precond_create_zd(ctx: *mut sprs_ctx, n: usize, diag: *const f64, out: *mut *mut sprs_diag) -> c_int;
        pub fn sprs_diag_precond_create_z(ctx: *mut sprs_ctx, n: usize, diag: *const Complex64, out: *mut *mut sprs_diag) -> c_int;
        pub fn sprs_diag_precond_destroy(p: *mut sprs_diag) -> c_int;
        pub fn sprs_diag_mul_vec_d(p: *const sprs_diag, v_in: *const f64, in_len: usize, v_out: *mut f64, out_len: usize) -> c_int;
        pub fn sprs_diag_mul_vec_z(p: *const sprs_diag, v_in: *const Complex64, in_len: usize, v_out: *mut Complex64, out_len: usize) -> c_int;
        pub fn sprs_diag_mul_vec_dev_d(p: *const sprs_diag, v_in: *const f64, v_out: *mut f64) -> c_int;
        pub fn sprs_diag_mul_vec_dev_z(p: *const sprs_diag, v_in: *const Complex64, v_out: *mut Complex64) -> c_int;

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

        pub fn sprs_csminres_create_d(a: *const sprs_csr, size: usize, out: *mut *mut sprs_csminres) -> c_int;
        pub fn sprs_csminres_create_z(a: *const sprs_csr, size: usize, out: *mut *mut sprs_csminres) -> c_int;
        pub fn sprs_csminres_destroy(s: *mut sprs_csminres) -> c_int;
        pub fn sprs_csminres_solve_d(s: *mut sprs_csminres, rhs: *const f64, rhs_len: usize, x: *mut f64,
            x_len: usize, max_iter: usize, tol: f64, its: *mut usize, res: *mut f64) -> c_int;
        pub fn sprs_csminres_solve_z(s: *mut sprs_csminres, rhs: *const Complex64, rhs_len: usize, x: *mut Complex64,
            x_len: usize, max_iter: usize, tol: f64, its: *mut usize, res: *mut f64) -> c_int;

        pub fn sprs_gauss_seidel_create(a: *const sprs_csr, out: *mut *mut sprs_gauss_seidel) -> c_int;
        pub fn sprs_gauss_seidel_destroy(g: *mut sprs_gauss_seidel) -> c_int;
        pub fn sprs_gauss_seidel_solve_d(g: *mut sprs_gauss_seidel, rhs: *const f64, rhs_len: usize, x: *mut f64, x_len: usize,
            max_iter: usize, eps: f64, its: *mut usize, res: *mut f64) -> c_int;

        // ---- f32 / Complex<f32> twins (`_s`, `_c`, `_cs`): the reference is generic over all four `cauchy::Scalar` types and
        // tests f32 / c32 BLAS-1 (src/vecalg.rs:647-658,669-677,771-830); every `T::Real` quantity is `f32` here
        pub fn sprs_csr_create_s(ctx: *mut sprs_ctx, nrows: i64, ncols: i64, nnz: i64, row_ptr: *const i32,
            col_idx: *const i32, val: *const f32, storage_csc: c_int, out: *mut *mut sprs_csr) -> c_int;
        pub fn sprs_csr_create_c(ctx: *mut sprs_ctx, nrows: i64, ncols: i64, nnz: i64, row_ptr: *const i32,
            col_idx: *const i32, val: *const Complex32, storage_csc: c_int, out: *mut *mut sprs_csr) -> c_int;
        pub fn sprs_csr_create_i64_s(ctx: *mut sprs_ctx, nrows: i64, ncols: i64, nnz: i64, row_ptr: *const i64,
            col_idx: *const i64, val: *const f32, storage_csc: c_int, out: *mut *mut sprs_csr) -> c_int;
        pub fn sprs_csr_create_i64_c(ctx: *mut sprs_ctx, nrows: i64, ncols: i64, nnz: i64, row_ptr: *const i64,
            col_idx: *const i64, val: *const Complex32, storage_csc: c_int, out: *mut *mut sprs_csr) -> c_int;
        pub fn sprs_mul_vec_s(a: *const sprs_csr, x: *const f32, x_len: usize, y: *mut f32, y_len: usize) -> c_int;
        pub fn sprs_mul_vec_c(a: *const sprs_csr, x: *const Complex32, x_len: usize, y: *mut Complex32, y_len: usize) -> c_int;
        pub fn sprs_mul_vec_dot_s(a: *const sprs_csr, x: *const f32, x_len: usize, y: *mut f32, y_len: usize, dot: *mut f32) -> c_int;
        pub fn sprs_mul_vec_dot_c(a: *const sprs_csr, x: *const Complex32, x_len: usize, y: *mut Complex32, y_len: usize,
            dot: *mut Complex32) -> c_int;
        pub fn sprs_mul_vec_dev_s(a: *const sprs_csr, x: *const f32, y: *mut f32) -> c_int;
        pub fn sprs_mul_vec_dev_c(a: *const sprs_csr, x: *const Complex32, y: *mut Complex32) -> c_int;
        pub fn sprs_mul_vec_dot_dev_s(a: *const sprs_csr, x: *const f32, y: *mut f32, dot: *mut f32) -> c_int;
        pub fn sprs_mul_vec_dot_dev_c(a: *const sprs_csr, x: *const Complex32, y: *mut Complex32, dot: *mut Complex32) -> c_int;
        pub fn sprs_dot_s(ctx: *mut sprs_ctx, n: usize, x: *const f32, y: *const f32, out: *mut f32) -> c_int;
        pub fn sprs_dot_c(ctx: *mut sprs_ctx, n: usize, x: *const Complex32, y: *const Complex32, out: *mut Complex32) -> c_int;
        pub fn sprs_conj_dot_s(ctx: *mut sprs_ctx, n: usize, x: *const f32, y: *const f32, out: *mut f32) -> c_int;
        pub fn sprs_conj_dot_c(ctx: *mut sprs_ctx, n: usize, x: *const Complex32, y: *const Complex32, out: *mut Complex32) -> c_int;
        pub fn sprs_norm2_s(ctx: *mut sprs_ctx, n: usize, x: *const f32, out: *mut f32) -> c_int;
        pub fn sprs_norm2_c(ctx: *mut sprs_ctx, n: usize, x: *const Complex32, out: *mut f32) -> c_int;
        pub fn sprs_scale_s(ctx: *mut sprs_ctx, n: usize, a: f32, x: *mut f32) -> c_int;
        pub fn sprs_scale_c(ctx: *mut sprs_ctx, n: usize, a: Complex32, x: *mut Complex32) -> c_int;
        pub fn sprs_rscale_s(ctx: *mut sprs_ctx, n: usize, a: f32, x: *mut f32) -> c_int;
        pub fn sprs_rscale_c(ctx: *mut sprs_ctx, n: usize, a: f32, x: *mut Complex32) -> c_int;
        pub fn sprs_conj_s(ctx: *mut sprs_ctx, n: usize, v_in: *const f32, v_out: *mut f32) -> c_int;
        pub fn sprs_conj_c(ctx: *mut sprs_ctx, n: usize, v_in: *const Complex32, v_out: *mut Complex32) -> c_int;
        pub fn sprs_axpy_s(ctx: *mut sprs_ctx, n: usize, a: f32, x: *const f32, y: *mut f32) -> c_int;
        pub fn sprs_axpy_c(ctx: *mut sprs_ctx, n: usize, a: Complex32, x: *const Complex32, y: *mut Complex32) -> c_int;
        pub fn sprs_axpy_cs(ctx: *mut sprs_ctx, n: usize, a: f32, x: *const Complex32, y: *mut Complex32) -> c_int;
        pub fn sprs_axpby_s(ctx: *mut sprs_ctx, n: usize, a: f32, x: *const f32, b: f32, y: *mut f32) -> c_int;
        pub fn sprs_axpby_c(ctx: *mut sprs_ctx, n: usize, a: Complex32, x: *const Complex32, b: Complex32, y: *mut Complex32) -> c_int;
        pub fn sprs_diag_precond_create_s(ctx: *mut sprs_ctx, n: usize, diag: *const f32, out: *mut *mut sprs_diag) -> c_int;
        pub fn sprs_diag_precond_create_cs(ctx: *mut sprs_ctx, n: usize, diag: *const f32, out: *mut *mut sprs_diag) -> c_int;
        pub fn sprs_diag_precond_create_c(ctx: *mut sprs_ctx, n: usize, diag: *const Complex32, out: *mut *mut sprs_diag) -> c_int;
        pub fn sprs_diag_mul_vec_s(p: *const sprs_diag, v_in: *const f32, in_len: usize, v_out: *mut f32, out_len: usize) -> c_int;
        pub fn sprs_diag_mul_vec_c(p: *const sprs_diag, v_in: *const Complex32, in_len: usize, v_out: *mut Complex32, out_len: usize) -> c_int;
        pub fn sprs_diag_mul_vec_dev_s(p: *const sprs_diag, v_in: *const f32, v_out: *mut f32) -> c_int;
        pub fn sprs_diag_mul_vec_dev_c(p: *const sprs_diag, v_in: *const Complex32, v_out: *mut Complex32) -> c_int;
        pub fn sprs_bicgstab_create_s(a: *const sprs_csr, size: usize, out: *mut *mut sprs_bicgstab) -> c_int;
        pub fn sprs_bicgstab_create_c(a: *const sprs_csr, size: usize, out: *mut *mut sprs_bicgstab) -> c_int;
        pub fn sprs_bicgstab_solve_s(s: *mut sprs_bicgstab, rhs: *const f32, rhs_len: usize, x: *mut f32, x_len: usize,
            max_iter: usize, tol: f32, its: *mut usize, res: *mut f32) -> c_int;
        pub fn sprs_bicgstab_solve_c(s: *mut sprs_bicgstab, rhs: *const Complex32, rhs_len: usize, x: *mut Complex32,
            x_len: usize, max_iter: usize, tol: f32, its: *mut usize, res: *mut f32) -> c_int;
        pub fn sprs_bicgstab_precond_solve_s(s: *mut sprs_bicgstab, p: *const sprs_diag, rhs: *const f32, rhs_len: usize,
            x: *mut f32, x_len: usize, max_iter: usize, tol: f32, its: *mut usize, res: *mut f32) -> c_int;
        pub fn sprs_bicgstab_precond_solve_c(s: *mut sprs_bicgstab, p: *const sprs_diag, rhs: *const Complex32,
            rhs_len: usize, x: *mut Complex32, x_len: usize, max_iter: usize, tol: f32, its: *mut usize,
            res: *mut f32) -> c_int;
        pub fn sprs_minres_create_s(a: *const sprs_csr, size: usize, out: *mut *mut sprs_minres) -> c_int;
        pub fn sprs_minres_create_c(a: *const sprs_csr, size: usize, out: *mut *mut sprs_minres) -> c_int;
        pub fn sprs_minres_solve_s(s: *mut sprs_minres, rhs: *const f32, rhs_len: usize, x: *mut f32, x_len: usize,
            max_iter: usize, tol: f32, its: *mut usize, res: *mut f32) -> c_int;
        pub fn sprs_minres_solve_c(s: *mut sprs_minres, rhs: *const Complex32, rhs_len: usize, x: *mut Complex32,
            x_len: usize, max_iter: usize, tol: f32, its: *mut usize, res: *mut f32) -> c_int;
        pub fn sprs_minres_precond_solve_s(s: *mut sprs_minres, p: *const sprs_diag, rhs: *const f32, rhs_len: usize,
            x: *mut f32, x_len: usize, max_iter: usize, tol: f32, its: *mut usize, res: *mut f32) -> c_int;
        pub fn sprs_minres_precond_solve_c(s: *mut sprs_minres, p: *const sprs_diag, rhs: *const Complex32, rhs_len: usize,
            x: *mut Complex32, x_len: usize, max_iter: usize, tol: f32, its: *mut usize, res: *mut f32) -> c_int;
        pub fn sprs_csminres_create_s(a: *const sprs_csr, size: usize, out: *mut *mut sprs_csminres) -> c_int;
        pub fn sprs_csminres_create_c(a: *const sprs_csr, size: usize, out: *mut *mut sprs_csminres) -> c_int;
        pub fn sprs_csminres_solve_s(s: *mut sprs_csminres, rhs: *const f32, rhs_len: usize, x: *mut f32,
            x_len: usize, max_iter: usize, tol: f32, its: *mut usize, res: *mut f32) -> c_int;
        pub fn sprs_csminres_solve_c(s: *mut sprs_csminres, rhs: *const Complex32, rhs_len: usize, x: *mut Complex32,
            x_len: usize, max_iter: usize, tol: f32, its: *mut usize, res: *mut f32) -> c_int;
        pub fn sprs_gauss_seidel_solve_s(g: *mut sprs_gauss_seidel, rhs: *const f32, rhs_len: usize, x: *mut f32, x_len: usize,
            max_iter: usize, eps: f32, its: *mut usize, res: *mut f32) -> c_int;
    }
}

/// One context (GPU 0, a private HIP stream) shared by every handle of the process, so that
/// `HipDiagPrecond::new(diag)` keeps the reference's one-argument signature (src/precond.rs:20).
/// Entry points that share per-context scratch serialise on the context's mutex inside the library
/// (`T: Send + Sync`, bicg_stab.rs:17-18): handles may be used from several threads.
pub fn default_ctx() -> *mut sys::sprs_ctx {
    static INIT: Once = Once::new();
    static mut CTX: *mut sys::sprs_ctx = ptr::null_mut();
    unsafe {
        INIT.call_once(|| {
            let mut c = ptr::null_mut();
            let st = sys::sprs_ctx_create(0, ptr::null_mut(), &mut c);
            if st != sys::SPRS_OK { panic!("sprsolve_hip: no usable GPU (status {})", st); }
            CTX = c;
        });
        CTX
    }
}

/// Waits for everything queued on the context's stream (the `vecalg` updates and `mul_vec_dev` are asynchronous; the
/// reductions, the host-slice entry points and the solves are blocking by themselves).
pub fn sync() { ok_or_panic(unsafe { sys::sprs_ctx_sync(default_ctx()) }); }

/// Text of the last HIP / RCCL failure on the context (empty when there was none).
pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(sys::sprs_last_error(default_ctx())).to_string_lossy().into_owned() }
}

/// Destroys the shared context.  Only after every handle and `DevVec` of the process has been dropped.
pub unsafe fn shutdown(ctx: *mut sys::sprs_ctx) { sys::sprs_ctx_destroy(ctx); }

/// Status -> the reference's `SolveResult` (src/error.rs:7-22).  `R` = `T::Real` (f64 or f32).
fn map_status<R: Copy + std::fmt::Display>(st: c_int, its: usize, res: R) -> SolveResult<(usize, R)> {
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
#[inline]
fn ok_or_panic(st: c_int) {
    if st == sys::SPRS_DIM_MISMATCH { panic!("Dimension mismatch"); } // src/mat.rs:50-52
    assert_eq!(st, sys::SPRS_OK, "sprsolve_hip backend error");
}

/// Index types of `src/mat.rs:196-199`.  i32 goes to the library as it is; the 8-byte types are
/// handed over as i64 arrays (the library range-checks and narrows); u32 is widened.
pub trait HipIndex: sprs::SpIndex {
    /// Calls `f32_(ptr, idx)` or `f64_(ptr, idx)` with arrays the C ABI accepts.
    fn with_arrays<R>(indptr: &[Self], indices: &[Self], f32_: impl FnOnce(*const i32, *const i32) -> R,
                      f64_: impl FnOnce(*const i64, *const i64) -> R) -> R;
}
impl HipIndex for i32 {
    fn with_arrays<R>(p: &[i32], i: &[i32], f32_: impl FnOnce(*const i32, *const i32) -> R,
                      _f64: impl FnOnce(*const i64, *const i64) -> R) -> R { f32_(p.as_ptr(), i.as_ptr()) }
}
macro_rules! impl_index8 {
    ($t:ty) => {
        impl HipIndex for $t {
            fn with_arrays<R>(p: &[$t], i: &[$t], _f32: impl FnOnce(*const i32, *const i32) -> R,
                              f64_: impl FnOnce(*const i64, *const i64) -> R) -> R {
                // same size and alignment; values above i64::MAX cannot index memory and are refused by the range check
                debug_assert_eq!(std::mem::size_of::<$t>(), 8);
                f64_(p.as_ptr() as *const i64, i.as_ptr() as *const i64)
            }
        }
    };
}
impl_index8!(usize); // the reference's default `CsMat<T>` (src/mat.rs:199, benches/bicgstab.rs:54)
impl_index8!(u64);
impl_index8!(i64);
impl HipIndex for u32 {
    fn with_arrays<R>(p: &[u32], i: &[u32], _f32: impl FnOnce(*const i32, *const i32) -> R,
                      f64_: impl FnOnce(*const i64, *const i64) -> R) -> R {
        let p64: Vec<i64> = p.iter().map(|&v| v as i64).collect();
        let i64_: Vec<i64> = i.iter().map(|&v| v as i64).collect();
        f64_(p64.as_ptr(), i64_.as_ptr())
    }
}

/// Scalars the backend implements — all four `cauchy::Scalar` types the reference is generic over (src/mat.rs:12-37;
/// f32 / c32 BLAS-1 tests src/vecalg.rs:647-658,771-830): one C symbol per operation and type (`_d`, `_z`, `_s`, `_c`) — the
/// build's analogue of the reference's TypeId dispatch (src/lib.rs:23-41).  Every `T::Real` quantity (tolerances, norms,
/// residuals, real scale factors) is `Self::Real`: f64 for f64 / Complex64, f32 for f32 / Complex32.
pub trait HipScalar: cauchy::Scalar {
    unsafe fn csr_create_i32(ctx: *mut sys::sprs_ctx, nr: i64, nc: i64, nnz: i64, p: *const i32, i: *const i32,
                             v: *const Self, csc: c_int, out: *mut *mut sys::sprs_csr) -> c_int;
    unsafe fn csr_create_i64(ctx: *mut sys::sprs_ctx, nr: i64, nc: i64, nnz: i64, p: *const i64, i: *const i64,
                             v: *const Self, csc: c_int, out: *mut *mut sys::sprs_csr) -> c_int;
    unsafe fn mul_vec(a: *const sys::sprs_csr, x: &[Self], y: &mut [Self]) -> c_int;
    unsafe fn mul_vec_dot(a: *const sys::sprs_csr, x: &[Self], y: &mut [Self], d: &mut Self) -> c_int;
    unsafe fn mul_vec_dev(a: *const sys::sprs_csr, x: *const Self, y: *mut Self) -> c_int;
    unsafe fn mul_vec_dot_dev(a: *const sys::sprs_csr, x: *const Self, y: *mut Self, d: &mut Self) -> c_int;
    unsafe fn diag_mul_vec(p: *const sys::sprs_diag, x: &[Self], y: &mut [Self]) -> c_int;
    unsafe fn diag_mul_vec_dev(p: *const sys::sprs_diag, x: *const Self, y: *mut Self) -> c_int;
    unsafe fn bicgstab_create(a: *const sys::sprs_csr, n: usize, out: *mut *mut sys::sprs_bicgstab) -> c_int;
    unsafe fn bicgstab_solve(s: *mut sys::sprs_bicgstab, p: *const sys::sprs_diag, rhs: &[Self], x: &mut [Self],
                             max_iter: usize, tol: Self::Real, its: &mut usize, res: &mut Self::Real) -> c_int;
    unsafe fn minres_create(a: *const sys::sprs_csr, n: usize, out: *mut *mut sys::sprs_minres) -> c_int;
    unsafe fn minres_solve(s: *mut sys::sprs_minres, p: *const sys::sprs_diag, rhs: &[Self], x: &mut [Self],
                           max_iter: usize, tol: Self::Real, its: &mut usize, res: &mut Self::Real) -> c_int;
    unsafe fn csminres_create(a: *const sys::sprs_csr, n: usize, out: *mut *mut sys::sprs_csminres) -> c_int;
    unsafe fn csminres_solve(s: *mut sys::sprs_csminres, rhs: &[Self], x: &mut [Self], max_iter: usize, tol: Self::Real,
                             its: &mut usize, res: &mut Self::Real) -> c_int;
    // vecalg on device pointers
    unsafe fn v_dot(c: *mut sys::sprs_ctx, n: usize, x: *const Self, y: *const Self, o: &mut Self) -> c_int;
    unsafe fn v_conj_dot(c: *mut sys::sprs_ctx, n: usize, x: *const Self, y: *const Self, o: &mut Self) -> c_int;
    unsafe fn v_norm2(c: *mut sys::sprs_ctx, n: usize, x: *const Self, o: &mut Self::Real) -> c_int;
    unsafe fn v_scale(c: *mut sys::sprs_ctx, n: usize, a: Self, x: *mut Self) -> c_int;
    unsafe fn v_rscale(c: *mut sys::sprs_ctx, n: usize, a: Self::Real, x: *mut Self) -> c_int;
    unsafe fn v_conj(c: *mut sys::sprs_ctx, n: usize, i: *const Self, o: *mut Self) -> c_int;
    unsafe fn v_axpy(c: *mut sys::sprs_ctx, n: usize, a: Self, x: *const Self, y: *mut Self) -> c_int;
    unsafe fn v_axpy_real(c: *mut sys::sprs_ctx, n: usize, a: Self::Real, x: *const Self, y: *mut Self) -> c_int;
    unsafe fn v_axpby(c: *mut sys::sprs_ctx, n: usize, a: Self, x: *const Self, b: Self, y: *mut Self) -> c_int;
}

macro_rules! impl_scalar {
    ($t:ty, $create:ident, $create64:ident, $mv:ident, $mvd:ident, $mvdev:ident, $mvddev:ident, $dmv:ident, $dmvdev:ident,
     $bc:ident, $bs:ident, $bps:ident, $mc:ident, $ms:ident, $mps:ident, $cc:ident, $cs:ident,
     $dot:ident, $cdot:ident, $nrm:ident, $scale:ident, $rscale:ident, $conj:ident, $axpy:ident, $axpyr:ident, $axpby:ident) => {
        impl HipScalar for $t {
            unsafe fn csr_create_i32(ctx: *mut sys::sprs_ctx, nr: i64, nc: i64, nnz: i64, p: *const i32, i: *const i32,
                                     v: *const Self, csc: c_int, out: *mut *mut sys::sprs_csr) -> c_int {
                sys::$create(ctx, nr, nc, nnz, p, i, v, csc, out)
            }
            unsafe fn csr_create_i64(ctx: *mut sys::sprs_ctx, nr: i64, nc: i64, nnz: i64, p: *const i64, i: *const i64,
                                     v: *const Self, csc: c_int, out: *mut *mut sys::sprs_csr) -> c_int {
                sys::$create64(ctx, nr, nc, nnz, p, i, v, csc, out)
            }
            unsafe fn mul_vec(a: *const sys::sprs_csr, x: &[Self], y: &mut [Self]) -> c_int {
                sys::$mv(a, x.as_ptr(), x.len(), y.as_mut_ptr(), y.len())
            }
            unsafe fn mul_vec_dot(a: *const sys::sprs_csr, x: &[Self], y: &mut [Self], d: &mut Self) -> c_int {
                sys::$mvd(a, x.as_ptr(), x.len(), y.as_mut_ptr(), y.len(), d)
            }
            unsafe fn mul_vec_dev(a: *const sys::sprs_csr, x: *const Self, y: *mut Self) -> c_int { sys::$mvdev(a, x, y) }
            unsafe fn mul_vec_dot_dev(a: *const sys::sprs_csr, x: *const Self, y: *mut Self, d: &mut Self) -> c_int {
                sys::$mvddev(a, x, y, d)
            }
            unsafe fn diag_mul_vec(p: *const sys::sprs_diag, x: &[Self], y: &mut [Self]) -> c_int {
                sys::$dmv(p, x.as_ptr(), x.len(), y.as_mut_ptr(), y.len())
            }
            unsafe fn diag_mul_vec_dev(p: *const sys::sprs_diag, x: *const Self, y: *mut Self) -> c_int { sys::$dmvdev(p, x, y) }
            unsafe fn bicgstab_create(a: *const sys::sprs_csr, n: usize, out: *mut *mut sys::sprs_bicgstab) -> c_int {
                sys::$bc(a, n, out)
            }
            unsafe fn bicgstab_solve(s: *mut sys::sprs_bicgstab, p: *const sys::sprs_diag, rhs: &[Self], x: &mut [Self],
                                     max_iter: usize, tol: Self::Real, its: &mut usize, res: &mut Self::Real) -> c_int {
                if p.is_null() { sys::$bs(s, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(), max_iter, tol, its, res) }
                else { sys::$bps(s, p, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(), max_iter, tol, its, res) }
            }
            unsafe fn minres_create(a: *const sys::sprs_csr, n: usize, out: *mut *mut sys::sprs_minres) -> c_int {
                sys::$mc(a, n, out)
            }
            unsafe fn minres_solve(s: *mut sys::sprs_minres, p: *const sys::sprs_diag, rhs: &[Self], x: &mut [Self],
                                   max_iter: usize, tol: Self::Real, its: &mut usize, res: &mut Self::Real) -> c_int {
                if p.is_null() { sys::$ms(s, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(), max_iter, tol, its, res) }
                else { sys::$mps(s, p, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(), max_iter, tol, its, res) }
            }
            unsafe fn csminres_create(a: *const sys::sprs_csr, n: usize, out: *mut *mut sys::sprs_csminres) -> c_int {
                sys::$cc(a, n, out)
            }
            unsafe fn csminres_solve(s: *mut sys::sprs_csminres, rhs: &[Self], x: &mut [Self], max_iter: usize, tol: Self::Real,
                                     its: &mut usize, res: &mut Self::Real) -> c_int {
                sys::$cs(s, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(), max_iter, tol, its, res)
            }
            unsafe fn v_dot(c: *mut sys::sprs_ctx, n: usize, x: *const Self, y: *const Self, o: &mut Self) -> c_int { sys::$dot(c, n, x, y, o) }
            unsafe fn v_conj_dot(c: *mut sys::sprs_ctx, n: usize, x: *const Self, y: *const Self, o: &mut Self) -> c_int { sys::$cdot(c, n, x, y, o) }
            unsafe fn v_norm2(c: *mut sys::sprs_ctx, n: usize, x: *const Self, o: &mut Self::Real) -> c_int { sys::$nrm(c, n, x, o) }
            unsafe fn v_scale(c: *mut sys::sprs_ctx, n: usize, a: Self, x: *mut Self) -> c_int { sys::$scale(c, n, a, x) }
            unsafe fn v_rscale(c: *mut sys::sprs_ctx, n: usize, a: Self::Real, x: *mut Self) -> c_int { sys::$rscale(c, n, a, x) }
            unsafe fn v_conj(c: *mut sys::sprs_ctx, n: usize, i: *const Self, o: *mut Self) -> c_int { sys::$conj(c, n, i, o) }
            unsafe fn v_axpy(c: *mut sys::sprs_ctx, n: usize, a: Self, x: *const Self, y: *mut Self) -> c_int { sys::$axpy(c, n, a, x, y) }
            unsafe fn v_axpy_real(c: *mut sys::sprs_ctx, n: usize, a: Self::Real, x: *const Self, y: *mut Self) -> c_int { sys::$axpyr(c, n, a, x, y) }
            unsafe fn v_axpby(c: *mut sys::sprs_ctx, n: usize, a: Self, x: *const Self, b: Self, y: *mut Self) -> c_int { sys::$axpby(c, n, a, x, b, y) }
        }
    };
}
impl_scalar!(f64, sprs_csr_create_d, sprs_csr_create_i64_d, sprs_mul_vec_d, sprs_mul_vec_dot_d, sprs_mul_vec_dev_d,
             sprs_mul_vec_dot_dev_d, sprs_diag_mul_vec_d, sprs_diag_mul_vec_dev_d,
             sprs_bicgstab_create_d, sprs_bicgstab_solve_d, sprs_bicgstab_precond_solve_d,
             sprs_minres_create_d, sprs_minres_solve_d, sprs_minres_precond_solve_d, sprs_csminres_create_d, sprs_csminres_solve_d,
             sprs_dot_d, sprs_conj_dot_d, sprs_norm2_d, sprs_scale_d, sprs_rscale_d, sprs_conj_d, sprs_axpy_d, sprs_axpy_d, sprs_axpby_d);
impl_scalar!(Complex64, sprs_csr_create_z, sprs_csr_create_i64_z, sprs_mul_vec_z, sprs_mul_vec_dot_z, sprs_mul_vec_dev_z,
             sprs_mul_vec_dot_dev_z, sprs_diag_mul_vec_z, sprs_diag_mul_vec_dev_z,
             sprs_bicgstab_create_z, sprs_bicgstab_solve_z, sprs_bicgstab_precond_solve_z,
             sprs_minres_create_z, sprs_minres_solve_z, sprs_minres_precond_solve_z, sprs_csminres_create_z, sprs_csminres_solve_z,
             sprs_dot_z, sprs_conj_dot_z, sprs_norm2_z, sprs_scale_z, sprs_rscale_z, sprs_conj_z, sprs_axpy_z, sprs_axpy_zd, sprs_axpby_z);
impl_scalar!(f32, sprs_csr_create_s, sprs_csr_create_i64_s, sprs_mul_vec_s, sprs_mul_vec_dot_s, sprs_mul_vec_dev_s,
             sprs_mul_vec_dot_dev_s, sprs_diag_mul_vec_s, sprs_diag_mul_vec_dev_s,
             sprs_bicgstab_create_s, sprs_bicgstab_solve_s, sprs_bicgstab_precond_solve_s,
             sprs_minres_create_s, sprs_minres_solve_s, sprs_minres_precond_solve_s, sprs_csminres_create_s, sprs_csminres_solve_s,
             sprs_dot_s, sprs_conj_dot_s, sprs_norm2_s, sprs_scale_s, sprs_rscale_s, sprs_conj_s, sprs_axpy_s, sprs_axpy_s, sprs_axpby_s);
impl_scalar!(Complex32, sprs_csr_create_c, sprs_csr_create_i64_c, sprs_mul_vec_c, sprs_mul_vec_dot_c, sprs_mul_vec_dev_c,
             sprs_mul_vec_dot_dev_c, sprs_diag_mul_vec_c, sprs_diag_mul_vec_dev_c,
             sprs_bicgstab_create_c, sprs_bicgstab_solve_c, sprs_bicgstab_precond_solve_c,
             sprs_minres_create_c, sprs_minres_solve_c, sprs_minres_precond_solve_c, sprs_csminres_create_c, sprs_csminres_solve_c,
             sprs_dot_c, sprs_conj_dot_c, sprs_norm2_c, sprs_scale_c, sprs_rscale_c, sprs_conj_c, sprs_axpy_c, sprs_axpy_cs, sprs_axpby_c);

// ------------------------------------------------------------------------------------------------ operator
/// Device-resident CSR operator: the MI355X twin of `MklMat<T>` (src/mkl_mat.rs:15-74).
pub struct HipCsr<T: HipScalar> {
    ctx: *mut sys::sprs_ctx,
    handle: *mut sys::sprs_csr,
    size: (usize, usize),
    _marker: PhantomData<T>,
}

impl<T: HipScalar> HipCsr<T> {
    /// Copies the matrix to HBM once (CSR or CSC; CSC is converted at creation, stable in column
    /// order so every row keeps the reference scatter's summation order).  Any index type of
    /// `src/mat.rs:196-199`: `CsMat<T>` (usize), `CsMatI<T, i32>`, u32, u64.
    pub fn new<I: HipIndex>(m: &CsMatI<T, I>) -> Result<Self, i32> {
        let ctx = default_ctx();
        let mut handle = ptr::null_mut();
        let (nr, nc, nnz) = (m.rows() as i64, m.cols() as i64, m.nnz() as i64);
        let csc = (m.storage() == CompressedStorage::CSC) as c_int;
        let val = m.data().as_ptr();
        let out: *mut *mut sys::sprs_csr = &mut handle;
        let st = I::with_arrays(m.indptr(), m.indices(),
            |p, i| unsafe { T::csr_create_i32(ctx, nr, nc, nnz, p, i, val, csc, out) },
            |p, i| unsafe { T::csr_create_i64(ctx, nr, nc, nnz, p, i, val, csc, out) });
        if st != sys::SPRS_OK { return Err(st); }
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
    /// LDS-window tiles the SpMV of this handle runs through: (tiles, 128-row blocks in them, 128-row blocks walked singly);
    /// zeros for handles without a tile plan.  Backend detail (csrc/spmv_dict.hip, ctx knob "spmv_tile").
    pub fn tile_plan(&self) -> (i64, i64, i64) {
        let (mut t, mut b, mut o) = (0i64, 0i64, 0i64);
        ok_or_panic(unsafe { sys::sprs_csr_tile_plan(self.handle, &mut t, &mut b, &mut o) });
        (t, b, o)
    }
    /// Plane-streaming chains the SpMV of this handle runs through (3-D stencils large enough to fill the chip): (2048-row tiles,
    /// segments, chains, 128-row blocks walked singly); zeros otherwise.  Backend detail (csrc/spmv_chain.hip, ctx knob "spmv_chain").
    pub fn chain_plan(&self) -> (i64, i64, i64, i64) {
        let (mut t, mut s, mut c, mut o) = (0i64, 0i64, 0i64, 0i64);
        ok_or_panic(unsafe { sys::sprs_csr_chain_plan(self.handle, &mut t, &mut s, &mut c, &mut o) });
        (t, s, c, o)
    }
    /// `mul_vec_unchecked` on vectors that live in HBM (nothing crosses PCIe; asynchronous on the context's stream).
    pub fn mul_vec_dev(&self, v_in: &DevVec<T>, v_out: &mut DevVec<T>) {
        assert!(v_in.len() == self.size.1 && v_out.len() == self.size.0, "Dimension mismatch");
        ok_or_panic(unsafe { T::mul_vec_dev(self.handle, v_in.as_ptr(), v_out.as_mut_ptr()) });
    }
    /// `mul_vec_dot_unchecked` on device vectors: y = A x and conj(x) . y from the kernel's fused epilogue (blocking).
    pub fn mul_vec_dot_dev(&self, v_in: &DevVec<T>, v_out: &mut DevVec<T>) -> T {
        assert!(v_in.len() == self.size.1 && v_out.len() == self.size.0, "Dimension mismatch");
        let mut d = T::zero();
        ok_or_panic(unsafe { T::mul_vec_dot_dev(self.handle, v_in.as_ptr(), v_out.as_mut_ptr(), &mut d) });
        d
    }
    pub(crate) fn raw(&self) -> *const sys::sprs_csr { self.handle }
    pub fn ctx(&self) -> *mut sys::sprs_ctx { self.ctx }
}

impl<T: HipScalar> MatVecMul<T> for HipCsr<T> {
    fn mul_vec(&self, v_in: &[T], v_out: &mut [T]) {
        ok_or_panic(unsafe { T::mul_vec(self.handle, v_in, v_out) });
    }
    fn mul_vec_dot(&self, v_in: &[T], v_out: &mut [T]) -> T {
        let mut d = T::zero();
        ok_or_panic(unsafe { T::mul_vec_dot(self.handle, v_in, v_out, &mut d) });
        d
    }
    unsafe fn mul_vec_unchecked(&self, v_in: &[T], v_out: &mut [T]) { self.mul_vec(v_in, v_out) }
    unsafe fn mul_vec_dot_unchecked(&self, v_in: &[T], v_out: &mut [T]) -> T { self.mul_vec_dot(v_in, v_out) }
}

impl<T: HipScalar> Drop for HipCsr<T> {
    fn drop(&mut self) { unsafe { sys::sprs_csr_destroy(self.handle); } } // src/mkl_mat.rs:322-333
}
unsafe impl<T: HipScalar> Send for HipCsr<T> {}
unsafe impl<T: HipScalar> Sync for HipCsr<T> {} // immutable after creation; entry points queue on the context's mutex

// ------------------------------------------------------------------------------------------------ Jacobi
/// `T * V` pairs of `DiagPrecond<T, V>` (src/precond.rs:6-12): V may be real while T is complex.
pub trait HipDiag<V>: HipScalar {
    unsafe fn diag_create(ctx: *mut sys::sprs_ctx, diag: &[V], out: *mut *mut sys::sprs_diag) -> c_int;
}
impl HipDiag<f64> for f64 {
    unsafe fn diag_create(ctx: *mut sys::sprs_ctx, d: &[f64], out: *mut *mut sys::sprs_diag) -> c_int {
        sys::sprs_diag_precond_create_d(ctx, d.len(), d.as_ptr(), out)
    }
}
impl HipDiag<f64> for Complex64 { // DiagPrecond<Complex64, f64>: tests/test_complex_solve.rs:35-88
    unsafe fn diag_create(ctx: *mut sys::sprs_ctx, d: &[f64], out: *mut *mut sys::sprs_diag) -> c_int {
        sys::sprs_diag_precond_create_zd(ctx, d.len(), d.as_ptr(), out)
    }
}
impl HipDiag<Complex64> for Complex64 { // DiagPrecond<Complex64, Complex64>: tests/test_complex_solve2.rs:4-28
    unsafe fn diag_create(ctx: *mut sys::sprs_ctx, d: &[Complex64], out: *mut *mut sys::sprs_diag) -> c_int {
        sys::sprs_diag_precond_create_z(ctx, d.len(), d.as_ptr(), out)
    }
}
impl HipDiag<f32> for f32 {
    unsafe fn diag_create(ctx: *mut sys::sprs_ctx, d: &[f32], out: *mut *mut sys::sprs_diag) -> c_int {
        sys::sprs_diag_precond_create_s(ctx, d.len(), d.as_ptr(), out)
    }
}
impl HipDiag<f32> for Complex32 {
    unsafe fn diag_create(ctx: *mut sys::sprs_ctx, d: &[f32], out: *mut *mut sys::sprs_diag) -> c_int {
        sys::sprs_diag_precond_create_cs(ctx, d.len(), d.as_ptr(), out)
    }
}
impl HipDiag<Complex32> for Complex32 {
    unsafe fn diag_create(ctx: *mut sys::sprs_ctx, d: &[Complex32], out: *mut *mut sys::sprs_diag) -> c_int {
        sys::sprs_diag_precond_create_c(ctx, d.len(), d.as_ptr(), out)
    }
}

/// Jacobi preconditioner resident in HBM: `DiagPrecond<T, V>` (src/precond.rs:6-63).  `new` stores 1 / diag
/// (no zero check, like the reference); `impl MatVecMul<T>` applies it.
pub struct HipDiagPrecond<T: HipDiag<V>, V> { handle: *mut sys::sprs_diag, n: usize, _marker: PhantomData<(T, V)> }
impl<T: HipDiag<V>, V> HipDiagPrecond<T, V> {
    pub fn new(diag: &[V]) -> Self { // src/precond.rs:20-29
        let mut h = ptr::null_mut();
        let st = unsafe { T::diag_create(default_ctx(), diag, &mut h) };
        assert_eq!(st, sys::SPRS_OK, "sprsolve_hip backend error");
        HipDiagPrecond { handle: h, n: diag.len(), _marker: PhantomData }
    }
    pub fn len(&self) -> usize { self.n }
    /// out = in * diag_inv on device vectors (src/precond.rs:48-52)
    pub fn mul_vec_dev(&self, v_in: &DevVec<T>, v_out: &mut DevVec<T>) {
        assert!(v_in.len() == self.n && v_out.len() == self.n, "Dimension mismatch");
        ok_or_panic(unsafe { T::diag_mul_vec_dev(self.handle, v_in.as_ptr(), v_out.as_mut_ptr()) });
    }
    pub(crate) fn raw(&self) -> *const sys::sprs_diag { self.handle }
}
impl<T: HipDiag<V>, V> MatVecMul<T> for HipDiagPrecond<T, V> {
    fn mul_vec(&self, v_in: &[T], v_out: &mut [T]) { ok_or_panic(unsafe { T::diag_mul_vec(self.handle, v_in, v_out) }); }
    unsafe fn mul_vec_unchecked(&self, v_in: &[T], v_out: &mut [T]) { self.mul_vec(v_in, v_out) }
    fn mul_vec_dot(&self, _v_in: &[T], _v_out: &mut [T]) -> T { unimplemented!() }                    // src/precond.rs:55-57
    unsafe fn mul_vec_dot_unchecked(&self, _v_in: &[T], _v_out: &mut [T]) -> T { unimplemented!() }  // src/precond.rs:60-62
}
impl<T: HipDiag<V>, V> Drop for HipDiagPrecond<T, V> { fn drop(&mut self) { unsafe { sys::sprs_diag_precond_destroy(self.handle); } } }

// ------------------------------------------------------------------------------------------------ solvers
/// `BiCGStab` whose whole recurrence runs on the device — same `new` / `solve` / `precond_solve`
/// signatures as src/bicg_stab.rs:25,35-41,204-211; the handle owns its 7n workspace across solves (:19,28).
pub struct HipBiCGStab<'data, T: HipScalar> { _a: &'data HipCsr<T>, handle: *mut sys::sprs_bicgstab }
impl<'data, T: HipScalar> HipBiCGStab<'data, T> {
    pub fn new(a: &'data HipCsr<T>, size: usize) -> Self {
        let mut h = ptr::null_mut();
        ok_or_panic(unsafe { T::bicgstab_create(a.raw(), size, &mut h) });
        HipBiCGStab { _a: a, handle: h }
    }
    pub fn solve(&mut self, rhs: &[T], x: &mut [T], max_iter: usize, tol: T::Real) -> SolveResult<(usize, T::Real)> {
        let (mut its, mut res) = (0usize, <T::Real>::zero());
        let st = unsafe { T::bicgstab_solve(self.handle, ptr::null(), rhs, x, max_iter, tol, &mut its, &mut res) };
        map_status(st, its, res)
    }
    pub fn precond_solve<V>(&mut self, precond: &HipDiagPrecond<T, V>, rhs: &[T], x: &mut [T], max_iter: usize, tol: T::Real)
        -> SolveResult<(usize, T::Real)> where T: HipDiag<V> {
        let (mut its, mut res) = (0usize, <T::Real>::zero());
        let st = unsafe { T::bicgstab_solve(self.handle, precond.raw(), rhs, x, max_iter, tol, &mut its, &mut res) };
        map_status(st, its, res)
    }
}
impl<'data, T: HipScalar> Drop for HipBiCGStab<'data, T> { fn drop(&mut self) { unsafe { sys::sprs_bicgstab_destroy(self.handle); } } }

/// `MinRes` on the device (src/minres.rs:21,31-37,178-185).
pub struct HipMinRes<'data, T: HipScalar> { _a: &'data HipCsr<T>, handle: *mut sys::sprs_minres }
impl<'data, T: HipScalar> HipMinRes<'data, T> {
    pub fn new(a: &'data HipCsr<T>, size: usize) -> Self {
        let mut h = ptr::null_mut();
        ok_or_panic(unsafe { T::minres_create(a.raw(), size, &mut h) });
        HipMinRes { _a: a, handle: h }
    }
    pub fn solve(&mut self, rhs: &[T], x: &mut [T], max_iter: usize, tol: T::Real) -> SolveResult<(usize, T::Real)> {
        let (mut its, mut res) = (0usize, <T::Real>::zero());
        let st = unsafe { T::minres_solve(self.handle, ptr::null(), rhs, x, max_iter, tol, &mut its, &mut res) };
        map_status(st, its, res)
    }
    pub fn precond_solve<V>(&mut self, precond: &HipDiagPrecond<T, V>, rhs: &[T], x: &mut [T], max_iter: usize, tol: T::Real)
        -> SolveResult<(usize, T::Real)> where T: HipDiag<V> {
        let (mut its, mut res) = (0usize, <T::Real>::zero());
        let st = unsafe { T::minres_solve(self.handle, precond.raw(), rhs, x, max_iter, tol, &mut its, &mut res) };
        map_status(st, its, res)
    }
}
impl<'data, T: HipScalar> Drop for HipMinRes<'data, T> { fn drop(&mut self) { unsafe { sys::sprs_minres_destroy(self.handle); } } }

/// `CSMinRes` on the device (src/cs_minres.rs:19,29-35): complex-symmetric systems; for real T it is MINRES.
pub struct HipCSMinRes<'data, T: HipScalar> { _a: &'data HipCsr<T>, handle: *mut sys::sprs_csminres }
impl<'data, T: HipScalar> HipCSMinRes<'data, T> {
    pub fn new(a: &'data HipCsr<T>, size: usize) -> Self {
        let mut h = ptr::null_mut();
        ok_or_panic(unsafe { T::csminres_create(a.raw(), size, &mut h) });
        HipCSMinRes { _a: a, handle: h }
    }
    pub fn solve(&mut self, rhs: &[T], x: &mut [T], max_iter: usize, tol: T::Real) -> SolveResult<(usize, T::Real)> {
        let (mut its, mut res) = (0usize, <T::Real>::zero());
        let st = unsafe { T::csminres_solve(self.handle, rhs, x, max_iter, tol, &mut its, &mut res) };
        map_status(st, its, res)
    }
}
impl<'data, T: HipScalar> Drop for HipCSMinRes<'data, T> { fn drop(&mut self) { unsafe { sys::sprs_csminres_destroy(self.handle); } } }

/// Real scalars the Gauss-Seidel sweep exists for (`_d`, `_s`).
pub trait HipGsScalar: HipScalar<Real = Self> {
    unsafe fn gs_solve(g: *mut sys::sprs_gauss_seidel, rhs: &[Self], x: &mut [Self], max_iter: usize, eps: Self, its: &mut usize,
                       res: &mut Self) -> c_int;
}
impl HipGsScalar for f64 {
    unsafe fn gs_solve(g: *mut sys::sprs_gauss_seidel, rhs: &[f64], x: &mut [f64], max_iter: usize, eps: f64, its: &mut usize,
                       res: &mut f64) -> c_int {
        sys::sprs_gauss_seidel_solve_d(g, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(), max_iter, eps, its, res)
    }
}
impl HipGsScalar for f32 {
    unsafe fn gs_solve(g: *mut sys::sprs_gauss_seidel, rhs: &[f32], x: &mut [f32], max_iter: usize, eps: f32, its: &mut usize,
                       res: &mut f32) -> c_int {
        sys::sprs_gauss_seidel_solve_s(g, rhs.as_ptr(), rhs.len(), x.as_mut_ptr(), x.len(), max_iter, eps, its, res)
    }
}

/// `GaussSeidel` on the device (src/gauss_seidel.rs:8-141): level-scheduled sweeps, iterates bit-identical to
/// the serial sweep.  `solve` returns the ABSOLUTE residual norm like the reference (:107,136).
pub struct HipGaussSeidel<'data, T: HipGsScalar = f64> { _a: &'data HipCsr<T>, handle: *mut sys::sprs_gauss_seidel }
impl<'data, T: HipGsScalar> HipGaussSeidel<'data, T> {
    pub fn new(a: &'data HipCsr<T>) -> SolveResult<Self> {
        let mut h = ptr::null_mut();
        let st = unsafe { sys::sprs_gauss_seidel_create(a.raw(), &mut h) };
        map_status(st, 0, 0.0f64).map(|_| HipGaussSeidel { _a: a, handle: h })
    }
    pub fn solve(&mut self, rhs: &[T], x: &mut [T], max_iter: usize, eps: T) -> SolveResult<(usize, T)> {
        let (mut its, mut res) = (0usize, T::zero());
        let st = unsafe { T::gs_solve(self.handle, rhs, x, max_iter, eps, &mut its, &mut res) };
        map_status(st, its, res)
    }
}
impl<'data, T: HipGsScalar> Drop for HipGaussSeidel<'data, T> { fn drop(&mut self) { unsafe { sys::sprs_gauss_seidel_destroy(self.handle); } } }

// ------------------------------------------------------------------------------------------------ device vectors
/// A vector resident in HBM.  With `vecalg` below and `HipCsr::mul_vec_dev` this is the level at which a
/// Rust host owns the Krylov recurrence (north_star) — see `examples/host_loop_bicgstab.rs`.
pub struct DevVec<T: HipScalar> { ptr: *mut T, n: usize }
impl<T: HipScalar> DevVec<T> {
    /// `vec![T::zero(); n]` in HBM.
    pub fn zeros(n: usize) -> Self {
        let mut p: *mut c_void = ptr::null_mut();
        let bytes = n * std::mem::size_of::<T>();
        ok_or_panic(unsafe { sys::sprs_malloc(default_ctx(), bytes, &mut p) });
        ok_or_panic(unsafe { sys::sprs_memset_zero(default_ctx(), p, bytes) });
        DevVec { ptr: p as *mut T, n }
    }
    pub fn from_slice(v: &[T]) -> Self {
        let d = DevVec::zeros(v.len());
        ok_or_panic(unsafe { sys::sprs_memcpy_h2d(default_ctx(), d.ptr as *mut c_void, v.as_ptr() as *const c_void,
                                                  v.len() * std::mem::size_of::<T>()) });
        d
    }
    pub fn to_vec(&self) -> Vec<T> {
        let mut v = vec![T::zero(); self.n];
        ok_or_panic(unsafe { sys::sprs_memcpy_d2h(default_ctx(), v.as_mut_ptr() as *mut c_void, self.ptr as *const c_void,
                                                  self.n * std::mem::size_of::<T>()) });
        v
    }
    /// `ptr::copy_nonoverlapping(src, self, n)` (bicg_stab.rs:78,91,140)
    pub fn copy_from(&mut self, src: &DevVec<T>) {
        assert_eq!(self.n, src.n);
        ok_or_panic(unsafe { sys::sprs_memcpy_d2d(default_ctx(), self.ptr as *mut c_void, src.ptr as *const c_void,
                                                  self.n * std::mem::size_of::<T>()) });
    }
    /// `iter_mut().for_each(|v| *v = T::zero())` (minres.rs:86-88)
    pub fn fill_zero(&mut self) {
        ok_or_panic(unsafe { sys::sprs_memset_zero(default_ctx(), self.ptr as *mut c_void, self.n * std::mem::size_of::<T>()) });
    }
    pub fn len(&self) -> usize { self.n }
    pub fn as_ptr(&self) -> *const T { self.ptr }
    pub fn as_mut_ptr(&mut self) -> *mut T { self.ptr }
}
impl<T: HipScalar> Drop for DevVec<T> { fn drop(&mut self) { unsafe { sys::sprs_free(default_ctx(), self.ptr as *mut c_void); } } }

/// `sprsolve::vecalg` (src/vecalg.rs:24-144) on device vectors: same names, argument order and conjugation side.
pub mod vecalg {
    use super::*;
    /// sum x_i * y_i, NO conjugate (src/vecalg.rs:24-32)
    pub fn dot<T: HipScalar>(x: &DevVec<T>, y: &DevVec<T>) -> T {
        assert_eq!(x.len(), y.len());
        let mut o = T::zero();
        ok_or_panic(unsafe { T::v_dot(default_ctx(), x.len(), x.as_ptr(), y.as_ptr(), &mut o) });
        o
    }
    /// sum conj(x_i) * y_i — conjugate-linear in the FIRST argument (src/vecalg.rs:51-59)
    pub fn conj_dot<T: HipScalar>(x: &DevVec<T>, y: &DevVec<T>) -> T {
        assert_eq!(x.len(), y.len());
        let mut o = T::zero();
        ok_or_panic(unsafe { T::v_conj_dot(default_ctx(), x.len(), x.as_ptr(), y.as_ptr(), &mut o) });
        o
    }
    /// sqrt(sum |x_i|^2), unscaled (src/vecalg.rs:63-69,601-605)
    pub fn norm2<T: HipScalar>(x: &DevVec<T>) -> T::Real {
        let mut o = <T::Real>::zero();
        ok_or_panic(unsafe { T::v_norm2(default_ctx(), x.len(), x.as_ptr(), &mut o) });
        o
    }
    /// x *= a (src/vecalg.rs:74-80)
    pub fn scale<T: HipScalar>(a: T, x: &mut DevVec<T>) { ok_or_panic(unsafe { T::v_scale(default_ctx(), x.len(), a, x.as_mut_ptr()) }); }
    /// x = x.mul_real(a) (src/vecalg.rs:86-92)
    pub fn rscale<T: HipScalar>(a: T::Real, x: &mut DevVec<T>) { ok_or_panic(unsafe { T::v_rscale(default_ctx(), x.len(), a, x.as_mut_ptr()) }); }
    /// out = conj(in) (src/vecalg.rs:96-104)
    pub fn conj<T: HipScalar>(v_in: &DevVec<T>, v_out: &mut DevVec<T>) {
        assert_eq!(v_in.len(), v_out.len());
        ok_or_panic(unsafe { T::v_conj(default_ctx(), v_in.len(), v_in.as_ptr(), v_out.as_mut_ptr()) });
    }
    /// y += a * x (src/vecalg.rs:109-118)
    pub fn axpy<T: HipScalar>(a: T, x: &DevVec<T>, y: &mut DevVec<T>) {
        assert_eq!(x.len(), y.len());
        ok_or_panic(unsafe { T::v_axpy(default_ctx(), x.len(), a, x.as_ptr(), y.as_mut_ptr()) });
    }
    /// y += a * x with a REAL scalar on a complex vector (`T: Mul<S>`, src/vecalg.rs:746-757)
    pub fn axpy_real<T: HipScalar>(a: T::Real, x: &DevVec<T>, y: &mut DevVec<T>) {
        assert_eq!(x.len(), y.len());
        ok_or_panic(unsafe { T::v_axpy_real(default_ctx(), x.len(), a, x.as_ptr(), y.as_mut_ptr()) });
    }
    /// y = a * x + b * y, in that order (src/vecalg.rs:135-144)
    pub fn axpby<T: HipScalar>(a: T, x: &DevVec<T>, b: T, y: &mut DevVec<T>) {
        assert_eq!(x.len(), y.len());
        ok_or_panic(unsafe { T::v_axpby(default_ctx(), x.len(), a, x.as_ptr(), b, y.as_mut_ptr()) });
    }
}
