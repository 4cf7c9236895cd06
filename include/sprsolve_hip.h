/* sprsolve_hip.h — C ABI of the MI355X (gfx950) backend for sprsolve's Krylov hot path.
 *
 * This is the drop-in boundary.  It exports exactly what a Rust FFI for the
 * path would bind: an opaque device CSR operator that implements the `MatVecMul` trait
 * (reference src/mat.rs:12-37; in-tree precedent for an opaque backend handle:
 * src/mkl_mat.rs:15-149,322-333), the BLAS-1 kernels of src/vecalg.rs on device vectors, the
 * Jacobi preconditioner of src/precond.rs, and the three solver objects with the reference's
 * `new` / `solve` / `precond_solve` signatures (src/bicg_stab.rs:25,35,204; src/minres.rs:21,31,
 * 178; src/cs_minres.rs:19,29).  Plain pointers and sizes only; nothing throws or aborts
 * across this boundary.  The reference-side binding is shown in INTEGRATION.md.
 *
 * Suffix convention:  _d = f64,  _z = Complex<f64> (layout {re, im} = num_complex::Complex<f64>
 * repr(C) = double2),  _zd = complex vector with a real scalar/diagonal;  _s / _c / _cs are the same
 * for f32 / Complex<f32> (declared in one block further down).
 * "host" pointers are ordinary CPU memory; "dev" pointers are HIP device memory on the
 * context's GPU.  All calls are blocking from the caller's view unless stated otherwise.
 */
#ifndef SPRSOLVE_HIP_H
#define SPRSOLVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } sprs_c64;   /* Complex<f64> */
typedef struct { float re, im; } sprs_c32;    /* Complex<f32> */

/* Status codes. 1..5 map 1:1 onto reference src/error.rs:7-22 `SolverError`; 6 is the
 * `panic!("Dimension mismatch")` of src/mat.rs:50-52,58-60 and src/precond.rs:39-41. */
enum {
    SPRS_OK = 0,
    SPRS_INCOMPATIBLE_RHS_SIZE = 1, /* IncompatibleMatrixFormat("Input vec dimension doesn't match the matrix size") bicg_stab.rs:44-48 */
    SPRS_INCOMPATIBLE_X_SIZE = 2,   /* IncompatibleMatrixFormat("Input and output vec dimension do not match")       bicg_stab.rs:49-53 */
    SPRS_INSUFFICIENT_ITER = 3,     /* InsufficientIterNum(max_iter)  bicg_stab.rs:199  (*its_out = max_iter) */
    SPRS_BREAKDOWN = 4,             /* BreakDown(its)                 bicg_stab.rs:164-167 (*its_out = its) */
    SPRS_INVALID_PRECOND = 5,       /* InvalidPreconditioner(..)      minres.rs:236-244,279-287 (*its_out = its, *res_out = re(beta^2)) */
    SPRS_DIM_MISMATCH = 6,          /* panic!("Dimension mismatch")   mat.rs:50-52 */
    SPRS_INVALID_ARGUMENT = 7,      /* null handle, index out of i32 range, malformed CSR */
    SPRS_ZERO_DIAGONAL = 8,         /* ZeorDiagonalElem(row)          gauss_seidel.rs:72-78 (*its_out = row) */
    SPRS_NOT_SQUARE = 9,            /* IncompatibleMatrixFormat("Not a square matrix")  gauss_seidel.rs:16-20 */
    SPRS_NOT_CSR = 10,              /* IncompatibleMatrixFormat("Not in CSR format")    gauss_seidel.rs:22-26 */
    SPRS_ERR_HIP = 100,             /* a HIP runtime call failed; see sprs_last_error() */
    SPRS_ERR_RCCL = 101,            /* an RCCL call failed */
    SPRS_ERR_NO_DEVICE = 102        /* no usable gfx950 device */
};

typedef struct sprs_ctx sprs_ctx;           /* one GPU + one HIP stream + reduction scratch */
typedef struct sprs_csr sprs_csr;           /* device CSR operator: impl MatVecMul (mat.rs:47-153) */
typedef struct sprs_diag sprs_diag;         /* DiagPrecond<T,V>              (precond.rs:6-63)  */
typedef struct sprs_bicgstab sprs_bicgstab; /* BiCGStab<T,M>                 (bicg_stab.rs:17-31) */
typedef struct sprs_minres sprs_minres;     /* MinRes<T,M>                   (minres.rs:13-27)    */
typedef struct sprs_csminres sprs_csminres; /* CSMinRes<T,M>                 (cs_minres.rs:11-25) */
typedef struct sprs_comm sprs_comm;         /* RCCL communicator of this rank (multi-GPU section)  */
typedef struct sprs_gauss_seidel sprs_gauss_seidel; /* GaussSeidel<T>        (gauss_seidel.rs:8-31) */

/* ---------------------------------------------------------------- context */
/* device: HIP device ordinal.  stream: an existing hipStream_t to run on (e.g. the caller's
 * framework stream), or NULL to create a private one. */
int sprs_ctx_create(int device, void *stream, sprs_ctx **out);
int sprs_ctx_destroy(sprs_ctx *ctx);               /* NULL is a no-op */
int sprs_ctx_sync(sprs_ctx *ctx);
const char *sprs_last_error(const sprs_ctx *ctx);  /* text of the last SPRS_ERR_* on this ctx */
const char *sprs_status_str(int status);
int sprs_version(void);
/* Tuning knobs.  Defaults are the measured best on MI355X (profiles/r0*_tuning.md); none changes a result except through
 * the summation order of the fused reductions' partials (y of every SpMV is bit-identical under all of them).
 * -1 = automatic where noted.  Knobs marked (creation) are read when a matrix handle is created.
 *   "grid"          workgroups of the streaming (BLAS-1 / fused recurrence) kernels, 8..4096, multiple of 8
 *   "spmv_grid"     workgroups of the SpMV kernels (-1: 4 per CU)
 *   "poll"          iterations between two host looks at the device-side status word (>= 1)
 *   "stream_nt"     fused recurrence kernels read and write their vectors, and the pair-code SpMV writes y, with
 *                   non-temporal accesses: -1 automatic (vectors of 72 MB and more), 0 / 1
 *   "xcd_chunk"     1: one contiguous chunk of row blocks per XCD (-1: automatic — cache-resident matrices only)
 *   "ew_chunk"      fused recurrence kernels walk one contiguous eighth of the vectors per XCD, the eighth whose rows
 *                   that XCD multiplies: -1 automatic (cache-resident matrices whose far band is at most 1/32 of
 *                   the rows), 0 / 1
 *   "spmv_dict"     SpMV stream: -1 auto / 0 plain CSR / 1 offset codes / 2 (offset, value) pair codes
 *                   (sprs_csr_stream_format reports what a handle got)                                   (creation)
 *   "spmv_wide"     f64 pair codes: two rows per lane, 128-row blocks (-1 / 1 on, 0 off)
 *   "spmv_uniform"  blocks whose rows repeat one code sequence are multiplied from that pattern          (creation)
 *   "spmv_triple"   ... and read columns c - 1, c + 1 of a column triple from column c's loads           (creation)
 *   "spmv_seam"     ... and so are blocks that are uniform but for one row, or two adjacent ones, holding only
 *                   part of the pattern or one entry of their own (line seams of truncated / Dirichlet grids) (creation)
 *   "spmv_tile"     f64 compressed streams: runs of 4096 rows of one stencil pattern are multiplied from an x window
 *                   staged in LDS (near columns) + per-row-pair far loads, one launch with the remaining blocks:
 *                   -1 automatic = vectors of 44 MiB and more, 1 = every matrix with such runs, 0 = off
 *                   (sprs_csr_tile_plan reports what a handle got)                       (creation; 0 also at launch)
 *   "spmv_chain"    f64 pair codes, patterns with one far slot a side at -P / +P (3-D stencils): a workgroup walks a
 *                   column of 2048-row tiles plane by plane with the x windows of three consecutive tiles in LDS — no far
 *                   load at all: -1 automatic = wherever "spmv_tile" applies and the chains fill the chip, 1 = wherever
 *                   chains exist, 0 = off (sprs_csr_chain_plan reports what a handle got)  (creation; 0 also at launch)
 *   "spmv_fuse"     BiCGStab, f64, no preconditioner, one GPU, SpMV through chains: the vector updates that produce an SpMV's
 *                   input (r -= alpha v before t = A r; p = (v (-beta w) + p beta) + r before v = A p) are formed inside that
 *                   SpMV — three launches per iteration instead of five, every scalar and element bit-identical; 0 = off (per solve).
 *                   MINRES / CSMINRES, no preconditioner, one GPU, SpMV through the lane-per-row kernel of a compressed stream
 *                   (f64 offset codes, complex offset / pair codes): the third kernel of an iteration (normalisation, Givens
 *                   rotation, p, x, convergence test) is not launched — the next SpMV multiplies by the un-normalised vector
 *                   scaled in its gathers and the element-wise work rides with the next iteration's second kernel: two launches
 *                   instead of three, fewer vector passes, bit-identical
 *   "p2p_allreduce" distributed solves: the scalar hand-offs go through peer-to-peer mailboxes (no stream operation) instead of
 *                   ncclAllReduce: -1 / 1 wherever the communicator has them (sprs_comm_p2p), 0 = RCCL
 *                   (communicator creation: 0 sets none up; per solve)
 *   "p2p_timeout_ms" how long a consumer kernel polls its mailbox before the solve fails with SPRS_ERR_RCCL (default 20000)
 *   "spmv_eqrows"   plain CSR: blocks of equal-length rows do not read row_ptr                           (creation)
 *   "spmv_wideload" plain CSR, f64: 16-byte stream loads (4 entries per lane), 3 workgroups per CU on HBM-sized
 *                   matrices; 0 = the kernel with 4- / 8-byte loads                                      (creation)
 *   "spmv_period"   XCD-period walk of the compressed streams' blocks for matrices with a far band (rows r and
 *                   r +- band on one XCD): -1 automatic = the f64 pair-code stream (cfg 5: SpMV -2.3 %),
 *                   1 = the offset-code stream too (measured slower), 0 = off                            (creation)
 *   "halo_overlap"  distributed SpMV: 1 (default) multiplies the interior rows while the halo travels
 *   "gs_graph"      1: Gauss-Seidel replays a sweep's level launches from a hipGraph (default 0)
 * sprs_ctx_get also answers "num_cu" and "device".  Unknown key: SPRS_INVALID_ARGUMENT / -1. */
int sprs_ctx_set(sprs_ctx *ctx, const char *key, int64_t value);
int64_t sprs_ctx_get(const sprs_ctx *ctx, const char *key);

/* device memory for callers that have no HIP of their own (a Rust host) */
int sprs_malloc(sprs_ctx *ctx, size_t bytes, void **dev_out);
int sprs_free(sprs_ctx *ctx, void *dev);
int sprs_memcpy_h2d(sprs_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int sprs_memcpy_d2h(sprs_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
int sprs_memcpy_d2d(sprs_ctx *ctx, void *dev_dst, const void *dev_src, size_t bytes); /* ptr::copy_nonoverlapping, bicg_stab.rs:78 */
int sprs_memset_zero(sprs_ctx *ctx, void *dev, size_t bytes);                          /* iter_mut().for_each(zero), minres.rs:86-88 */

/* ---------------------------------------------------------------- CSR operator (MatVecMul) */
/* Create from host arrays (copied to HBM; the caller keeps its own copy — cf. MklMat::new,
 * mkl_mat.rs:32-74).  Index types of mat.rs:196-199: i32 natively; i64 covers u32/u64/usize by
 * narrowing with a range check (SPRS_INVALID_ARGUMENT if anything exceeds i32).
 * `storage_csc` != 0: the arrays are CSC (mat.rs:130-142); converted to CSR once at creation. */
int sprs_csr_create_d(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t *row_ptr,
                      const int32_t *col_idx, const double *val, int storage_csc, sprs_csr **out);
int sprs_csr_create_z(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t *row_ptr,
                      const int32_t *col_idx, const sprs_c64 *val, int storage_csc, sprs_csr **out);
int sprs_csr_create_i64_d(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t *row_ptr,
                          const int64_t *col_idx, const double *val, int storage_csc, sprs_csr **out);
int sprs_csr_create_i64_z(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t *row_ptr,
                          const int64_t *col_idx, const sprs_c64 *val, int storage_csc, sprs_csr **out);
/* Create from arrays already resident in HBM (CSR, i32).  adopt == 0: copied; adopt != 0: the
 * handle references the caller's arrays, which must outlive it (no copy of multi-GB matrices) and must not
 * be modified while the handle lives: like mkl_sparse_optimize (mkl_mat.rs:81-148), creation analyses the
 * matrix once and keeps what it derived (row blocks, the compressed code stream of sprs_csr_stream_format). */
int sprs_csr_create_dev_d(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t *dev_row_ptr,
                          const int32_t *dev_col_idx, const double *dev_val, int adopt, sprs_csr **out);
int sprs_csr_create_dev_z(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t *dev_row_ptr,
                          const int32_t *dev_col_idx, const sprs_c64 *dev_val, int adopt, sprs_csr **out);
int sprs_csr_destroy(sprs_csr *A); /* Drop, mkl_mat.rs:322-333; NULL is a no-op */
int64_t sprs_csr_rows(const sprs_csr *A);
int64_t sprs_csr_cols(const sprs_csr *A);
int64_t sprs_csr_nnz(const sprs_csr *A);
/* Which stream the SpMV of this handle reads (backend detail, csrc/spmv_dict.hip): 0 = plain CSR (12 B/nnz for
 * f64), 1 = one-byte column-offset codes + the values (9 B/nnz), 2 = one-byte codes of the (offset, value) pairs
 * (1 B/nnz).  The compressed streams are built at creation when the matrix has <= 256 distinct (col - row) offsets
 * (and, for real scalars, <= 256 distinct values and pairs; for complex scalars <= 255 distinct (offset, value) pairs
 * among the entries OFF the diagonal — the offset-0 entries are kept one per row, 16 B per row for Complex<f64>) and
 * the ctx knob "spmv_dict" allows it; y is bit-identical in all three.  Automatic policy: pair codes where they
 * exist; else offset codes for real scalars and the plain stream for complex ones.  n_offsets / n_pairs (may be
 * NULL) receive the table sizes (0 = table absent; complex pair codes count the row-value slot as one pair). */
int sprs_csr_stream_format(const sprs_csr *A, int *n_offsets, int *n_pairs);
/* Diagnostics of the compressed streams: the number of row blocks the SpMV of this handle walks (128-row blocks of
 * the f64 pair-code stream, 64-row blocks of the offset-code stream) and how many of them are "uniform" (all rows
 * repeat one code sequence: multiplied from a scalar pattern, no code bytes and no row_ptr read).  Plain stream: its
 * 64-row blocks and how many of them hold rows of equal length (row extents from the descriptor, no row_ptr read).
 * Copies the descriptors to the host: not for hot paths. */
int sprs_csr_wide_blocks(const sprs_csr *A, int64_t *n_blocks, int64_t *n_uniform);
/* Diagnostics of the f64 pair-code stream's LDS-window tiles (ctx knob "spmv_tile", csrc/spmv_dict.hip): how many tiles
 * the SpMV of this handle multiplies from an x window staged in LDS, how many 128-row blocks they cover, and how many
 * 128-row blocks the same launch walks one by one.  All zero when the handle has no tile plan (other streams, matrices
 * without long runs of one stencil pattern, cache-resident matrices under the automatic policy). */
int sprs_csr_tile_plan(const sprs_csr *A, int64_t *n_tiles, int64_t *n_tile_blocks, int64_t *n_other_blocks);
/* ... and of its plane-streaming chains (ctx knob "spmv_chain", csrc/spmv_chain.hip), which take precedence over the tiles
 * where a handle has both (sprs_csr_tile_plan then reports zeros): chain tiles (2048 rows each), chain segments (the work
 * items, about one per workgroup), chains, and the 128-row blocks the same launch walks one by one.  All zero otherwise. */
int sprs_csr_chain_plan(const sprs_csr *A, int64_t *n_tiles, int64_t *n_segments, int64_t *n_chains, int64_t *n_other_blocks);

/* MatVecMul::mul_vec / mul_vec_dot (mat.rs:49-64): host slices, checked — returns
 * SPRS_DIM_MISMATCH where the reference panics.  y = A x ; *dot_out = conj(x) . y
 * Thread-safety (bicg_stab.rs:17-18 `T: Send + Sync`, `A: &M` shared): the handle is immutable after creation and
 * every entry point that uses per-context or per-handle scratch — the host-slice mul_vec / mul_vec_dot (staging
 * buffers), everything that returns a scalar, handle creation, all solves — takes the context's mutex, so concurrent
 * `&self` calls on one handle from several host threads are safe: they queue (one context = one stream; use one
 * context per thread for concurrency).  tests/test_gpu_threads.py.  The asynchronous device-pointer entry points
 * (sprs_mul_vec_dev_*, element-wise vecalg) touch no shared scratch; sprs_last_error is last-writer-wins. */
int sprs_mul_vec_d(const sprs_csr *A, const double *x_host, size_t x_len, double *y_host, size_t y_len);
int sprs_mul_vec_z(const sprs_csr *A, const sprs_c64 *x_host, size_t x_len, sprs_c64 *y_host, size_t y_len);
int sprs_mul_vec_dot_d(const sprs_csr *A, const double *x_host, size_t x_len, double *y_host, size_t y_len, double *dot_out);
int sprs_mul_vec_dot_z(const sprs_csr *A, const sprs_c64 *x_host, size_t x_len, sprs_c64 *y_host, size_t y_len, sprs_c64 *dot_out);
/* MatVecMul::mul_vec_unchecked / mul_vec_dot_unchecked (mat.rs:68-152) on device vectors:
 * no dimension check, no PCIe traffic.  x_dev has ncols elements, y_dev nrows.  sprs_mul_vec_dev_* and the
 * element-wise vecalg entry points below are ASYNCHRONOUS on the context's stream (sprs_ctx_sync to wait);
 * entry points that return a scalar, and all solves, block. */
int sprs_mul_vec_dev_d(const sprs_csr *A, const double *x_dev, double *y_dev);
int sprs_mul_vec_dev_z(const sprs_csr *A, const sprs_c64 *x_dev, sprs_c64 *y_dev);
int sprs_mul_vec_dot_dev_d(const sprs_csr *A, const double *x_dev, double *y_dev, double *dot_out);
int sprs_mul_vec_dot_dev_z(const sprs_csr *A, const sprs_c64 *x_dev, sprs_c64 *y_dev, sprs_c64 *dot_out);
/* Launch `reps` back-to-back SpMVs bracketed by HIP events on the context's stream and
 * return the mean device time of one launch in milliseconds (roofline measurement). */
int sprs_mul_vec_dev_timed_d(const sprs_csr *A, const double *x_dev, double *y_dev, int reps, double *ms_per_launch);
int sprs_mul_vec_dev_timed_z(const sprs_csr *A, const sprs_c64 *x_dev, sprs_c64 *y_dev, int reps, double *ms_per_launch);

/* ---------------------------------------------------------------- vecalg (src/vecalg.rs) on device vectors */
int sprs_dot_d(sprs_ctx *ctx, size_t n, const double *x, const double *y, double *out);            /* vecalg.rs:24,556  sum x*y (no conj) */
int sprs_dot_z(sprs_ctx *ctx, size_t n, const sprs_c64 *x, const sprs_c64 *y, sprs_c64 *out);
int sprs_conj_dot_d(sprs_ctx *ctx, size_t n, const double *x, const double *y, double *out);       /* vecalg.rs:51,563  sum conj(x)*y */
int sprs_conj_dot_z(sprs_ctx *ctx, size_t n, const sprs_c64 *x, const sprs_c64 *y, sprs_c64 *out);
int sprs_norm2_d(sprs_ctx *ctx, size_t n, const double *x, double *out);                           /* vecalg.rs:63,601  sqrt(sum |x|^2), unscaled */
int sprs_norm2_z(sprs_ctx *ctx, size_t n, const sprs_c64 *x, double *out);
int sprs_scale_d(sprs_ctx *ctx, size_t n, double a, double *x);                                    /* vecalg.rs:74,592  x *= a */
int sprs_scale_z(sprs_ctx *ctx, size_t n, sprs_c64 a, sprs_c64 *x);
int sprs_rscale_d(sprs_ctx *ctx, size_t n, double a, double *x);                                   /* vecalg.rs:86,596  x = x.mul_real(a) */
int sprs_rscale_z(sprs_ctx *ctx, size_t n, double a, sprs_c64 *x);
int sprs_conj_d(sprs_ctx *ctx, size_t n, const double *in, double *out);                           /* vecalg.rs:96,577  out = conj(in) */
int sprs_conj_z(sprs_ctx *ctx, size_t n, const sprs_c64 *in, sprs_c64 *out);
int sprs_axpy_d(sprs_ctx *ctx, size_t n, double a, const double *x, double *y);                    /* vecalg.rs:109,570 y += x*a */
int sprs_axpy_z(sprs_ctx *ctx, size_t n, sprs_c64 a, const sprs_c64 *x, sprs_c64 *y);
int sprs_axpy_zd(sprs_ctx *ctx, size_t n, double a, const sprs_c64 *x, sprs_c64 *y);               /* S = f64, T = Complex<f64> (vecalg.rs:746-757) */
int sprs_axpby_d(sprs_ctx *ctx, size_t n, double a, const double *x, double b, double *y);         /* vecalg.rs:135,585 y = x*a + y*b */
int sprs_axpby_z(sprs_ctx *ctx, size_t n, sprs_c64 a, const sprs_c64 *x, sprs_c64 b, sprs_c64 *y);

/* ---------------------------------------------------------------- Jacobi preconditioner (src/precond.rs) */
/* DiagPrecond::new(diag): stores 1/diag (precond.rs:20-29; no zero check, as in the reference). */
int sprs_diag_precond_create_d(sprs_ctx *ctx, size_t n, const double *diag_host, sprs_diag **out);    /* DiagPrecond<f64,f64> */
int sprs_diag_precond_create_zd(sprs_ctx *ctx, size_t n, const double *diag_host, sprs_diag **out);   /* DiagPrecond<Complex64,f64>       (tests/test_complex_solve.rs:44) */
int sprs_diag_precond_create_z(sprs_ctx *ctx, size_t n, const sprs_c64 *diag_host, sprs_diag **out);  /* DiagPrecond<Complex64,Complex64> (tests/test_complex_solve2.rs:10) */
int sprs_diag_precond_destroy(sprs_diag *P);
/* MatVecMul::mul_vec for DiagPrecond (precond.rs:37-52): host slices, checked */
int sprs_diag_mul_vec_d(const sprs_diag *P, const double *in_host, size_t in_len, double *out_host, size_t out_len);
int sprs_diag_mul_vec_z(const sprs_diag *P, const sprs_c64 *in_host, size_t in_len, sprs_c64 *out_host, size_t out_len);
/* mul_vec_unchecked on device vectors */
int sprs_diag_mul_vec_dev_d(const sprs_diag *P, const double *in_dev, double *out_dev);
int sprs_diag_mul_vec_dev_z(const sprs_diag *P, const sprs_c64 *in_dev, sprs_c64 *out_dev);

/* ---------------------------------------------------------------- solvers */
/* Common contract (bicg_stab.rs:35-41):  rhs read-only, x in/out (initial guess -> solution),
 * returns a status; on SPRS_OK (*its_out, *res_out) is the reference's Ok((iters, rel_residual)).
 * `*_solve_*`      : rhs/x are host slices (one H2D + one D2H per solve, all iteration state in HBM).
 * `*_solve_dev_*`  : rhs/x are device vectors (nothing crosses PCIe).
 * The solver borrows A (and the preconditioner) — they must outlive it (bicg_stab.rs:18) — and
 * owns its 7n/8n-element workspace, reused across solves (bicg_stab.rs:28).  One in-flight solve
 * per solver handle (`&mut self`). */
int sprs_bicgstab_create_d(const sprs_csr *A, size_t size, sprs_bicgstab **out);   /* BiCGStab::new  bicg_stab.rs:25 */
int sprs_bicgstab_create_z(const sprs_csr *A, size_t size, sprs_bicgstab **out);
int sprs_bicgstab_destroy(sprs_bicgstab *S);
int sprs_bicgstab_solve_d(sprs_bicgstab *S, const double *rhs, size_t rhs_len, double *x, size_t x_len,
                          size_t max_iter, double tol, size_t *its_out, double *res_out);            /* bicg_stab.rs:35-200 */
int sprs_bicgstab_solve_z(sprs_bicgstab *S, const sprs_c64 *rhs, size_t rhs_len, sprs_c64 *x, size_t x_len,
                          size_t max_iter, double tol, size_t *its_out, double *res_out);
int sprs_bicgstab_precond_solve_d(sprs_bicgstab *S, const sprs_diag *P, const double *rhs, size_t rhs_len, double *x,
                                  size_t x_len, size_t max_iter, double tol, size_t *its_out, double *res_out); /* bicg_stab.rs:204-366 */
int sprs_bicgstab_precond_solve_z(sprs_bicgstab *S, const sprs_diag *P, const sprs_c64 *rhs, size_t rhs_len, sprs_c64 *x,
                                  size_t x_len, size_t max_iter, double tol, size_t *its_out, double *res_out);
int sprs_bicgstab_solve_dev_d(sprs_bicgstab *S, const sprs_diag *P_or_null, const double *rhs_dev, size_t rhs_len,
                              double *x_dev, size_t x_len, size_t max_iter, double tol, size_t *its_out, double *res_out);
int sprs_bicgstab_solve_dev_z(sprs_bicgstab *S, const sprs_diag *P_or_null, const sprs_c64 *rhs_dev, size_t rhs_len,
                              sprs_c64 *x_dev, size_t x_len, size_t max_iter, double tol, size_t *its_out, double *res_out);

int sprs_minres_create_d(const sprs_csr *A, size_t size, sprs_minres **out);       /* MinRes::new  minres.rs:21 */
int sprs_minres_create_z(const sprs_csr *A, size_t size, sprs_minres **out);
int sprs_minres_destroy(sprs_minres *S);
int sprs_minres_solve_d(sprs_minres *S, const double *rhs, size_t rhs_len, double *x, size_t x_len,
                        size_t max_iter, double tol, size_t *its_out, double *res_out);              /* minres.rs:31-172 */
int sprs_minres_solve_z(sprs_minres *S, const sprs_c64 *rhs, size_t rhs_len, sprs_c64 *x, size_t x_len,
                        size_t max_iter, double tol, size_t *its_out, double *res_out);
int sprs_minres_precond_solve_d(sprs_minres *S, const sprs_diag *P, const double *rhs, size_t rhs_len, double *x,
                                size_t x_len, size_t max_iter, double tol, size_t *its_out, double *res_out); /* minres.rs:178-341 */
int sprs_minres_precond_solve_z(sprs_minres *S, const sprs_diag *P, const sprs_c64 *rhs, size_t rhs_len, sprs_c64 *x,
                                size_t x_len, size_t max_iter, double tol, size_t *its_out, double *res_out);
int sprs_minres_solve_dev_d(sprs_minres *S, const sprs_diag *P_or_null, const double *rhs_dev, size_t rhs_len,
                            double *x_dev, size_t x_len, size_t max_iter, double tol, size_t *its_out, double *res_out);
int sprs_minres_solve_dev_z(sprs_minres *S, const sprs_diag *P_or_null, const sprs_c64 *rhs_dev, size_t rhs_len,
                            sprs_c64 *x_dev, size_t x_len, size_t max_iter, double tol, size_t *its_out, double *res_out);

int sprs_csminres_create_z(const sprs_csr *A, size_t size, sprs_csminres **out);   /* CSMinRes::new  cs_minres.rs:19 */
int sprs_csminres_create_d(const sprs_csr *A, size_t size, sprs_csminres **out);   /* generic over T: real T degenerates to MINRES arithmetic */
int sprs_csminres_destroy(sprs_csminres *S);
int sprs_csminres_solve_z(sprs_csminres *S, const sprs_c64 *rhs, size_t rhs_len, sprs_c64 *x, size_t x_len,
                          size_t max_iter, double tol, size_t *its_out, double *res_out);            /* cs_minres.rs:29-158 */
int sprs_csminres_solve_d(sprs_csminres *S, const double *rhs, size_t rhs_len, double *x, size_t x_len,
                          size_t max_iter, double tol, size_t *its_out, double *res_out);
int sprs_csminres_solve_dev_z(sprs_csminres *S, const sprs_c64 *rhs_dev, size_t rhs_len, sprs_c64 *x_dev, size_t x_len,
                              size_t max_iter, double tol, size_t *its_out, double *res_out);
int sprs_csminres_solve_dev_d(sprs_csminres *S, const double *rhs_dev, size_t rhs_len, double *x_dev, size_t x_len,
                              size_t max_iter, double tol, size_t *its_out, double *res_out);

/* ---------------------------------------------------------------- Gauss-Seidel (src/gauss_seidel.rs)
 * `GaussSeidel::new(A.view())` / `::solve(rhs, x, max_iter, eps)`; real scalars only (the reference bounds
 * T: PartialOrd).  The serial sweep is made data-parallel by dependency levels (rows of one level per launch)
 * without changing its arithmetic: x after k sweeps is bit-identical to the reference's.  On SPRS_OK
 * (*its_out, *res_out) = Ok((iters, ABSOLUTE residual norm)) exactly as gauss_seidel.rs:107,136 return them.
 * create: SPRS_NOT_SQUARE / SPRS_NOT_CSR (the handle was built from CSC arrays) as the reference's `new`. */
int sprs_gauss_seidel_create(const sprs_csr *A, sprs_gauss_seidel **out);
int sprs_gauss_seidel_destroy(sprs_gauss_seidel *G);
int64_t sprs_gauss_seidel_levels(const sprs_gauss_seidel *G);   /* number of dependency levels (launches per sweep) */
int sprs_gauss_seidel_solve_d(sprs_gauss_seidel *G, const double *rhs, size_t rhs_len, double *x, size_t x_len, size_t max_iter, double eps, size_t *its_out, double *res_out);
int sprs_gauss_seidel_solve_s(sprs_gauss_seidel *G, const float *rhs, size_t rhs_len, float *x, size_t x_len, size_t max_iter, float eps, size_t *its_out, float *res_out);
int sprs_gauss_seidel_solve_dev_d(sprs_gauss_seidel *G, const double *rhs_dev, size_t rhs_len, double *x_dev, size_t x_len, size_t max_iter, double eps, size_t *its_out, double *res_out);
int sprs_gauss_seidel_solve_dev_s(sprs_gauss_seidel *G, const float *rhs_dev, size_t rhs_len, float *x_dev, size_t x_len, size_t max_iter, float eps, size_t *its_out, float *res_out);

/* ---------------------------------------------------------------- f32 / Complex<f32>
 * The reference is generic over cauchy::Scalar = {f32, f64, c32, c64} and its unit tests exercise f32 / c32
 * BLAS-1 (src/vecalg.rs:647-658,669-677,771-798,816-830).  Every typed entry point above exists again with
 * suffix _s (f32), _c (Complex<f32> = sprs_c32) and _cs (complex vector, real scalar/diagonal); T::Real
 * quantities (tol, residual, norm2, rscale factor) are float. */
int sprs_csr_create_s(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t *row_ptr, const int32_t *col_idx, const float *val, int storage_csc, sprs_csr **out);
int sprs_csr_create_c(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t *row_ptr, const int32_t *col_idx, const sprs_c32 *val, int storage_csc, sprs_csr **out);
int sprs_csr_create_i64_s(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t *row_ptr, const int64_t *col_idx, const float *val, int storage_csc, sprs_csr **out);
int sprs_csr_create_i64_c(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t *row_ptr, const int64_t *col_idx, const sprs_c32 *val, int storage_csc, sprs_csr **out);
int sprs_csr_create_dev_s(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t *dev_row_ptr, const int32_t *dev_col_idx, const float *dev_val, int adopt, sprs_csr **out);
int sprs_csr_create_dev_c(sprs_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int32_t *dev_row_ptr, const int32_t *dev_col_idx, const sprs_c32 *dev_val, int adopt, sprs_csr **out);
int sprs_mul_vec_s(const sprs_csr *A, const float *x_host, size_t x_len, float *y_host, size_t y_len);
int sprs_mul_vec_c(const sprs_csr *A, const sprs_c32 *x_host, size_t x_len, sprs_c32 *y_host, size_t y_len);
int sprs_mul_vec_dot_s(const sprs_csr *A, const float *x_host, size_t x_len, float *y_host, size_t y_len, float *dot_out);
int sprs_mul_vec_dot_c(const sprs_csr *A, const sprs_c32 *x_host, size_t x_len, sprs_c32 *y_host, size_t y_len, sprs_c32 *dot_out);
int sprs_mul_vec_dev_s(const sprs_csr *A, const float *x_dev, float *y_dev);
int sprs_mul_vec_dev_c(const sprs_csr *A, const sprs_c32 *x_dev, sprs_c32 *y_dev);
int sprs_mul_vec_dot_dev_s(const sprs_csr *A, const float *x_dev, float *y_dev, float *dot_out);
int sprs_mul_vec_dot_dev_c(const sprs_csr *A, const sprs_c32 *x_dev, sprs_c32 *y_dev, sprs_c32 *dot_out);
int sprs_mul_vec_dev_timed_s(const sprs_csr *A, const float *x_dev, float *y_dev, int reps, double *ms_per_launch);
int sprs_mul_vec_dev_timed_c(const sprs_csr *A, const sprs_c32 *x_dev, sprs_c32 *y_dev, int reps, double *ms_per_launch);
int sprs_dot_s(sprs_ctx *ctx, size_t n, const float *x, const float *y, float *out);
int sprs_dot_c(sprs_ctx *ctx, size_t n, const sprs_c32 *x, const sprs_c32 *y, sprs_c32 *out);
int sprs_conj_dot_s(sprs_ctx *ctx, size_t n, const float *x, const float *y, float *out);
int sprs_conj_dot_c(sprs_ctx *ctx, size_t n, const sprs_c32 *x, const sprs_c32 *y, sprs_c32 *out);
int sprs_norm2_s(sprs_ctx *ctx, size_t n, const float *x, float *out);
int sprs_norm2_c(sprs_ctx *ctx, size_t n, const sprs_c32 *x, float *out);
int sprs_scale_s(sprs_ctx *ctx, size_t n, float a, float *x);
int sprs_scale_c(sprs_ctx *ctx, size_t n, sprs_c32 a, sprs_c32 *x);
int sprs_rscale_s(sprs_ctx *ctx, size_t n, float a, float *x);
int sprs_rscale_c(sprs_ctx *ctx, size_t n, float a, sprs_c32 *x);
int sprs_conj_s(sprs_ctx *ctx, size_t n, const float *in, float *out);
int sprs_conj_c(sprs_ctx *ctx, size_t n, const sprs_c32 *in, sprs_c32 *out);
int sprs_axpy_s(sprs_ctx *ctx, size_t n, float a, const float *x, float *y);
int sprs_axpy_c(sprs_ctx *ctx, size_t n, sprs_c32 a, const sprs_c32 *x, sprs_c32 *y);
int sprs_axpy_cs(sprs_ctx *ctx, size_t n, float a, const sprs_c32 *x, sprs_c32 *y);
int sprs_axpby_s(sprs_ctx *ctx, size_t n, float a, const float *x, float b, float *y);
int sprs_axpby_c(sprs_ctx *ctx, size_t n, sprs_c32 a, const sprs_c32 *x, sprs_c32 b, sprs_c32 *y);
int sprs_diag_precond_create_s(sprs_ctx *ctx, size_t n, const float *diag_host, sprs_diag **out);
int sprs_diag_precond_create_cs(sprs_ctx *ctx, size_t n, const float *diag_host, sprs_diag **out);
int sprs_diag_precond_create_c(sprs_ctx *ctx, size_t n, const sprs_c32 *diag_host, sprs_diag **out);
int sprs_diag_mul_vec_s(const sprs_diag *P, const float *in_host, size_t in_len, float *out_host, size_t out_len);
int sprs_diag_mul_vec_c(const sprs_diag *P, const sprs_c32 *in_host, size_t in_len, sprs_c32 *out_host, size_t out_len);
int sprs_diag_mul_vec_dev_s(const sprs_diag *P, const float *in_dev, float *out_dev);
int sprs_diag_mul_vec_dev_c(const sprs_diag *P, const sprs_c32 *in_dev, sprs_c32 *out_dev);
int sprs_bicgstab_create_s(const sprs_csr *A, size_t size, sprs_bicgstab **out);
int sprs_bicgstab_create_c(const sprs_csr *A, size_t size, sprs_bicgstab **out);
int sprs_bicgstab_solve_s(sprs_bicgstab *S, const float *rhs, size_t rhs_len, float *x, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_bicgstab_solve_c(sprs_bicgstab *S, const sprs_c32 *rhs, size_t rhs_len, sprs_c32 *x, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_bicgstab_precond_solve_s(sprs_bicgstab *S, const sprs_diag *P, const float *rhs, size_t rhs_len, float *x, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_bicgstab_precond_solve_c(sprs_bicgstab *S, const sprs_diag *P, const sprs_c32 *rhs, size_t rhs_len, sprs_c32 *x, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_bicgstab_solve_dev_s(sprs_bicgstab *S, const sprs_diag *P_or_null, const float *rhs_dev, size_t rhs_len, float *x_dev, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_bicgstab_solve_dev_c(sprs_bicgstab *S, const sprs_diag *P_or_null, const sprs_c32 *rhs_dev, size_t rhs_len, sprs_c32 *x_dev, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_minres_create_s(const sprs_csr *A, size_t size, sprs_minres **out);
int sprs_minres_create_c(const sprs_csr *A, size_t size, sprs_minres **out);
int sprs_minres_solve_s(sprs_minres *S, const float *rhs, size_t rhs_len, float *x, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_minres_solve_c(sprs_minres *S, const sprs_c32 *rhs, size_t rhs_len, sprs_c32 *x, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_minres_precond_solve_s(sprs_minres *S, const sprs_diag *P, const float *rhs, size_t rhs_len, float *x, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_minres_precond_solve_c(sprs_minres *S, const sprs_diag *P, const sprs_c32 *rhs, size_t rhs_len, sprs_c32 *x, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_minres_solve_dev_s(sprs_minres *S, const sprs_diag *P_or_null, const float *rhs_dev, size_t rhs_len, float *x_dev, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_minres_solve_dev_c(sprs_minres *S, const sprs_diag *P_or_null, const sprs_c32 *rhs_dev, size_t rhs_len, sprs_c32 *x_dev, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_csminres_create_c(const sprs_csr *A, size_t size, sprs_csminres **out);
int sprs_csminres_create_s(const sprs_csr *A, size_t size, sprs_csminres **out);
int sprs_csminres_solve_c(sprs_csminres *S, const sprs_c32 *rhs, size_t rhs_len, sprs_c32 *x, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_csminres_solve_s(sprs_csminres *S, const float *rhs, size_t rhs_len, float *x, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_csminres_solve_dev_c(sprs_csminres *S, const sprs_c32 *rhs_dev, size_t rhs_len, sprs_c32 *x_dev, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_csminres_solve_dev_s(sprs_csminres *S, const float *rhs_dev, size_t rhs_len, float *x_dev, size_t x_len, size_t max_iter, float tol, size_t *its_out, float *res_out);
int sprs_dist_csr_create_dev_s(sprs_comm *comm, int64_t n_local, int64_t n_ext, int64_t nnz, const int32_t *dev_row_ptr, const int32_t *dev_col_idx_ext, const float *dev_val, int adopt, int n_peers, const int32_t *peer_rank, const int64_t *send_off, const int32_t *send_idx_dev, const int64_t *recv_off, sprs_csr **out);
int sprs_dist_csr_create_dev_c(sprs_comm *comm, int64_t n_local, int64_t n_ext, int64_t nnz, const int32_t *dev_row_ptr, const int32_t *dev_col_idx_ext, const sprs_c32 *dev_val, int adopt, int n_peers, const int32_t *peer_rank, const int64_t *send_off, const int32_t *send_idx_dev, const int64_t *recv_off, sprs_csr **out);
int sprs_dist_mul_vec_dev_s(const sprs_csr *A, float *x_ext_dev, float *y_local_dev);
int sprs_dist_mul_vec_dev_c(const sprs_csr *A, sprs_c32 *x_ext_dev, sprs_c32 *y_local_dev);

/* ---------------------------------------------------------------- multi-GPU (one process per GPU)
 * No reference analogue: the reference is single-process (rayon).  The matrix is row-partitioned;
 * rank r owns rows [r0, r1) and the matching slices of every vector.  A distributed operator is the
 * local row block with column indices renumbered into the rank's EXTENDED x vector
 * [ owned entries (n_local) | entries received from peer 0 | peer 1 | ... ]  (n_ext elements).
 * Every solver above accepts such an operator (create it with size = n_local): SpMV is preceded by
 * the halo exchange (RCCL send/recv with the owning peers over xGMI), every dot product / norm is
 * followed by an RCCL all-reduce, and rhs / x are the rank's slices.  sprsolve_amd/partition.py
 * derives the exchange plan from global column indices. */
int sprs_comm_unique_id(void *id128_out);  /* rank 0: 128-byte RCCL id to broadcast to the other ranks */
int sprs_comm_create(sprs_ctx *ctx, int world, int rank, const void *id128, sprs_comm **out); /* collective */
int sprs_comm_destroy(sprs_comm *comm);
/* 1 when the ranks of this communicator mapped each other's mailboxes at creation (same node, hipIpc): the distributed solvers'
 * scalar hand-offs then need no stream operation — the producing kernel's last workgroup posts its reduced values into every
 * rank's mailbox, the consumer kernels sum the entries in rank order (ctx knobs "p2p_allreduce", "p2p_timeout_ms"); 0: ncclAllReduce. */
int sprs_comm_p2p(const sprs_comm *comm, int *enabled_out);
int sprs_comm_count(const sprs_comm *comm, int *count_out); /* ncclCommCount: the number of ranks RCCL itself reports */
int sprs_comm_allreduce_sum_f64(sprs_comm *comm, double *dev, size_t count); /* in place; blocking */
/* mean time of `reps` back-to-back in-place all-reduces of `count` doubles on the context's stream (HIP events): the
 * price of one dot-product hand-off of a distributed solve on this communicator; collective */
int sprs_comm_allreduce_timed_f64(sprs_comm *comm, double *dev, size_t count, int reps, double *us_out);
/* peer_rank[n_peers]; send_off/recv_off[n_peers+1] are element offsets; send_idx_dev[send_off[n_peers]]
 * (device, i32) lists the local entries to pack for each peer; entries from peer p land at
 * x_ext[n_local + recv_off[p] ...].  recv_off[n_peers] == n_ext - n_local. */
int sprs_dist_csr_create_dev_d(sprs_comm *comm, int64_t n_local, int64_t n_ext, int64_t nnz,
                               const int32_t *dev_row_ptr, const int32_t *dev_col_idx_ext, const double *dev_val, int adopt,
                               int n_peers, const int32_t *peer_rank, const int64_t *send_off,
                               const int32_t *send_idx_dev, const int64_t *recv_off, sprs_csr **out);
int sprs_dist_csr_create_dev_z(sprs_comm *comm, int64_t n_local, int64_t n_ext, int64_t nnz,
                               const int32_t *dev_row_ptr, const int32_t *dev_col_idx_ext, const sprs_c64 *dev_val, int adopt,
                               int n_peers, const int32_t *peer_rank, const int64_t *send_off,
                               const int32_t *send_idx_dev, const int64_t *recv_off, sprs_csr **out);
/* north_star's literal exchange instead of the sparse halo: before every SpMV an ncclAllGather of all ranks'
 * x slices (each padded to `slice` >= max n_local elements, the same value on every rank).  Column indices
 * address the gathered vector: col = owner_rank * slice + (global_col - first_row_of_owner).  Moves
 * world*slice elements per SpMV (400 MB for cfg 5) — kept for comparison; the sparse halo is the default. */
/* The same operator from GLOBAL column indices — the exchange plan is derived inside the library (device passes +
 * one ncclAllGather of the per-peer counts + one ncclSend/ncclRecv group of the index lists), which is what a host
 * without numpy (the Rust binding) calls.  Collective over comm.  row_starts[world + 1] (host, identical on all ranks):
 * rank r owns rows / x entries [row_starts[r], row_starts[r+1]); dev_col_idx_global (device, i32[nnz]) holds global
 * column numbers and is renumbered IN PLACE into the rank's extended numbering when adopt != 0 (into a private copy
 * otherwise).  exchange: 0 = sparse halo (only the referenced remote entries travel, from the ranks that own them),
 * 1 = north_star's literal ncclAllGather of every rank's x slice.  Temporary device memory: 5 bytes per global column.
 * SPRS_INVALID_ARGUMENT when a column index is outside [0, row_starts[world]). */
int sprs_dist_csr_create_global_dev_d(sprs_comm *comm, const int64_t *row_starts, int64_t nnz, const int32_t *dev_row_ptr, int32_t *dev_col_idx_global, const double *dev_val, int adopt, int exchange, sprs_csr **out);
int sprs_dist_csr_create_global_dev_z(sprs_comm *comm, const int64_t *row_starts, int64_t nnz, const int32_t *dev_row_ptr, int32_t *dev_col_idx_global, const sprs_c64 *dev_val, int adopt, int exchange, sprs_csr **out);
int sprs_dist_csr_create_global_dev_s(sprs_comm *comm, const int64_t *row_starts, int64_t nnz, const int32_t *dev_row_ptr, int32_t *dev_col_idx_global, const float *dev_val, int adopt, int exchange, sprs_csr **out);
int sprs_dist_csr_create_global_dev_c(sprs_comm *comm, const int64_t *row_starts, int64_t nnz, const int32_t *dev_row_ptr, int32_t *dev_col_idx_global, const sprs_c32 *dev_val, int adopt, int exchange, sprs_csr **out);
/* The plan of a distributed operator, read back (tests, bench evidence): sizes; then peer ranks and element offsets
 * (arrays of cap >= n_peers and n_peers + 1 entries); then the local indices this rank packs, grouped by peer. */
int sprs_dist_csr_info(const sprs_csr *A, int64_t *n_local, int64_t *n_ext, int *n_peers, int64_t *send_entries, int64_t *recv_entries);
int sprs_dist_csr_peers(const sprs_csr *A, int cap, int32_t *peer_rank, int64_t *send_off, int64_t *recv_off);
int sprs_dist_csr_send_idx(const sprs_csr *A, int64_t cap, int32_t *send_idx_host);
int sprs_dist_csr_create_allgather_dev_d(sprs_comm *comm, int64_t n_local, int64_t slice, int64_t nnz, const int32_t *dev_row_ptr, const int32_t *dev_col_idx_gathered, const double *dev_val, int adopt, sprs_csr **out);
int sprs_dist_csr_create_allgather_dev_z(sprs_comm *comm, int64_t n_local, int64_t slice, int64_t nnz, const int32_t *dev_row_ptr, const int32_t *dev_col_idx_gathered, const sprs_c64 *dev_val, int adopt, sprs_csr **out);
int sprs_dist_csr_create_allgather_dev_s(sprs_comm *comm, int64_t n_local, int64_t slice, int64_t nnz, const int32_t *dev_row_ptr, const int32_t *dev_col_idx_gathered, const float *dev_val, int adopt, sprs_csr **out);
int sprs_dist_csr_create_allgather_dev_c(sprs_comm *comm, int64_t n_local, int64_t slice, int64_t nnz, const int32_t *dev_row_ptr, const int32_t *dev_col_idx_gathered, const sprs_c32 *dev_val, int adopt, sprs_csr **out);
/* y_local = A_local * x after exchanging the halo tail of x_ext (n_ext elements, owned slice first) */
int sprs_dist_mul_vec_dev_d(const sprs_csr *A, double *x_ext_dev, double *y_local_dev);
int sprs_dist_mul_vec_dev_z(const sprs_csr *A, sprs_c64 *x_ext_dev, sprs_c64 *y_local_dev);

/* ---------------------------------------------------------------- solver options / instrumentation
 * `solver` is any of the three solver handle types. */
enum { SPRS_SOLVER_BICGSTAB = 1, SPRS_SOLVER_MINRES = 2, SPRS_SOLVER_CSMINRES = 3 };
/* mode 0 (default): fused kernels, device-resident scalars, lazy host polling.
 * mode 1: "literal" — the reference's op list one kernel per op, every scalar consumed on the
 *         host exactly where the reference consumes it (bicg_stab.rs:122-197). */
int sprs_solver_set_mode(void *solver, int kind, int mode);
/* Per-iteration scalar trace (8 doubles per row: BiCGStab [its, r_norm, rho, alpha, w]; MINRES [its, beta, alpha, c, s, res_norm]): the solver
 * synchronises every iteration while a trace buffer is set.  rows_out: rows written by the last solve. */
int sprs_solver_set_trace(void *solver, int kind, double *trace_host, size_t capacity_rows);
int sprs_solver_trace_rows(const void *solver, int kind, size_t *rows_out);
/* Device-time profile of the last solve: total milliseconds and launch count of the SpMV
 * kernel measured with HIP events on the solver's stream.  enable: 0 off; 1 every SpMV launch; k >= 2 a SAMPLE — one pair
 * of consecutive SpMV launches in k (a launch that carries events costs ~6 us more: a fifth of a 30 us iteration). */
int sprs_solver_set_profile(void *solver, int kind, int enable);
/* ... and what the timed launches were: `steps` = SpMV launches of the solve (timed or not); among the TIMED ones (the
 * `launches` of sprs_solver_get_profile) how many read a dot operand other than their input vector, how many were K2 / K4
 * launches that formed their input on the fly ("spmv_fuse").  Any pointer may be NULL. */
int sprs_solver_get_profile_counts(const void *solver, int kind, int64_t *steps, int64_t *timed_dot_other, int64_t *timed_k2_fused,
                                   int64_t *timed_k4_fused);
int sprs_solver_get_profile(const void *solver, int kind, double *spmv_ms_total, int64_t *spmv_launches,
                            double *solve_ms_total);
/* How many SpMV launches of the last solve formed their input vector on the fly (ctx knob "spmv_fuse"; MINRES: its SpMV launches that
 * multiplied by the un-normalised vector are reported in k2_fused): BiCGStab's K2 with K1's
 * update p = (v (-beta w) + p beta) + r inside, K4 with K3's r -= alpha v inside (bicg_stab.rs:155-156,172).  Zero for the other
 * solvers and wherever the five-launch iteration ran. */
int sprs_solver_get_fused_launches(const void *solver, int kind, int64_t *k2_fused, int64_t *k4_fused);

#ifdef __cplusplus
}
#endif
#endif /* SPRSOLVE_HIP_H */
