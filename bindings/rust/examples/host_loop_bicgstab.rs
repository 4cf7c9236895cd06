//! north_star's literal structure in Rust: the HOST owns the Krylov recurrence (src/bicg_stab.rs:35-200,
//! statement for statement) and calls one HIP kernel per reference op — `HipCsr::mul_vec_dev` for
//! `A.mul_vec_unchecked`, `sprsolve_hip::vecalg::*` for `sprsolve::vecalg::*`, `DevVec::copy_from` for
//! `ptr::copy_nonoverlapping`.  Vectors stay in HBM; five scalars per iteration cross PCIe.
//! The Rust twin of `examples/host_loop_bicgstab.c` (which the GPU tests compile and run; this file is
//! UNVERIFIED like the rest of the crate — no Rust toolchain in the build environment).
//!
//!     SPRSOLVE_HIP_LIB_DIR=../../sprsolve_amd cargo run --example host_loop_bicgstab -- 96
use sprs::{CsMat, TriMat};
use sprsolve::error::{SolveResult, SolverError};
use sprsolve_hip::{vecalg, DevVec, HipBiCGStab, HipCsr};

/// src/bicg_stab.rs:35-200 over device vectors: r = A x - b (negative residual), x -= ...
fn host_bicgstab(a: &HipCsr<f64>, rhs: &DevVec<f64>, x: &mut DevVec<f64>, max_iter: usize, tol: f64) -> SolveResult<(usize, f64)> {
    let n = rhs.len();
    let eps = f64::EPSILON;
    let rhs_norm = vecalg::norm2(rhs);                              // :55
    if rhs_norm <= eps { x.fill_zero(); return Ok((0, rhs_norm)); } // :56-60
    let tol2 = tol * rhs_norm;
    let (mut r, mut r0, mut y, mut v, mut t) =
        (DevVec::<f64>::zeros(n), DevVec::<f64>::zeros(n), DevVec::<f64>::zeros(n), DevVec::<f64>::zeros(n), DevVec::<f64>::zeros(n));
    a.mul_vec_dev(x, &mut r);                                       // :73
    vecalg::axpy(-1.0, rhs, &mut r);                                // :75
    r0.copy_from(&r);                                               // :78
    let r0_norm = vecalg::norm2(&r0);                               // :80
    if r0_norm <= tol2 { return Ok((0, r0_norm / rhs_norm)); }
    let mut r0_norm_tol = r0_norm * eps; r0_norm_tol *= r0_norm_tol; // :84-85
    let mut rho = r0_norm * r0_norm;                                // :88
    y.copy_from(&r);                                                // :91
    a.mul_vec_dev(&y, &mut v);                                      // :93
    let mut alpha = rho / vecalg::conj_dot(&r0, &v);                // :96 (no breakdown test in the unrolled iteration)
    vecalg::axpy(-alpha, &v, &mut r);                               // :101
    a.mul_vec_dev(&r, &mut t);                                      // :104
    let tt = vecalg::conj_dot(&t, &t);                              // :107
    let mut w = if tt > 0.0 { vecalg::conj_dot(&t, &r) / tt } else { 0.0 };
    vecalg::axpy(-alpha, &y, x);                                    // :115
    vecalg::axpy(-w, &r, x);                                        // :117
    vecalg::axpy(-w, &t, &mut r);                                   // :120
    for its in 1..max_iter {                                        // :122
        let r_norm = vecalg::norm2(&r);                             // :123
        if r_norm <= tol2 { return Ok((its, r_norm / rhs_norm)); }
        let rho_old = rho;
        rho = vecalg::conj_dot(&r0, &r);                            // :128
        if rho.abs() < r0_norm_tol {                                // :131-145 restart
            a.mul_vec_dev(x, &mut r);
            vecalg::axpy(-1.0, rhs, &mut r);
            r0.copy_from(&r);
            let rn = vecalg::norm2(&r);
            rho = rn * rn;
            r0_norm_tol = rho * eps * eps;
        }
        let beta = (rho / rho_old) * (alpha / w);                   // :146
        vecalg::axpby(-beta * w, &v, beta, &mut y);                 // :155
        vecalg::axpy(1.0, &r, &mut y);                              // :156
        a.mul_vec_dev(&y, &mut v);                                  // :160
        let tmp = vecalg::conj_dot(&r0, &v);                        // :163
        if tmp.abs() <= 0.0 { return Err(SolverError::BreakDown(its)); } // :164-167
        alpha = rho / tmp;
        vecalg::axpy(-alpha, &v, &mut r);                           // :172
        a.mul_vec_dev(&r, &mut t);                                  // :175
        let tt = vecalg::conj_dot(&t, &t);                          // :178
        w = if tt > 0.0 { vecalg::conj_dot(&t, &r) / tt } else { 0.0 };
        vecalg::axpy(-alpha, &y, x);                                // :188
        vecalg::axpy(-w, &r, x);                                    // :191
        vecalg::axpy(-w, &t, &mut r);                               // :196
    }
    Err(SolverError::InsufficientIterNum(max_iter))                 // :199
}

/// benches/bicgstab.rs:54-104: Dirichlet rows are identity, interior rows the 5-point stencil; rhs = i + j on the border.
fn grid_laplacian(rows: usize) -> (CsMat<f64>, Vec<f64>) {
    let n = rows * rows;
    let mut tri = TriMat::new((n, n));
    let mut rhs = vec![0f64; n];
    for i in 0..rows {
        for j in 0..rows {
            let r = i * rows + j;
            if i == 0 || j == 0 || i == rows - 1 || j == rows - 1 {
                tri.add_triplet(r, r, 1.0);
                rhs[r] = (i + j) as f64;
            } else {
                tri.add_triplet(r, r - rows, 1.0); tri.add_triplet(r, r - 1, 1.0); tri.add_triplet(r, r, -4.0);
                tri.add_triplet(r, r + 1, 1.0); tri.add_triplet(r, r + rows, 1.0);
            }
        }
    }
    (tri.to_csr(), rhs)
}

fn main() {
    let rows: usize = std::env::args().nth(1).map(|s| s.parse().unwrap()).unwrap_or(64);
    let (lap, rhs) = grid_laplacian(rows);
    let a = HipCsr::new(&lap).expect("handle creation");          // CsMat<f64> = usize indices -> sprs_csr_create_i64_d
    let (d_rhs, mut d_x) = (DevVec::from_slice(&rhs), DevVec::<f64>::zeros(rhs.len()));
    let (its_a, res_a) = host_bicgstab(&a, &d_rhs, &mut d_x, 20000, 1e-10).unwrap();
    let xa = d_x.to_vec();
    // the library's own recurrence on the same system (device-resident scalars)
    let mut xb = vec![0f64; rhs.len()];
    let mut solver = HipBiCGStab::new(&a, a.cols());
    let (its_b, res_b) = solver.solve(rhs.as_slice(), xb.as_mut_slice(), 20000, 1e-10).unwrap();
    let err = xa.iter().enumerate().map(|(r, v)| (v - ((r / rows + r % rows) as f64)).abs()).fold(0f64, f64::max);
    let dlt = xa.iter().zip(xb.iter()).map(|(p, q)| (p - q).abs()).fold(0f64, f64::max);
    println!("n={} host-loop its={} res={:.3e} | library its={} res={:.3e} max|dx|={:.2e} | max_err={:.3e}",
             rhs.len(), its_a, res_a, its_b, res_b, dlt, err);
    assert!(err < 1e-6 && dlt < 1e-6);
}
