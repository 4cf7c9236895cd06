/* The drop-in boundary from plain C: build the reference's bench matrix (benches/bicgstab.rs:54-104 — 5-point
 * Laplacian with identity rows on the border, rhs = i + j on the border), solve it with BiCGStab + Jacobi through
 * include/sprsolve_hip.h exactly as a Rust / C host would (host slices in, host slice out), and check the known
 * solution x[i*R + j] = i + j.
 *   gcc -std=c99 -I include examples/c_abi_demo.c -o c_abi_demo -L sprsolve_amd -l:libsprsolve_hip.so \
 *       -Wl,-rpath,$PWD/sprsolve_amd -Wl,-rpath,/opt/rocm/lib -lm && ./c_abi_demo 96 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "sprsolve_hip.h"

#define CHECK(call) do { int st__ = (call); if (st__ != SPRS_OK) { fprintf(stderr, "%s -> %d (%s) %s\n", #call, st__, \
    sprs_status_str(st__), ctx ? sprs_last_error(ctx) : ""); return 1; } } while (0)

int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 64;
    const int n = R * R;
    int32_t *rp = malloc(sizeof(int32_t) * (size_t)(n + 1)), *ci = malloc(sizeof(int32_t) * (size_t)n * 5);
    double *val = malloc(sizeof(double) * (size_t)n * 5), *rhs = calloc((size_t)n, sizeof(double));
    double *x = calloc((size_t)n, sizeof(double)), *diag = malloc(sizeof(double) * (size_t)n);
    int64_t nnz = 0;
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) {
            const int r = i * R + j;
            rp[r] = (int32_t)nnz;
            if (i == 0 || j == 0 || i == R - 1 || j == R - 1) {          /* Dirichlet row */
                ci[nnz] = r; val[nnz++] = 1.0; diag[r] = 1.0; rhs[r] = (double)(i + j);
            } else {
                const int c[5] = {r - R, r - 1, r, r + 1, r + R};
                const double v[5] = {1.0, 1.0, -4.0, 1.0, 1.0};
                for (int k = 0; k < 5; ++k) { ci[nnz] = c[k]; val[nnz++] = v[k]; }
                diag[r] = -4.0;
            }
        }
    rp[n] = (int32_t)nnz;

    sprs_ctx *ctx = NULL; sprs_csr *A = NULL; sprs_diag *P = NULL; sprs_bicgstab *S = NULL;
    CHECK(sprs_ctx_create(0, NULL, &ctx));
    CHECK(sprs_csr_create_d(ctx, n, n, nnz, rp, ci, val, 0, &A));
    int n_off = 0, n_pair = 0;
    const int stream = sprs_csr_stream_format(A, &n_off, &n_pair);
    CHECK(sprs_diag_precond_create_d(ctx, (size_t)n, diag, &P));
    CHECK(sprs_bicgstab_create_d(A, (size_t)n, &S));
    size_t its = 0; double res = 0.0;
    CHECK(sprs_bicgstab_precond_solve_d(S, P, rhs, (size_t)n, x, (size_t)n, 20000, 1e-10, &its, &res));
    double err = 0.0;
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) err = fmax(err, fabs(x[i * R + j] - (double)(i + j)));
    printf("n=%d nnz=%lld stream=%d (offsets %d, pairs %d) iterations=%zu rel_res=%.3e max_err=%.3e\n", n, (long long)nnz,
           stream, n_off, n_pair, its, res, err);
    sprs_bicgstab_destroy(S); sprs_diag_precond_destroy(P); sprs_csr_destroy(A); sprs_ctx_destroy(ctx);
    free(rp); free(ci); free(val); free(rhs); free(x); free(diag);
    return err < 1e-6 ? 0 : 2;
}
