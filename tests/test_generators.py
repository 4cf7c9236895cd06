"""Generators: the numpy restatements against scipy structure checks, and the torch (device)
twins against the numpy ones (on CPU tensors here)."""
import numpy as np
import scipy.sparse as sp

from sprsolve_amd import gen


def test_bench_grid_matches_reference_counts():
    # benches/bicgstab.rs "Laplacian-100": n = 10 000, nnz = 48 416; mat_vec_mul "-140": 95 776
    ip, ix, d = gen.grid_laplacian_dirichlet(100, 100)
    assert ip[-1] == 48416 and ip.size == 10001
    ip, ix, d = gen.grid_laplacian_dirichlet(140, 140)
    assert ip[-1] == 95776
    # BASELINE cfg 2 / cfg 4 / cfg 3 / cfg 1(ii) nnz counts
    assert gen.grid_laplacian_dirichlet(1000, 1000)[0][-1] == 4984016
    assert gen.random_tridiagonal(10000)[0][-1] == 29998
    assert gen.symmetric_banded(10000)[0][-1] == 9 * 10000 - 20
    assert gen.complex_symmetric_grid(50, 100)[0][-1] == 5 * 5000 - 2 * 150


def test_structure_properties():
    ip, ix, d, rhs = gen.minres_grid_laplacian(8, 8)
    A = sp.csr_matrix((d, ix, ip))
    assert abs(A - A.T).max() == 0                      # sprs::is_symmetric (tests/test_minres.rs:10)
    ip, ix, d, rhs, dg = gen.complex_hermitian_grid(8, 8)
    A = sp.csr_matrix((d, ix, ip))
    assert abs(A - A.conj().T).max() == 0
    assert np.allclose(A @ gen.grid_exact_solution(8, 8), rhs)
    ip, ix, d, rhs, dg = gen.complex_symmetric_grid(8, 8)
    A = sp.csr_matrix((d, ix, ip))
    assert abs(A - A.T).max() == 0
    assert np.allclose(A @ gen.grid_exact_solution(8, 8), rhs)
    assert np.array_equal(A.diagonal(), dg)
    for m in (sp.csr_matrix((d, ix, ip)),):
        assert m.has_sorted_indices                      # TriMat::to_csr order
    ip, ix, d, rhs = gen.poisson3d(7, 6, 5)
    A = sp.csr_matrix((d, ix, ip))
    assert abs(A - A.T).max() == 0 and np.array_equal(A @ np.ones(210), rhs)
    # a z-slab is the same rows of the full matrix
    ip2, ix2, d2, rhs2 = gen.poisson3d(7, 6, 5, 2, 4)
    B = sp.csr_matrix((d2, ix2, ip2), shape=(2 * 42, 210))
    assert abs(B - A[2 * 42:4 * 42]).max() == 0 and np.array_equal(rhs2, rhs[84:168])


def test_splitmix_is_deterministic():
    a = gen.splitmix64(gen.SEED, 4)
    assert a.dtype == np.uint64 and len(set(a.tolist())) == 4
    assert np.array_equal(a, gen.splitmix64(gen.SEED, 4))
    u = gen.uniform(gen.SEED, 1000)
    assert u.min() >= -1 and u.max() < 1 and abs(u.mean()) < 0.1


def test_torch_generators_match_numpy():
    from sprsolve_amd import gen_torch
    ip, ix, d, rhs = gen.poisson3d(9, 8, 7, 2, 6)
    tip, tix, td, trhs = gen_torch.poisson3d(9, 8, 7, 2, 6, device="cpu")
    assert np.array_equal(tip.numpy(), ip) and np.array_equal(tix.numpy(), ix)
    assert np.array_equal(td.numpy(), d) and np.array_equal(trhs.numpy(), rhs)
    ip, ix, d = gen.grid_laplacian_dirichlet(13, 13)
    tip, tix, td, trhs, tdiag = gen_torch.grid_laplacian_dirichlet(13, 13, device="cpu")
    assert np.array_equal(tip.numpy(), ip) and np.array_equal(tix.numpy(), ix) and np.array_equal(td.numpy(), d)
    assert np.array_equal(trhs.numpy(), gen.dirichlet_rhs(13, 13))
    assert np.array_equal(tdiag.numpy(), np.where(np.diff(ip) == 1, 1.0, -4.0))


def test_random_value_poisson3d():
    """cfg-5 pattern with variable coefficients (bench.py's also.cfg5_random_values): keyed on (row, slot), so a
    z-slab of the partitioned matrix carries the same values as the whole; strictly dominant diagonal; the torch twin
    is bit-identical."""
    from sprsolve_amd import gen_torch
    ip, ix, d, rhs = gen.poisson3d(7, 6, 5, values="random")
    ip0, ix0, d0, _ = gen.poisson3d(7, 6, 5)
    assert np.array_equal(ip, ip0) and np.array_equal(ix, ix0) and not np.array_equal(d, d0)
    A = sp.csr_matrix((d, ix, ip))
    off = A - sp.diags(A.diagonal())
    assert np.all(A.diagonal() > abs(off).sum(axis=1).A1)          # strict row dominance
    assert np.allclose(A @ np.ones(210), rhs, rtol=0, atol=1e-14)
    assert np.unique(d).size > 0.9 * d.size                          # no value dictionary to be had
    ip2, ix2, d2, rhs2 = gen.poisson3d(7, 6, 5, 2, 4, values="random")
    assert np.array_equal(d2, d[ip[84]:ip[168]]) and np.array_equal(rhs2, rhs[84:168])
    tip, tix, td, trhs = gen_torch.poisson3d(7, 6, 5, 1, 4, device="cpu", values="random")
    ip3, ix3, d3, rhs3 = gen.poisson3d(7, 6, 5, 1, 4, values="random")
    assert np.array_equal(tip.numpy(), ip3) and np.array_equal(tix.numpy(), ix3)
    assert np.array_equal(td.numpy(), d3) and np.array_equal(trhs.numpy(), rhs3)
    assert np.array_equal(gen.splitmix64_keys(gen.SEED, np.arange(16)), gen.splitmix64(gen.SEED, 16))
