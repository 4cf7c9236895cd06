"""MatrixMarket ingest against scipy.io and a round trip of the reference's test matrices."""
import numpy as np
import scipy.io
import scipy.sparse as sp

from sprsolve_amd import gen, io


def test_round_trip_real_and_complex(tmp_path):
    for name, (ip, ix, d) in {
        "lap": gen.grid_laplacian_dirichlet(6, 6),
        "herm": gen.complex_hermitian_grid(5, 4)[:3],
    }.items():
        n = ip.size - 1
        p = str(tmp_path / (name + ".mtx"))
        io.write_matrix_market(p, (n, n), ip, ix, d)
        shape, ip2, ix2, d2 = io.read_matrix_market(p)
        assert shape == (n, n) and np.array_equal(ip2, ip) and np.array_equal(ix2, ix) and np.array_equal(d2, d)
        M = scipy.io.mmread(p).tocsr()
        assert abs(M - sp.csr_matrix((d, ix, ip), shape=(n, n))).max() == 0


def test_symmetric_and_pattern_files(tmp_path):
    A = sp.random(30, 30, density=0.1, random_state=1, format="coo")
    S = sp.coo_matrix(A + A.T)
    p = str(tmp_path / "sym.mtx")
    scipy.io.mmwrite(p, S, symmetry="symmetric")
    shape, ip, ix, d = io.read_matrix_market(p)
    assert abs(sp.csr_matrix((d, ix, ip), shape=shape) - S.tocsr()).max() < 1e-15
    H = sp.coo_matrix(A + 1j * sp.triu(A, 1) - 1j * sp.triu(A, 1).T + A.T)
    p = str(tmp_path / "herm.mtx")
    scipy.io.mmwrite(p, H, symmetry="hermitian")
    shape, ip, ix, d = io.read_matrix_market(p)
    assert abs(sp.csr_matrix((d, ix, ip), shape=shape) - H.tocsr()).max() < 1e-15
    with open(str(tmp_path / "pat.mtx"), "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern general\n% comment\n3 3 3\n1 1\n2 3\n2 3\n")
    shape, ip, ix, d = io.read_matrix_market(str(tmp_path / "pat.mtx"))
    assert ip.tolist() == [0, 1, 2, 2] and ix.tolist() == [0, 2] and d.tolist() == [1.0, 2.0]   # duplicate summed
