"""Deterministic workload generators (host side, numpy).

Each generator restates — as data, not code — a matrix the reference builds in its own tests
and benches, or a BASELINE.json config.  All return CSR as
``(indptr:int32, indices:int32, data)`` plus whatever right-hand side / diagonal the
reference pairs with it.  Column indices inside a row are ascending wherever the reference
goes through ``sprs::TriMat::to_csr`` and in the literal push order where it builds CSR
arrays directly (benches/bicgstab.rs:54-89 — which also happens to be ascending).
"""
import numpy as np

SEED = 0x5052534F4C5645  # "PRSOLVE"


# ------------------------------------------------------------------ PRNG
def splitmix64(seed, n, stream=0):
    """n counter-based splitmix64 outputs as uint64 (vectorised; state_i = seed + (i+1)*gamma)."""
    with np.errstate(over="ignore"):
        gamma = np.uint64(0x9E3779B97F4A7C15)
        base = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * gamma)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def splitmix64_keys(seed, keys, stream=0):
    """splitmix64 at arbitrary counters: output i = mix(seed + stream*C + (keys[i]+1)*gamma); equals splitmix64(seed, n)
    for keys = arange(n).  Lets a generator key its randomness on (row, slot) so that a row block of a partitioned
    matrix gets the same values as the whole matrix."""
    with np.errstate(over="ignore"):
        gamma = np.uint64(0x9E3779B97F4A7C15)
        base = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)
        z = base + ((np.asarray(keys).astype(np.uint64) + np.uint64(1)) * gamma)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_keys(seed, keys, lo=-1.0, hi=1.0, stream=0):
    u = (splitmix64_keys(seed, keys, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def uniform(seed, n, lo=-1.0, hi=1.0, stream=0):
    """U[lo,hi) doubles from the top 53 bits of splitmix64."""
    u = (splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


# ------------------------------------------------------------------ reference bench / test matrices
def grid_laplacian_dirichlet(rows, cols):
    """benches/bicgstab.rs:54-89 (= tests/test_solvers.rs:74-109, src/main.rs:53-88).

    Interior rows [1, 1, -4, 1, 1] at columns ((i-1)*rows+j, i*rows+j-1, i*rows+j, i*rows+j+1,
    (i+1)*rows+j); border rows are identity (Dirichlet).  Note the reference indexes with
    ``i * rows + j`` (sic); it only ever uses square grids.
    Returns (indptr, indices, data) with float64 data.
    """
    i, j = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    i = i.ravel(); j = j.ravel()
    border = (i == 0) | (i + 1 == rows) | (j == 0) | (j + 1 == cols)
    cnt = np.where(border, 1, 5).astype(np.int64)
    indptr = np.zeros(rows * cols + 1, dtype=np.int64)
    np.cumsum(cnt, out=indptr[1:])
    nnz = int(indptr[-1])
    indices = np.empty(nnz, dtype=np.int64)
    data = np.empty(nnz, dtype=np.float64)
    b = np.nonzero(border)[0]
    indices[indptr[b]] = i[b] * rows + j[b]
    data[indptr[b]] = 1.0
    q = np.nonzero(~border)[0]
    st = indptr[q]
    ii = i[q]; jj = j[q]
    for k, (ci, cj, val) in enumerate(((ii - 1, jj, 1.0), (ii, jj - 1, 1.0), (ii, jj, -4.0),
                                       (ii, jj + 1, 1.0), (ii + 1, jj, 1.0))):
        indices[st + k] = ci * rows + cj
        data[st + k] = val
    return indptr.astype(np.int32), indices.astype(np.int32), data


def dirichlet_rhs(rows, cols, f=lambda i, j: i + j):
    """benches/bicgstab.rs:91-104 set_boundary_condition: rhs = f(i,j) on the border, else 0."""
    i, j = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    border = (i == 0) | (i + 1 == rows) | (j == 0) | (j + 1 == cols)
    rhs = np.where(border, f(i, j).astype(np.float64), 0.0).ravel()
    # the reference writes rhs[i*rows + j]; identical to ravel() for square grids
    return rhs


def _coo_to_csr(n, r, c, v):
    """TriMat::to_csr: sort by (row, col); the generators below never emit duplicates."""
    order = np.lexsort((c, r))
    r = r[order]; c = c[order]; v = v[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, r + 1, 1)
    np.cumsum(indptr, out=indptr)
    return indptr.astype(np.int32), c.astype(np.int32), v


def _grid_stencil(rows, cols, diag_fn, off_fn, val_fn, dtype):
    """Shared skeleton of tests/test_minres.rs:76-120 and tests/test_complex_solve*.rs:
    diagonal + 4 neighbours where they exist; returns COO triplets and the per-row list of
    (neighbour grid coords) in the reference's insertion order (diag, up, left, down, right)."""
    i, j = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    i = i.ravel(); j = j.ravel()
    vid = i * cols + j
    R = [vid]; Cc = [vid]; V = [diag_fn(i, j).astype(dtype)]
    terms = [(np.ones_like(vid, dtype=bool), i, j)]
    for di, dj in ((-1, 0), (0, -1), (1, 0), (0, 1)):
        ni = i + di; nj = j + dj
        ok = (ni >= 0) & (ni < rows) & (nj >= 0) & (nj < cols)
        tid = ni * cols + nj
        R.append(vid[ok]); Cc.append(tid[ok]); V.append(off_fn(vid[ok], tid[ok]).astype(dtype))
        terms.append((ok, ni, nj))
    return i, j, vid, R, Cc, V, terms


def minres_grid_laplacian(rows, cols):
    """tests/test_minres.rs:76-120: symmetric 5-point Laplacian (-4 diag, +1 neighbours), the
    out-of-grid neighbours moved to the rhs with boundary value bv(r,c)=r+c."""
    i, j, vid, R, Cc, V, terms = _grid_stencil(
        rows, cols, lambda a, b: np.full(a.shape, -4.0), lambda r, c: np.ones(r.shape), None, np.float64)
    n = rows * cols
    rhs = np.zeros(n)
    for (ok, ni, nj) in terms[1:]:
        rhs[~ok] -= (ni[~ok] + nj[~ok]).astype(np.float64)
    indptr, indices, data = _coo_to_csr(n, np.concatenate(R), np.concatenate(Cc), np.concatenate(V))
    return indptr, indices, data, rhs


def minres_simple_diag(rows, cols):
    """tests/test_minres.rs:62-74: diag(2,4,...,2n), rhs = 1..n  =>  exact solution 0.5."""
    n = rows * cols
    indptr = np.arange(n + 1, dtype=np.int32)
    indices = np.arange(n, dtype=np.int32)
    data = (np.arange(n) + 1.0) * 2.0
    rhs = np.arange(n) + 1.0
    return indptr, indices, data, rhs


def _complex_rhs(n, i, j, V, terms):
    """rhs[vid] = sum over the row, in insertion order, of c * val(ni, nj) with val = ni + nj*i
    (tests/test_complex_solve.rs:117-147) — accumulated in exactly that order."""
    rhs = np.zeros(n, dtype=np.complex128)
    pos = 0
    for (ok, ni, nj), v in zip(terms, V):
        val = ni[ok].astype(np.float64) + 1j * nj[ok].astype(np.float64)
        idx = np.nonzero(ok)[0]
        # complex multiply exactly as num-complex: (ac-bd) + (ad+bc)i
        re = v.real * val.real - v.imag * val.imag
        im = v.real * val.imag + v.imag * val.real
        rhs[idx] = rhs[idx] + (re + 1j * im)
        pos += 1
    return rhs


def complex_hermitian_grid(rows, cols):
    """tests/test_complex_solve.rs:95-151 / :153-214: Hermitian; diag -3-i (real), off-diagonal
    1+2.5i below the diagonal, 1-2.5i above; rhs = A * (i + j*1i).  Returns
    (indptr, indices, data, rhs, diag_for_precond) with diag_for_precond = 3+i (real, :177)."""
    off = lambda r, c: np.where(r > c, 1.0 + 2.5j, 1.0 - 2.5j)
    i, j, vid, R, Cc, V, terms = _grid_stencil(
        rows, cols, lambda a, b: (-3.0 - a) + 0j, off, None, np.complex128)
    n = rows * cols
    rhs = _complex_rhs(n, i, j, V, terms)
    indptr, indices, data = _coo_to_csr(n, np.concatenate(R), np.concatenate(Cc), np.concatenate(V))
    diag = (3.0 + i).astype(np.float64)
    return indptr, indices, data, rhs, diag


def complex_symmetric_grid(rows, cols):
    """tests/test_complex_solve2.rs:35-96: complex-symmetric; diag (-2-i) + (-2-j)*1i, every
    off-diagonal 1-2.5i; rhs = A * (i + j*1i); complex Jacobi diagonal = the matrix diagonal."""
    off = lambda r, c: np.full(r.shape, 1.0 - 2.5j)
    dg = lambda a, b: (-2.0 - a) + 1j * (-2.0 - b)
    i, j, vid, R, Cc, V, terms = _grid_stencil(rows, cols, dg, off, None, np.complex128)
    n = rows * cols
    rhs = _complex_rhs(n, i, j, V, terms)
    indptr, indices, data = _coo_to_csr(n, np.concatenate(R), np.concatenate(Cc), np.concatenate(V))
    diag = dg(i, j).astype(np.complex128)
    return indptr, indices, data, rhs, diag


def grid_exact_solution(rows, cols):
    """x*[i*cols+j] = i + j*1i — the solution the complex tests imply (rhs := A*val)."""
    i, j = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    return (i + 1j * j).ravel().astype(np.complex128)


# ------------------------------------------------------------------ BASELINE configs
def random_tridiagonal(n, seed=SEED):
    """cfg 1(ii): strictly diagonally dominant random tridiagonal; off-diagonals U(-1,1),
    diag = 2 + |l| + |u|, rhs U(-1,1)."""
    lo = uniform(seed, n, stream=1); up = uniform(seed, n, stream=2)
    lo[0] = 0.0; up[-1] = 0.0
    dg = 2.0 + np.abs(lo) + np.abs(up)
    cnt = np.full(n, 3, dtype=np.int64); cnt[0] = 2; cnt[-1] = 2
    indptr = np.zeros(n + 1, dtype=np.int64); np.cumsum(cnt, out=indptr[1:])
    indices = np.empty(indptr[-1], dtype=np.int64); data = np.empty(indptr[-1])
    r = np.arange(n)
    st = indptr[:-1]
    has_lo = r > 0
    indices[st[has_lo]] = r[has_lo] - 1; data[st[has_lo]] = lo[has_lo]
    dpos = st + has_lo
    indices[dpos] = r; data[dpos] = dg
    has_up = r < n - 1
    indices[dpos[has_up] + 1] = r[has_up] + 1; data[dpos[has_up] + 1] = up[has_up]
    rhs = uniform(seed, n, stream=3)
    return indptr.astype(np.int32), indices.astype(np.int32), data, rhs


def symmetric_banded(n, hbw=4, seed=SEED):
    """cfg 3: symmetric banded, half-bandwidth hbw; a[i,i+k] = a[i+k,i] = U(-1,1),
    diag = 1 + sum |off-diagonal of the row| (strictly dominant => SPD); rhs U(-1,1)."""
    offs = [uniform(seed, n, stream=10 + k) for k in range(1, hbw + 1)]  # offs[k-1][i] = a[i, i+k]
    R = []; Cc = []; V = []
    absrow = np.zeros(n)
    r = np.arange(n)
    for k in range(1, hbw + 1):
        a = offs[k - 1][: n - k]
        R += [r[: n - k], r[k:]]; Cc += [r[k:], r[: n - k]]; V += [a, a]
        absrow[: n - k] += np.abs(a); absrow[k:] += np.abs(a)
    R.append(r); Cc.append(r); V.append(1.0 + absrow)
    indptr, indices, data = _coo_to_csr(n, np.concatenate(R), np.concatenate(Cc), np.concatenate(V))
    rhs = uniform(seed, n, stream=30)
    return indptr, indices, data, rhs


def poisson3d(nx, ny, nz, z0=0, z1=None, index_dtype=np.int32, values="poisson", seed=SEED):
    """cfg 5: 7-point 3-D Poisson on an nx*ny*nz grid (x fastest), diag +6, neighbours -1,
    truncated at the faces.  Returns the CSR row block for planes [z0, z1) with GLOBAL column
    indices, and rhs = A*1 (row sums) for those rows.  Columns ascending within a row.
    values="random": the same pattern with variable coefficients — off-diagonal (row g, slot j) = U(-1,1) keyed on
    g*7 + j (stream 50), diagonal = 1 + sum |off-diagonals of the row| (strictly dominant), rhs = A*1 summed left to
    right over the row's entries in column order."""
    if z1 is None:
        z1 = nz
    plane = nx * ny
    g = np.arange(z0 * plane, z1 * plane, dtype=np.int64)
    x = g % nx; y = (g // nx) % ny; z = g // plane
    cand = np.stack([g - plane, g - nx, g - 1, g, g + 1, g + nx, g + plane], axis=1)
    ok = np.stack([z > 0, y > 0, x > 0, np.ones_like(g, dtype=bool), x < nx - 1, y < ny - 1, z < nz - 1], axis=1)
    cnt = ok.sum(axis=1)
    indptr = np.zeros(g.size + 1, dtype=np.int64); np.cumsum(cnt, out=indptr[1:])
    indices = cand[ok]
    if values == "random":
        vals = uniform_keys(seed, g[:, None] * 7 + np.arange(7)[None, :], stream=50)
        vals = np.where(ok, vals, 0.0)
        vals[:, 3] = 0.0
        acc = np.zeros(g.size)
        for j in range(7):                      # explicit slot order (numpy's axis sum is free to re-associate)
            acc = acc + np.abs(vals[:, j])
        vals[:, 3] = 1.0 + acc
        rhs = np.zeros(g.size)
        for j in range(7):                      # left-to-right row sum in column order (absent slots add 0.0 exactly)
            rhs = rhs + vals[:, j]
    else:
        vals = np.broadcast_to(np.array([-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0]), cand.shape)
        rhs = 6.0 - (cnt - 1).astype(np.float64)
    data = np.ascontiguousarray(vals[ok])
    return indptr.astype(index_dtype if indptr[-1] < 2**31 else np.int64), indices.astype(index_dtype), data, rhs
