"""The dictionary-compressed SpMV stream (csrc/spmv_dict.hip) against the plain CSR kernel and the oracle:
bit-identical y in all three formats, correct fall-back for matrices that do not qualify."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DTYPES = [np.float64, np.complex128, np.float32, np.complex64]
IDS = ["f64", "c64", "f32", "c32"]


@pytest.fixture(scope="module")
def sa():
    import sprsolve_amd
    from sprsolve_amd import _lib
    _lib.lib()
    sprsolve_amd.default_ctx(0)
    return sprsolve_amd


@pytest.fixture(autouse=True)
def _restore_knob(sa):
    yield
    sa.default_ctx(0).set("spmv_dict", -1)
    sa.default_ctx(0).set("spmv_wide", -1)
    sa.default_ctx(0).set("spmv_triple", -1)
    sa.default_ctx(0).set("spmv_seam", -1)
    sa.default_ctx(0).set("spmv_tile", -1)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def rand_vec(n, dtype, seed):
    rng = np.random.default_rng(seed)
    if np.dtype(dtype).kind == "c":
        return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(dtype)
    return rng.uniform(-1, 1, n).astype(dtype)


def matrices(dtype):
    from sprsolve_amd import gen
    out = {}
    ip, ix, d, _ = gen.poisson3d(13, 11, 9)
    out["poisson3d"] = (ip, ix, d.astype(dtype))                       # 7 offsets, 2 values
    ip, ix, d = gen.grid_laplacian_dirichlet(37, 37)
    out["grid2d"] = (ip, ix, d.astype(dtype))                          # identity rows + 5-point rows
    ip, ix, d, _ = gen.symmetric_banded(5000, 4)
    dd = d.astype(dtype)
    if np.dtype(dtype).kind == "c":
        dd = dd * (1 + 0.5j)
    out["banded_random_values"] = (ip, ix, dd)                          # 9 offsets, n distinct values
    return out


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("name", ["poisson3d", "grid2d", "banded_random_values"])
def test_formats_bit_identical(sa, oracle, dtype, name):
    ctx = sa.default_ctx(0)
    indptr, indices, data = matrices(dtype)[name]
    n = indptr.size - 1
    x = rand_vec(n, dtype, 7)
    ref = oracle.spmv(indptr, indices, data, x)
    real = np.dtype(dtype).kind != "c"
    got = {}
    ctx.set("spmv_wide", 0)           # the 64-row kernels: even the fused dot partials are grouped identically
    for knob in (0, 1, 2, -1):
        ctx.set("spmv_dict", knob)
        A = sa.HipCsr.new((n, n), indptr, indices, data)
        mode, n_off, n_val = A.stream_format()
        if knob == 0:
            assert mode == 0
        elif knob == 1:
            assert mode == 1 and n_val == 0
        else:
            few_values = name != "banded_random_values"
            # auto (-1): pair codes wherever the (offset, value) pairs are few — complex matrices through the row-value slot (their
            # offset-0 entries are kept per row, csrc/spmv_dict.hip cpair stage); else offset codes for every real matrix that has
            # them and the plain stream for complex ones (offset codes alone do not pay there: 17 instead of 20 B/nnz against the
            # lane-per-row layout)
            assert mode == (2 if few_values else (1 if (knob == 2 or real) else 0)), (mode, n_off, n_val)
            if few_values:
                # distinct (offset, value) pairs; complex: the pairs off the diagonal + 1 for the row-value slot
                assert n_val == (7 if name == "poisson3d" else (6 if real else 5))
        if knob != 0:
            assert n_off == {"poisson3d": 7, "grid2d": 5, "banded_random_values": 9}[name]
        y = np.full(n, 3.0, dtype=dtype)
        A.mul_vec(x, y)
        assert np.array_equal(bits(y), bits(ref)), (name, knob)       # the reference fold, bit for bit
        y2 = np.zeros(n, dtype=dtype)
        dot = A.mul_vec_dot(x, y2)
        assert np.array_equal(bits(y2), bits(ref))
        got[knob] = dot
    # the fused dot epilogue reduces the same y the same way in every format
    assert got[0] == got[1] == got[2] == got[-1]


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64], ids=["c64", "c32"])
def test_complex_pair_codes_row_value_slot(sa, oracle, dtype):
    """Complex pair codes (csrc/spmv_dict.hip, cpair stage): the offset-0 entry of a row has the reserved code 255 = the row's own
    value.  Rows WITHOUT a diagonal entry, a diagonal that is different in every row, conjugate-transposed use (CSMINRES' gather),
    more than 255 off-diagonal pairs (no pair codes), and a row that holds TWO entries at offset 0 (the slot cannot hold both: the
    handle keeps the offset-code stream) — y bit for bit the reference fold every time."""
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    ip, ix, d, _, _ = gen.complex_symmetric_grid(40, 50)          # tests/test_complex_solve2.rs:35-96's operator: 5 offsets
    d = d.astype(dtype)
    n = ip.size - 1
    x = rand_vec(n, dtype, 11)
    rng = np.random.default_rng(8)

    def check(indptr, indices, data, want_mode, label):
        A = sa.HipCsr.new((n, n), indptr, indices, data)
        assert A.stream_format()[0] == want_mode, (label, A.stream_format())
        ref = oracle.spmv(indptr, indices, data, x)
        y = np.full(n, 3.0, dtype=dtype)
        A.mul_vec(x, y)
        assert np.array_equal(bits(y), bits(ref)), label
        y2 = np.zeros(n, dtype=dtype)
        A.mul_vec_dot(x, y2)
        assert np.array_equal(bits(y2), bits(ref)), label
        return A

    try:
        ctx.set("spmv_dict", -1)
        check(ip, ix, d, 2, "the grid")
        # a diagonal that differs row by row (the slot holds n distinct values), some diagonals -0.0 / denormal / inf-free specials
        dd = d.copy()
        diag = np.nonzero(ix == np.repeat(np.arange(n), np.diff(ip)))[0]
        dd[diag] = (rng.uniform(-3, 3, diag.size) + 1j * rng.uniform(-3, 3, diag.size)).astype(dtype)
        dd[diag[:3]] = np.array([-0.0 + 0j, 5e-324 if dtype == np.complex128 else 1e-45, 1j], dtype=dtype)
        check(ip, ix, dd, 2, "row-by-row diagonal")
        # rows without a diagonal entry: drop every third diagonal
        keep = np.ones(ix.size, bool); keep[diag[::3]] = False
        ip2 = np.concatenate([[0], np.cumsum(np.add.reduceat(keep.astype(np.int64), ip[:-1]))]).astype(np.int32)
        check(ip2, ix[keep], dd[keep], 2, "missing diagonals")
        # more than 255 distinct off-diagonal pairs: no pair codes, the plain stream under the automatic policy
        dr = dd.copy()
        off = np.setdiff1d(np.arange(ix.size), diag)
        dr[off] = (rng.uniform(-1, 1, off.size) + 1j * rng.uniform(-1, 1, off.size)).astype(dtype)
        check(ip, ix, dr, 0, "random off-diagonals")
        # a duplicated diagonal entry in one row (CSR with a repeated column: the fold adds both products, in order)
        r = n // 2
        k = int(diag[r])
        ip3 = ip.copy(); ip3[r + 1:] += 1
        ix3 = np.insert(ix, k + 1, ix[k]); d3 = np.insert(dd, k + 1, dtype(0.25 - 2j))
        A = check(ip3, ix3, d3, 0, "two entries at offset 0")
        ctx.set("spmv_dict", 2)                                   # forced: still no pair codes for this handle — offset codes
        check(ip3, ix3, d3, 1, "two entries at offset 0, knob 2")
    finally:
        ctx.set("spmv_dict", -1)


def test_fallback_too_many_offsets(sa, oracle):
    import scipy.sparse as sp
    n = 4000
    rng = np.random.default_rng(2)
    M = (sp.random(n, n, density=0.003, random_state=rng, format="csr") + sp.eye(n)).tocsr()
    M.sort_indices()
    A = sa.HipCsr.new((n, n), M.indptr, M.indices, M.data)
    assert A.stream_format() == (0, 0, 0)
    x = rand_vec(n, np.float64, 3); y = np.zeros(n)
    A.mul_vec(x, y)
    assert np.array_equal(bits(y), bits(oracle.spmv(M.indptr, M.indices, M.data, x)))


def test_fallback_long_rows(sa, oracle):
    """A row longer than LONG_ROW (96) takes the wavefront-per-row path, which reads the plain stream."""
    import scipy.sparse as sp
    n = 600
    D = sp.diags([np.ones(n - abs(k)) for k in range(-60, 61)], list(range(-60, 61)), format="csr")   # 121 per row
    A = sa.HipCsr.new((n, n), D.indptr, D.indices, D.data)
    assert A.stream_format()[0] == 0
    x = rand_vec(n, np.float64, 4); y = np.zeros(n)
    A.mul_vec(x, y)
    ref = oracle.spmv(D.indptr, D.indices, D.data, x)
    assert np.allclose(y, ref, rtol=0, atol=1e-13 * 121)


def test_value_bit_patterns_survive(sa, oracle):
    """Values are matched by bit pattern: -0.0 and +0.0, denormals and a NaN payload are separate dictionary
    entries and reproduce the plain kernel's bits."""
    n = 300
    indptr = np.arange(0, 3 * n + 1, 3, dtype=np.int32)
    cols = np.stack([(np.arange(n) - 1) % n, np.arange(n), (np.arange(n) + 1) % n], axis=1)
    cols.sort(axis=1)
    special = np.array([0.0, -0.0, 5e-324, -2.5, np.inf, 1.0, 3.0], dtype=np.float64)
    rng = np.random.default_rng(9)
    data = special[rng.integers(0, special.size, 3 * n)]
    x = rand_vec(n, np.float64, 5)
    ctx = sa.default_ctx(0)
    ys = []
    for knob in (0, 2):
        ctx.set("spmv_dict", knob)
        A = sa.HipCsr.new((n, n), indptr, cols.ravel().astype(np.int32), data)
        if knob == 2:
            assert A.stream_format()[0] == 2 and special.size <= A.stream_format()[2] <= 5 * special.size
        y = np.zeros(n)
        A.mul_vec(x, y)
        ys.append(y)
    assert np.array_equal(bits(ys[0]), bits(ys[1]))
    assert np.array_equal(bits(ys[0]), bits(oracle.spmv(indptr, cols.ravel(), data, x)))


@pytest.mark.parametrize("span", [20, 48], ids=["pairs_fit", "pairs_overflow"])
def test_ragged_and_empty_rows(sa, oracle, span):
    """Empty rows, rows of every length up to LONG_ROW, block boundaries at every 4-byte phase of the code stream.
    span 20: 41 offsets x 4 values <= 256 pairs (pair codes); span 48: too many pairs -> offset codes + values."""
    rng = np.random.default_rng(12)
    n = 2000
    lens = rng.integers(0, 13, n)
    lens[::97] = 0
    lens[5::211] = 2 * span - 6
    rows = []
    for r, l in enumerate(lens):
        c = r + rng.choice(np.arange(-span, span + 1), l, replace=False)
        rows.append(np.sort(c[(c >= 0) & (c < n)]))
    indptr = np.zeros(n + 1, dtype=np.int32); np.cumsum([len(c) for c in rows], out=indptr[1:])
    cols = np.concatenate(rows).astype(np.int32)
    data = rng.choice(np.array([1.0, -2.0, 0.5, 4.0]), cols.size)
    x = rand_vec(n, np.float64, 6)
    ctx = sa.default_ctx(0)
    ref = oracle.spmv(indptr, cols, data, x)
    for knob in (0, 1, 2):
        ctx.set("spmv_dict", knob)
        A = sa.HipCsr.new((n, n), indptr, cols, data)
        assert A.stream_format()[0] == (knob if span == 20 else min(knob, 1))
        y = np.full(n, 9.0)
        A.mul_vec(x, y)
        assert np.array_equal(bits(y), bits(ref)), knob


@pytest.mark.parametrize("knob", [0, 1, 2])
def test_solver_same_iterates_in_every_format(sa, oracle, knob):
    """BiCGStab on the 3-D Poisson problem: with the 64-row kernels the SpMV and its fused partials are
    bit-identical across formats, so the whole solve (iteration count, residual, x) is.  (The two-rows-per-lane
    kernel groups the dot partials differently: see test_wide_kernel.)"""
    from sprsolve_amd import gen
    indptr, indices, data, rhs = gen.poisson3d(24, 20, 16)
    n = rhs.size
    ctx = sa.default_ctx(0)
    ctx.set("spmv_wide", 0)
    res = {}
    for k in (0, knob):
        ctx.set("spmv_dict", k)
        A = sa.HipCsr.new((n, n), indptr, indices, data)
        x = np.zeros(n)
        its, r = sa.BiCGStab.new(A, n).solve(rhs, x, 500, 1e-10)
        res[k] = (its, r, x)
    assert res[0][0] == res[knob][0] and res[0][1] == res[knob][1]
    assert np.array_equal(bits(res[0][2]), bits(res[knob][2]))
    assert np.max(np.abs(res[knob][2] - 1.0)) < 1e-7


@pytest.mark.parametrize("index_dtype", [np.int32, np.int64, np.uint32], ids=["i32", "i64", "u32"])
def test_csc_and_wide_indices_take_the_compressed_stream(sa, oracle, index_dtype):
    """CSC input (converted once, mat.rs:130-142) and 64-bit / unsigned index arrays end up in the same code stream."""
    import scipy.sparse as sp
    from sprsolve_amd import gen
    ip, ix, d, _ = gen.poisson3d(10, 9, 8)
    n = ip.size - 1
    M = sp.csr_matrix((d, ix, ip), shape=(n, n))
    C = M.tocsc()
    x = rand_vec(n, np.float64, 8)
    ref = oracle.spmv(ip, ix, d, x)
    A = sa.HipCsr.new((n, n), ip.astype(index_dtype), ix.astype(index_dtype), d)
    assert A.stream_format() == (2, 7, 7)
    y = np.zeros(n); A.mul_vec(x, y)
    assert np.array_equal(bits(y), bits(ref))
    B = sa.HipCsr.new((n, n), C.indptr.astype(index_dtype), C.indices.astype(index_dtype), C.data, storage="CSC")
    assert B.stream_format()[0] == 2
    y2 = np.zeros(n); B.mul_vec(x, y2)
    # symmetric matrix, columns ascending: the CSC scatter order equals the CSR row order here
    assert np.array_equal(bits(y2), bits(oracle.spmv_csc(n, C.indptr, C.indices, C.data, x)))


def test_minres_and_csminres_on_compressed_streams(sa, oracle):
    """MINRES (real, pair codes) and CSMINRES (complex, forced offset codes with the conjugated gather) give the
    plain stream's iterates bit for bit."""
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    ctx.set("spmv_wide", 0)           # 64-row kernels: identical dot partials, hence identical iterates
    ip, ix, d, rhs = gen.minres_grid_laplacian(24, 24)
    n = rhs.size
    out = {}
    for knob in (0, 2):
        ctx.set("spmv_dict", knob)
        A = sa.HipCsr.new((n, n), ip, ix, d)
        assert A.stream_format()[0] == knob
        x = np.zeros(n)
        out[knob] = (sa.MinRes.new(A, n).solve(rhs, x, 2000, 1e-10), x)
    assert out[0][0] == out[2][0] and np.array_equal(bits(out[0][1]), bits(out[2][1]))
    ip, ix, d, rhs, _ = gen.complex_symmetric_grid(12, 12)
    n = rhs.size
    out = {}
    for knob in (0, 1):
        ctx.set("spmv_dict", knob)
        A = sa.HipCsr.new((n, n), ip, ix, d)
        assert A.stream_format()[0] == knob
        x = np.zeros(n, dtype=np.complex128)
        out[knob] = (sa.CSMinRes.new(A, n).solve(rhs, x, 2000, 1e-10), x)
    assert out[0][0] == out[1][0] and np.array_equal(bits(out[0][1]), bits(out[1][1]))


@pytest.mark.parametrize("shape", [(13, 11, 9), (64, 3, 5), (7, 1, 1), (500, 2, 3)], ids=lambda s: "x".join(map(str, s)))
def test_wide_kernel_bit_identical_y(sa, oracle, shape):
    """The two-rows-per-lane kernel (f64 pair codes): y bit-identical to the reference fold, for block tails with an odd
    row count, rows at the matrix end (16-byte loads clamped to x[ncols-2]), and slots where the two rows of a lane
    disagree (grid boundaries); the fused dot agrees with the 64-row kernel's to reduction-order tolerance."""
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    indptr, indices, data, _ = gen.poisson3d(*shape)
    n = indptr.size - 1
    x = rand_vec(n, np.float64, 21)
    ref = oracle.spmv(indptr, indices, data, x)
    dots = {}
    for wide in (0, 1):
        ctx.set("spmv_wide", wide)
        A = sa.HipCsr.new((n, n), indptr, indices, data)
        assert A.stream_format()[0] == 2
        y = np.full(n, 5.0)
        A.mul_vec(x, y)
        assert np.array_equal(bits(y), bits(ref)), wide
        y2 = np.zeros(n)
        dots[wide] = A.mul_vec_dot(x, y2)
        assert np.array_equal(bits(y2), bits(ref))
    assert abs(dots[0] - dots[1]) <= 1e-13 * np.sum(np.abs(x * ref))


def test_wide_kernel_irregular_rows(sa, oracle):
    """Ragged rows (lengths 0..12, a few of 34), pair codes: the lanes' two rows rarely agree on a slot's offset."""
    rng = np.random.default_rng(31)
    n = 3001                                   # odd: the last lane owns a single row
    lens = rng.integers(0, 13, n)
    lens[::97] = 0
    lens[5::211] = 34
    rows = []
    for r, l in enumerate(lens):
        c = r + rng.choice(np.arange(-20, 21), l, replace=False)
        rows.append(np.sort(c[(c >= 0) & (c < n)]))
    indptr = np.zeros(n + 1, dtype=np.int32); np.cumsum([len(c) for c in rows], out=indptr[1:])
    cols = np.concatenate(rows).astype(np.int32)
    data = rng.choice(np.array([1.0, -2.0, 0.5, 4.0]), cols.size)
    x = rand_vec(n, np.float64, 6)
    ref = oracle.spmv(indptr, cols, data, x)
    ctx = sa.default_ctx(0)
    for wide in (0, 1):
        ctx.set("spmv_wide", wide)
        A = sa.HipCsr.new((n, n), indptr, cols, data)
        assert A.stream_format()[0] == 2
        y = np.full(n, 9.0)
        A.mul_vec(x, y)
        assert np.array_equal(bits(y), bits(ref)), wide


def test_wide_kernel_solve(sa, oracle):
    """The full solve on the wide kernel: same iteration count as the 64-row kernel up to reduction noise, exact solution."""
    from sprsolve_amd import gen
    indptr, indices, data, rhs = gen.poisson3d(24, 20, 16)
    n = rhs.size
    ctx = sa.default_ctx(0)
    out = {}
    for wide in (0, 1):
        ctx.set("spmv_wide", wide)
        A = sa.HipCsr.new((n, n), indptr, indices, data)
        x = np.zeros(n)
        out[wide] = (sa.BiCGStab.new(A, n).solve(rhs, x, 500, 1e-10), x)
    assert abs(out[0][0][0] - out[1][0][0]) <= 2
    assert np.max(np.abs(out[1][1] - 1.0)) < 1e-7


def test_fuzz_small_matrices(sa, oracle):
    """200 small random matrices (1..300 rows, banded patterns with a handful of offsets and values so that the
    pair-code stream and the two-rows-per-lane kernel are taken, empty rows, single rows, odd sizes): y bit-identical
    to the reference fold on every stream / kernel."""
    rng = np.random.default_rng(2026)
    ctx = sa.default_ctx(0)
    vals = np.array([1.0, -1.0, 0.5, 2.0, -3.25])
    for trial in range(200):
        n = int(rng.integers(1, 301))
        offs = np.unique(rng.integers(-min(n - 1, 9), min(n - 1, 9) + 1, size=int(rng.integers(1, 8))))
        keep_p = rng.uniform(0.3, 1.0)
        rows = []
        for r in range(n):
            c = r + offs
            c = c[(c >= 0) & (c < n)]
            c = c[rng.uniform(size=c.size) < keep_p] if rng.uniform() < 0.7 else c
            rows.append(c)
        indptr = np.zeros(n + 1, dtype=np.int32); np.cumsum([len(c) for c in rows], out=indptr[1:])
        if indptr[-1] == 0:
            continue
        cols = np.concatenate(rows).astype(np.int32)
        data = vals[rng.integers(0, vals.size, cols.size)]
        x = rng.uniform(-1, 1, n)
        ref = oracle.spmv(indptr, cols, data, x)
        for knob, wide in ((0, 0), (1, 0), (2, 0), (2, 1)):
            ctx.set("spmv_dict", knob); ctx.set("spmv_wide", wide)
            A = sa.HipCsr.new((n, n), indptr, cols, data)
            y = np.full(n, 7.0)
            A.mul_vec(x, y)
            assert np.array_equal(bits(y), bits(ref)), (trial, n, knob, wide, A.stream_format())
            y2 = np.zeros(n)
            d = A.mul_vec_dot(x, y2)
            assert np.array_equal(bits(y2), bits(ref))
            assert abs(d - float(np.dot(x, ref))) <= 1e-12 * max(1.0, float(np.sum(np.abs(x * ref))))


def test_dictionary_limits(sa, oracle):
    """Exactly 256 distinct offsets still compress, 257 fall back to the plain stream; a value whose bit pattern is the
    collector's EMPTY marker (all ones, a NaN) disables the value dictionary instead of being mis-keyed."""
    ctx = sa.default_ctx(0)
    ctx.set("spmv_dict", 2)
    n = 4000
    for n_off, want in ((256, True), (257, False)):
        rows = np.arange(n)
        offs = np.arange(n_off) - n_off // 2
        # row r holds the 3 offsets (r, r+1, r+2) mod n_off: every offset occurs, each row is short
        cols = rows[:, None] + offs[(rows[:, None] + np.arange(3)[None, :]) % n_off]
        cols = np.sort(np.clip(cols, 0, n - 1), axis=1)
        keep = np.concatenate([np.ones((n, 1), bool), cols[:, 1:] != cols[:, :-1]], axis=1)      # drop clipped duplicates
        indptr = np.zeros(n + 1, dtype=np.int32); np.cumsum(keep.sum(axis=1), out=indptr[1:])
        ci = cols[keep].astype(np.int32)
        data = np.ones(ci.size)
        got_offs = np.unique(ci - np.repeat(rows, keep.sum(axis=1))).size
        A = sa.HipCsr.new((n, n), indptr, ci, data)
        mode, no, npair = A.stream_format()
        assert (mode != 0) == (got_offs <= 256), (n_off, got_offs, mode)
        x = rand_vec(n, np.float64, 1); y = np.zeros(n)
        A.mul_vec(x, y)
        assert np.array_equal(bits(y), bits(oracle.spmv(indptr, ci, data, x)))
    # the all-ones NaN
    n = 64
    indptr = np.arange(0, 2 * n + 1, 2, dtype=np.int32)
    ci = np.stack([np.arange(n), (np.arange(n) + 1) % n], axis=1); ci.sort(axis=1)
    data = np.tile(np.array([1.5, -2.0]), n)
    data[7] = np.frombuffer(np.uint64(0xFFFFFFFFFFFFFFFF).tobytes(), dtype=np.float64)[0]
    A = sa.HipCsr.new((n, n), indptr, ci.ravel().astype(np.int32), data)
    assert A.stream_format()[0] == 1            # offsets compress, values do not
    x = rand_vec(n, np.float64, 2); y = np.zeros(n)
    A.mul_vec(x, y)
    ctx.set("spmv_dict", 0)
    B = sa.HipCsr.new((n, n), indptr, ci.ravel().astype(np.int32), data)
    y0 = np.zeros(n); B.mul_vec(x, y0)
    assert np.array_equal(bits(y), bits(y0))


@pytest.mark.parametrize("kind", ["tridiagonal", "poisson3d_long_lines", "poisson2d"])
def test_uniform_blocks(sa, oracle, kind):
    """Blocks whose rows all repeat one code sequence are multiplied from a scalar pattern (no codes, no row_ptr):
    the flag is found where it should be, and y is bit-identical with and without that path."""
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    if kind == "tridiagonal":
        n = 5001
        indptr = np.zeros(n + 1, dtype=np.int32)
        lens = np.full(n, 3); lens[0] = lens[-1] = 2
        np.cumsum(lens, out=indptr[1:])
        cols = np.concatenate([[0, 1]] + [[i - 1, i, i + 1] for i in range(1, n - 1)] + [[n - 2, n - 1]]).astype(np.int32)
        data = np.tile(np.array([-1.0, 2.0, -1.0]), n)[1:-1].copy()
        expect_uniform = (n - 2) // 128 - 1            # at least: every block strictly inside the interior run
    elif kind == "poisson3d_long_lines":
        indptr, cols, data, _ = gen.poisson3d(300, 4, 3)
        n = indptr.size - 1
        expect_uniform = 4 * 3                          # >= one fully interior 128-row block per 300-row line
    else:
        indptr, cols, data = gen.grid_laplacian_dirichlet(200, 200)
        n = 200 * 200
        expect_uniform = 90                             # interior rows come in runs of 198 between the Dirichlet rows: ~1 block in 3
    x = rand_vec(n, np.float64, 17)
    ref = oracle.spmv(indptr, cols, data, x)
    ys = {}
    for uni in (1, 0):
        ctx.set("spmv_uniform", uni)
        A = sa.HipCsr.new((n, n), indptr, cols, data)
        assert A.stream_format()[0] == 2
        nb, nu = A.wide_blocks()
        assert nb == (n + 127) // 128
        if uni:
            assert expect_uniform <= nu < nb, (nb, nu)
        else:
            assert nu == 0
        y = np.full(n, 4.0)
        A.mul_vec(x, y)
        assert np.array_equal(bits(y), bits(ref)), (kind, uni)
        y2 = np.zeros(n)
        ys[uni] = A.mul_vec_dot(x, y2)
        assert np.array_equal(bits(y2), bits(ref))
    assert ys[0] == ys[1]           # same kernel, same lane/row grouping: even the fused dot is identical
    ctx.set("spmv_uniform", -1)


@pytest.mark.parametrize("offs", [(-1, 0, 1), (-40, -1, 0, 1, 40), (-900, -30, -1, 0, 1, 30, 900), (-1, 0, 1, 5), (-7, -1, 0, 1),
                                  (-3, -2, -1), (2, 3, 4, 9), (-2, 0, 1, 2), (-5, -4, -3, 0, 3, 4, 5), (0, 1, 2, 3, 4, 5, 6, 7)],
                         ids=lambda o: "_".join(map(str, o)))
def test_column_triples_of_uniform_blocks(sa, oracle, offs):
    """Full uniform blocks whose pattern holds a column triple (c - 1, c, c + 1) read x for c -+ 1 from column c's loads
    (wavefront shifts + one two-line load for the block's ends; compile-time paths for 3, 5 and 7 slots with the triple
    in the middle, the general path for every other shape).  Same values, same fold: y and the fused dot are
    bit-identical with the shortcut on and off, at every alignment of the interior run against the 128-row blocks."""
    ctx = sa.default_ctx(0)
    offs = np.array(offs)
    for n, lead in ((6000, 0), (6151, 37)):
        # rows [lo, hi) carry the full pattern (one value per offset); the rest only their diagonal-ish first entry
        lo, hi = max(0, -offs.min()) + lead, n - max(0, offs.max())
        rows = [r + offs if lo <= r < hi else np.array([r]) for r in range(n)]
        indptr = np.zeros(n + 1, dtype=np.int32)
        np.cumsum([len(c) for c in rows], out=indptr[1:])
        cols = np.concatenate(rows).astype(np.int32)
        data = np.concatenate([(c - r) * 0.375 + 1.25 for r, c in enumerate(rows)])
        x = rand_vec(n, np.float64, 29)
        ref = oracle.spmv(indptr, cols, data, x)
        dots = {}
        for tri in (1, 0):
            ctx.set("spmv_triple", tri)
            A = sa.HipCsr.new((n, n), indptr, cols, data)
            assert A.stream_format()[0] == 2
            nb, nu = A.wide_blocks()
            assert nu >= (hi - lo) // 128 - 1, (nb, nu)
            y = np.full(n, -3.0)
            A.mul_vec(x, y)
            assert np.array_equal(bits(y), bits(ref)), (tri, n)
            y2 = np.zeros(n)
            dots[tri] = A.mul_vec_dot(x, y2)
            assert np.array_equal(bits(y2), bits(ref))
        assert dots[0] == dots[1]
    ctx.set("spmv_triple", -1)


@pytest.mark.parametrize("shape", [(300, 6, 5), (517, 5, 4), (129, 7, 3), (1000, 4, 1), "dirichlet300", "dirichlet1000"],
                         ids=lambda sh: sh if isinstance(sh, str) else "x".join(map(str, sh)))
def test_seam_blocks_run_the_uniform_path(sa, oracle, shape):
    """A 128-row block that holds the x = nx - 1 | x = 0 seam of a truncated stencil is uniform but for two adjacent rows
    that lack one slot each (or, in a Dirichlet grid, hold a single entry of their own): flagged at creation (knob
    spmv_seam), it runs the uniform path, the two rows folding only what they have.  More blocks flagged, y bit-identical,
    the fused dot identical (same lanes, same rows, same order)."""
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    if isinstance(shape, str):
        # the reference bench's own matrix (benches/bicgstab.rs:54-89): identity rows on the border — the two rows at a
        # line seam hold ONE entry with a value of their own, on an offset the interior pattern has
        indptr, cols, data = gen.grid_laplacian_dirichlet(*((300, 300) if shape == "dirichlet300" else (1000, 1000)))
    else:
        indptr, cols, data, _ = gen.poisson3d(*shape)
    n = indptr.size - 1
    x = rand_vec(n, np.float64, 41)
    ref = oracle.spmv(indptr, cols, data, x)
    got = {}
    try:
        for seam in (0, 1):
            ctx.set("spmv_seam", seam)
            A = sa.HipCsr.new((n, n), indptr, cols, data)
            assert A.stream_format()[0] == 2
            nb, nu = A.wide_blocks()
            y = np.full(n, 9.0)
            A.mul_vec(x, y)
            assert np.array_equal(bits(y), bits(ref)), seam
            y2 = np.zeros(n)
            d = A.mul_vec_dot(x, y2)
            assert np.array_equal(bits(y2), bits(ref))
            got[seam] = (nu, d)
        assert got[1][0] > got[0][0], got              # seam blocks were found
        assert got[1][1] == got[0][1]
    finally:
        ctx.set("spmv_seam", -1)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("kind", ["band_random", "poisson3d_random", "ragged_mix", "len1_and_len8", "head_shift"])
def test_offset_code_uniform_blocks(sa, oracle, dtype, kind):
    """Uniform 64-row blocks of the OFFSET-code stream (variable coefficients: values stay 8 B/nnz, but rows that all
    repeat one offset pattern need neither row_ptr nor their code bytes).  y must stay bit-identical to the reference
    fold with the uniform path on and off, for every scalar type, for blocks at every 4-byte phase of the code stream,
    for uniform blocks next to ragged ones and at the matrix end, for row lengths 1, 8, 9 (a second 8-slot chunk of the
    pattern), 32 (the longest a uniform block may have) and 33 (never uniform)."""
    from sprsolve_amd import gen
    import scipy.sparse as sp
    ctx = sa.default_ctx(0)
    rng = np.random.default_rng(5)
    if kind == "band_random":
        indptr, cols, data, _ = gen.symmetric_banded(64 * 40 + 17, 3)               # 7 per interior row, random values
        min_uniform = 36
    elif kind == "poisson3d_random":
        indptr, cols, data, _ = gen.poisson3d(150, 9, 7, values="random")
        min_uniform = 7 * 9                                                          # >= one interior block per 150-row line
    elif kind == "ragged_mix":
        # rows 0..639: a 5-diagonal band (uniform blocks); rows 640..: random lengths 0..6 (ragged, incl. empty rows); then a band again
        n = 2000
        rows, cc = [], []
        for r in range(n):
            if r < 640 or r >= 1408:
                c = [r + o for o in (-2, -1, 0, 1, 2) if 0 <= r + o < n]
            else:
                k = int(rng.integers(0, 7))
                c = sorted(set(int(v) for v in np.clip(r + rng.integers(-9, 10, k), 0, n - 1)))
            rows += [r] * len(c); cc += c
        M = sp.csr_matrix((rng.uniform(-1, 1, len(cc)), (rows, cc)), shape=(n, n))
        M.sort_indices()
        indptr, cols, data = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data
        min_uniform = 8 + 7
    elif kind == "len1_and_len8":
        n = 64 * 12
        rows, cc = [], []
        for r in range(n):
            L = 1 if r < 64 * 3 else (8 if r < 64 * 6 else (9 if r < 64 * 8 else (32 if r < 64 * 10 else 33)))
            c = [(r + 3 * j) % n for j in range(L)]
            rows += [r] * L; cc += sorted(c)
        M = sp.csr_matrix((rng.uniform(-1, 1, len(cc)), (rows, cc)), shape=(n, n))
        M.sort_indices()
        indptr, cols, data = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data
        min_uniform = 2 + 2 + 1
    else:   # head_shift: 1, 2, 3 short rows in front so that the uniform run starts at every code-stream phase
        out = []
        for shift in (1, 2, 3):
            n = 64 * 9 + 5
            rows, cc = [0] * shift, list(range(shift))
            for r in range(1, n):
                c = [r + o for o in (-1, 0, 1) if 0 <= r + o < n]
                rows += [r] * len(c); cc += c
            M = sp.csr_matrix((rng.uniform(-1, 1, len(cc)), (rows, cc)), shape=(n, n))
            M.sort_indices()
            out.append((M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data))
        for indptr, cols, data in out:
            _check_offset_uniform(sa, oracle, ctx, dtype, indptr, cols, data, 6)
        return
    _check_offset_uniform(sa, oracle, ctx, dtype, indptr, cols, data, min_uniform)


def _check_offset_uniform(sa, oracle, ctx, dtype, indptr, cols, data, min_uniform):
    n = indptr.size - 1
    d = data.astype(dtype)
    if np.dtype(dtype).kind == "c":
        d = d * (1 - 0.5j)
    x = rand_vec(n, dtype, 23)
    ref = oracle.spmv(indptr, cols, d, x)
    ctx.set("spmv_dict", 1)
    try:
        for uni in (1, 0):
            ctx.set("spmv_uniform", uni)
            A = sa.HipCsr.new((n, n), indptr, cols, d)
            assert A.stream_format()[0] == 1
            nb, nu = A.wide_blocks()
            if uni:
                assert nb >= (n + 63) // 64 and min_uniform <= nu <= nb, (nb, nu, min_uniform)
            else:
                assert nu == 0
            y = np.full(n, 3.0, dtype=dtype)
            A.mul_vec(x, y)
            assert np.array_equal(bits(y), bits(ref)), ("mul_vec", uni)
            y2 = np.zeros(n, dtype=dtype)
            dot = A.mul_vec_dot(x, y2)
            assert np.array_equal(bits(y2), bits(ref)), ("mul_vec_dot", uni)
            e = oracle.conj_dot(x, ref)
            tol = 2e-4 if np.dtype(dtype).itemsize in (4, 8) and np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-12
            assert abs(dot - e) <= tol * max(1.0, abs(e))
    finally:
        ctx.set("spmv_uniform", -1)
        ctx.set("spmv_dict", -1)


def test_xcd_period_schedule_is_a_pure_reordering(sa, oracle):
    """Knob spmv_period = 1: the 128-row blocks of the f64 pair-code stream are walked in the XCD-period order (rows r
    and r +- plane on one XCD).  Only the ORDER in which blocks are multiplied changes: y bit-identical, the solve
    reaches the same solution; the fused dot groups rows differently (tolerance)."""
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    nx, ny, nz = 160, 128, 12                      # plane = 20480 rows: chunks of 2560 rows = 20 wide blocks
    indptr, cols, data, rhs = gen.poisson3d(nx, ny, nz)
    n = rhs.size
    x = rand_vec(n, np.float64, 3)
    ref = oracle.spmv(indptr, cols, data, x)
    out = {}
    try:
        for period, triple in ((0, 1), (1, 1), (0, 0), (1, 0)):
            ctx.set("spmv_period", period)
            ctx.set("spmv_triple", triple)   # columns c -+ 1 of a uniform block read from column c's loads: same values
            A = sa.HipCsr.new((n, n), indptr, cols, data)
            assert A.stream_format()[0] == 2
            y = np.zeros(n); d = A.mul_vec_dot(x, y)
            assert np.array_equal(bits(y), bits(ref)), period
            e = oracle.conj_dot(x, ref)
            assert abs(d - e) <= 1e-12 * max(1.0, abs(e))
            s = sa.BiCGStab.new(A, n)
            sol = np.zeros(n)
            its, res = s.solve(rhs, sol, 2000, 1e-10)
            assert np.max(np.abs(sol - 1.0)) < 1e-7
            out[period, triple] = (its, d)
        assert abs(out[0, 1][0] - out[1, 1][0]) <= max(3, out[0, 1][0] // 10)
        assert out[0, 1] == out[0, 0] and out[1, 1] == out[1, 0]        # the triple shortcut changes timing only
    finally:
        ctx.set("spmv_period", -1)
        ctx.set("spmv_triple", -1)



def _tile_cases():
    from sprsolve_amd import gen
    def p3(nx, ny, nz):
        ip, ix, d, rhs = gen.poisson3d(nx, ny, nz)
        return ip, ix, d, rhs, 1.0
    def dirichlet(r):
        ip, ix, d = gen.grid_laplacian_dirichlet(r, r)
        return ip, ix, d, (gen.dirichlet_rhs(r, r) if r <= 300 else None), None      # (the 1000 x 1000 solve needs > 3000 iterations)
    def tridiag(n):
        ip = np.concatenate([[0], np.cumsum(np.r_[2, np.full(n - 2, 3), 2])]).astype(np.int32)
        ix = np.concatenate([[0, 1]] + [np.arange(i - 1, i + 2) for i in range(1, n - 1)] + [[n - 2, n - 1]]).astype(np.int32)
        d = np.concatenate([[2.5, -1.0]] + [np.array([-1.0, 2.5, -1.0])] * (n - 2) + [[-1.0, 2.5]])
        return ip, ix, d, None, None
    return {
        "p3_160x128x12_eighths": lambda: p3(160, 128, 12),        # shape (7, 1, 1); plane = 5 tiles: dealt in eighths
        "p3_160x256x8_period": lambda: p3(160, 256, 8),           # far band 40960 = 10 tiles: the period sections
        "p3_500x100x8_seams": lambda: p3(500, 100, 8),            # cfg 5's line length: 8 seams per tile
        "p2_dirichlet_1000_far_lines": lambda: dirichlet(1000),   # (5, 1, 1): +-1000 are far; identity rows with their own value
        "p2_dirichlet_300_near_lines": lambda: dirichlet(300),    # (5, 0, 0): everything from the window
        "tridiagonal_200k": lambda: tridiag(200_000),             # (3, 0, 0)
    }


@pytest.mark.parametrize("name", list(_tile_cases()))
def test_lds_window_tiles_bit_identical(sa, oracle, name):
    """Knob spmv_tile (csrc/spmv_dict.hip, spmv_tile_kernel): runs of 32 full uniform 128-row blocks of one stencil pattern are
    multiplied from an x window staged in LDS (near columns) plus per-row-pair far loads; the remaining blocks by the
    per-block walk inside the SAME launch.  Same products, same left-to-right fold per row: y bit-identical to the oracle,
    to the per-block kernel and in all three launch flavours (plain, fused dot, fused double dot inside a solve)."""
    ctx = sa.default_ctx(0)
    indptr, cols, data, rhs, exact = _tile_cases()[name]()
    n = indptr.size - 1
    x = rand_vec(n, np.float64, 77)
    ref = oracle.spmv(indptr, cols, data, x)
    e = oracle.conj_dot(x, ref)
    plans = {}
    try:
        for tile in (0, 1):
            ctx.set("spmv_tile", tile)
            A = sa.HipCsr.new((n, n), indptr, cols, data)
            assert A.stream_format()[0] == 2
            plans[tile] = A.tile_plan()
            y = np.full(n, 9.0)
            A.mul_vec(x, y)
            assert np.array_equal(bits(y), bits(ref)), (tile, int(np.argmax(y != ref)))
            y2 = np.full(n, -3.0)
            d = A.mul_vec_dot(x, y2)
            assert np.array_equal(bits(y2), bits(ref))
            assert abs(d - e) <= 1e-12 * max(1.0, float(np.sum(np.abs(x * ref))))
            if rhs is not None:
                s = sa.BiCGStab.new(A, n)
                sol = np.zeros(n)
                its, res = s.solve(rhs, sol, 3000, 1e-9)
                r = rhs - oracle.spmv(indptr, cols, data, sol)
                assert np.linalg.norm(r) <= 1e-8 * np.linalg.norm(rhs) + 1e-9, (tile, its, res)
                if exact is not None:
                    assert np.max(np.abs(sol - exact)) < 1e-6
                if tile == 1 and name.startswith("p3_"):
                    # MINRES on the same (symmetric) operator: its v.Av rides on the SpMV with the INPUT vector as the dot operand,
                    # which the tile kernels take from the staged window instead of loading it
                    m = sa.MinRes.new(A, n); sol2 = np.zeros(n)
                    its2, res2 = m.solve(rhs, sol2, 3000, 1e-9)
                    assert np.max(np.abs(sol2 - exact)) < 1e-5, (its2, res2)
        assert plans[0] == (0, 0, 0)
        nt, ntb, nob = plans[1]
        nb, nu = A.wide_blocks()
        assert nt >= 8 and ntb == 32 * nt and ntb + nob == nb, (plans, nb)
        assert ntb >= 0.3 * nb, (plans, nb)                               # a good part of the matrix runs through tiles
    finally:
        ctx.set("spmv_tile", -1)


def _chain_cases():
    from sprsolve_amd import gen
    def p3(nx, ny, nz):
        ip, ix, d, rhs = gen.poisson3d(nx, ny, nz)
        return ip, ix, d, rhs, 1.0
    def dirichlet(r):
        ip, ix, d = gen.grid_laplacian_dirichlet(r, r)
        return ip, ix, d, None, None
    return {
        "p3_160x128x24": lambda: p3(160, 128, 24),            # 9 chains of 22 tiles; plane = 160 x 128: the 128-row grid does not drift
        "p3_500x100x14_seams_drift": lambda: p3(500, 100, 14),  # cfg 5's line length (seams in every block); plane 50000 = 390.6 blocks: tiles drift by 80 rows a plane
        "p3_250x84x20_drift": lambda: p3(250, 84, 20),        # plane 21000 = 164.06 blocks: 8 rows a plane, chains survive all 18 planes
        "p2_dirichlet_1200_far_lines": lambda: dirichlet(1200),   # UL = 5, far = +-1200 < a tile: no tile links to another — chains of one tile (both windows virtual)
    }


@pytest.mark.parametrize("name", list(_chain_cases()))
def test_plane_streaming_chains_bit_identical(sa, oracle, name):
    """Knob spmv_chain (csrc/spmv_chain.hip): for patterns with one far slot a side at -P / +P a workgroup walks a column of
    2048-row tiles plane by plane with the x windows of three consecutive tiles in LDS — the +-P operands come from the
    neighbouring tiles' windows, no far load exists.  Same products, same left-to-right fold per row (seam rows included): y
    bit-identical to the oracle, to the tile kernel and to the per-block kernel, in every launch flavour a solve uses (plain; dot
    with another vector; dot with the input vector; double dot with the input / with another vector)."""
    ctx = sa.default_ctx(0)
    indptr, cols, data, rhs, exact = _chain_cases()[name]()
    n = indptr.size - 1
    x = rand_vec(n, np.float64, 79)
    ref = oracle.spmv(indptr, cols, data, x)
    e = oracle.conj_dot(x, ref)
    got = {}
    try:
        for chain, tile in ((1, 1), (0, 1), (0, 0)):
            ctx.set("spmv_chain", chain); ctx.set("spmv_tile", tile)
            A = sa.HipCsr.new((n, n), indptr, cols, data)
            assert A.stream_format()[0] == 2
            cp, tp = A.chain_plan(), A.tile_plan()
            if chain:
                assert cp[0] >= 64 and tp == (0, 0, 0), (cp, tp)
                nb, _ = A.wide_blocks()
                assert 16 * cp[0] + cp[3] == nb and 16 * cp[0] >= 0.3 * nb, (cp, nb)
                if name.startswith("p3_"):
                    assert cp[0] >= 6 * cp[2], cp                           # chains run through the planes (>= 6 tiles each on average)
                else:
                    assert cp[0] == cp[2] == cp[1], cp                      # chains of one tile
            else:
                assert cp == (0, 0, 0, 0) and (tp[0] >= 8) == bool(tile), (cp, tp)
            y = np.full(n, 9.0)
            A.mul_vec(x, y)
            bad = np.flatnonzero(y != ref)
            assert bad.size == 0, (chain, tile, bad[:8], bad.size)
            assert np.array_equal(bits(y), bits(ref))
            y2 = np.full(n, -3.0)
            d = A.mul_vec_dot(x, y2)                                         # DOT 1, operand = the input vector (from the window)
            assert np.array_equal(bits(y2), bits(ref))
            assert abs(d - e) <= 1e-12 * max(1.0, float(np.sum(np.abs(x * ref))))
            if rhs is not None:
                s = sa.BiCGStab.new(A, n); s.set_trace(8)
                sol = np.zeros(n)
                its, res = s.solve(rhs, sol, 3000, 1e-9)                     # K2: DOT 1 with r0; K4: DOT 2 with its input
                assert np.max(np.abs(sol - exact)) < 1e-6
                sj = sa.BiCGStab.new(A, n); solj = np.zeros(n)
                itsj, resj = sj.precond_solve(sa.DiagPrecond.new(np.full(n, 6.0)), rhs, solj, 3000, 1e-9)   # K4: DOT 2 with another vector
                assert np.max(np.abs(solj - exact)) < 1e-6
                m = sa.MinRes.new(A, n); sol2 = np.zeros(n)
                its2, res2 = m.solve(rhs, sol2, 3000, 1e-9)
                assert np.max(np.abs(sol2 - exact)) < 1e-5, (its2, res2)
                got[(chain, tile)] = (its, res, bits(sol).copy(), s.trace().copy(), itsj, bits(solj).copy(), its2, bits(sol2).copy())
        if rhs is not None:
            # chains and tiles group the rows into the same workgroups' partials differently (the grids differ), so the solves agree
            # to rounding; the iteration counts at 1e-9 must be close and the first trace rows equal to 1e-9
            a, b = got[(1, 1)], got[(0, 1)]
            # (BiCGStab's count at 1e-9 is reduction-order noise; MINRES' is not)
            assert abs(a[0] - b[0]) <= max(3, b[0] // 4) and abs(a[4] - b[4]) <= max(3, b[4] // 4) and abs(a[6] - b[6]) <= max(2, b[6] // 20)
            assert np.allclose(a[3][:4], b[3][:4], rtol=1e-9, atol=1e-12)
    finally:
        ctx.set("spmv_chain", -1); ctx.set("spmv_tile", -1)


def _fused_against_five_launches(sa, ctx, A, n, rhs, rhs2, x0, exact, exact_tol=1e-6):
    """BiCGStab with spmv_fuse 1 against spmv_fuse 0 on handle A: outcome, iteration count, residual, x and every traced scalar bit for
    bit over a set of solves (to convergence, tol = 0, x0 != 0, one / two iterations, a loose tolerance, a long tol = 0 run through the
    restart branch, traced and polled)."""
    out = {}
    for fuse in (1, 0):
        ctx.set("spmv_fuse", fuse)
        res = []
        # (the 700-iteration run at tol = 0 goes on long after rounding-level convergence: rho = r0.r decays to a cancellation
        # residue and the restart branch, bicg_stab.rs:131-145, fires — executed by the host, resumed by a K2 in "resumed" mode)
        for b, start, max_iter, tol in ((rhs, None, 3000, 1e-10), (rhs2, x0, 3000, 1e-9), (rhs2, None, 37, 0.0), (rhs, None, 1, 0.0),
                                        (rhs2, None, 2, 0.0), (rhs, None, 3000, 0.5), (rhs, None, 700, 0.0)):
            s = sa.BiCGStab.new(A, n); s.set_trace(64); s.set_profile(True)
            x = np.zeros(n) if start is None else start.copy()
            try:
                its, rr = s.solve(b, x, max_iter, tol)
                st = "ok"
            except sa.error.InsufficientIterNum as e:
                its, rr, st = e.iters, None, "insufficient"
            prof = s.profile()
            res.append((st, its, rr, bits(x).copy(), bits(s.trace()).copy(), prof["fused_k2"], prof["fused_k4"], prof["spmv_launches"]))
        # the restart run again WITHOUT a trace: the host then polls every `poll` iterations, so a restart request is followed by up
        # to poll - 1 iterations of idle launches for each of which the host has already rotated its buffer names (odd and even counts)
        for poll in (16, 5, 2):
            ctx.set("poll", poll)
            s = sa.BiCGStab.new(A, n); s.set_profile(True)
            x = np.zeros(n)
            try:
                its, rr = s.solve(rhs, x, 700, 0.0); st = "ok"
            except sa.error.InsufficientIterNum as e:
                its, rr, st = e.iters, None, "insufficient"
            prof = s.profile()
            res.append((st, its, rr, bits(x).copy(), np.zeros(0, np.uint8), prof["fused_k2"], prof["fused_k4"], prof["spmv_launches"]))
        ctx.set("poll", 16)
        out[fuse] = res
    for a, b in zip(out[1], out[0]):
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], (a[:3], b[:3])
        assert np.array_equal(a[3], b[3]), "x differs between the fused and the five-launch iteration"
        assert np.array_equal(a[4], b[4]), "a traced scalar differs"
        assert b[5] == 0 and b[6] == 0
        assert a[6] >= 1 and (a[5] >= 1 or a[1] <= 1), a[5:]          # K4 is fused from the first iteration on, K2 from the second
        assert a[7] == b[7] or a[4].size == 0                          # the same number of SpMV launches either way (traced runs)
    err = np.max(np.abs(out[1][0][3].view(np.float64) - exact))
    assert out[1][0][0] == "ok" and err < exact_tol * max(1.0, float(np.max(np.abs(exact)))), (out[1][0][:3], err)


@pytest.mark.parametrize("name", [k for k in _chain_cases() if k.startswith("p3_")])
def test_fused_spmv_input_is_bit_identical(sa, oracle, name):
    """Knob spmv_fuse (csrc/krylov.hip "fused SpMV input", csrc/spmv_chain.hip FUSE): on a handle whose SpMV runs through chains
    BiCGStab forms K3's r -= alpha v inside K4 and K1's p = (v (-beta w) + p beta) + r inside K2 (bicg_stab.rs:155-156,172) — the
    same prologues, the same rounding sequence per element, the same dot partials.  So the three-launch iteration must reproduce
    the five-launch one BIT FOR BIT: iteration count, residual, every traced scalar, x — to convergence and for a fixed number
    of iterations (tol = 0), with a non-zero initial guess, and through the breakdown / early-convergence exits."""
    ctx = sa.default_ctx(0)
    indptr, cols, data, rhs, exact = _chain_cases()[name]()
    n = indptr.size - 1
    rng = np.random.default_rng(5)
    rhs2 = rng.uniform(-1, 1, n)
    x0 = rng.uniform(-1, 1, n)
    out = {}
    try:
        ctx.set("spmv_chain", 1); ctx.set("spmv_tile", 1)
        A = sa.HipCsr.new((n, n), indptr, cols, data)
        assert A.chain_plan()[0] >= 64
        _fused_against_five_launches(sa, ctx, A, n, rhs, rhs2, x0, exact)
    finally:
        ctx.set("spmv_chain", -1); ctx.set("spmv_tile", -1); ctx.set("spmv_fuse", -1); ctx.set("poll", 16)


def _m3_cases():
    from sprsolve_amd import gen

    def banded_f64():
        ip, ix, d, rhs = gen.symmetric_banded(40000, 4)                     # cfg 3's matrix in small: offset codes, 16-byte value loads
        return "minres", ip, ix, d, rhs, -1

    def cs_grid_pair():
        ip, ix, d, rhs, _ = gen.complex_symmetric_grid(150, 200)            # cfg 4's: complex pair codes
        return "csminres", ip, ix, d, rhs, -1

    def cs_grid_offsets():
        ip, ix, d, rhs, _ = gen.complex_symmetric_grid(120, 90)             # the same operator through offset codes (knob)
        return "csminres", ip, ix, d, rhs, 1

    def hermitian_pair():
        ip, ix, d, rhs, _ = gen.complex_hermitian_grid(100, 110)            # MINRES on a complex Hermitian operator (tests/test_complex_solve.rs:95-151)
        return "minres", ip, ix, d, rhs, -1

    return {"banded_f64_offsets": banded_f64, "complex_symmetric_pairs": cs_grid_pair, "complex_symmetric_offsets": cs_grid_offsets,
            "complex_hermitian_pairs": hermitian_pair}


@pytest.mark.parametrize("name", list(_m3_cases()))
def test_minres_m3_inside_m1_is_bit_identical(sa, oracle, name):
    """Knob spmv_fuse for MINRES / CSMINRES (csrc/krylov.hip "M3 deferred", csrc/minres_fuse.hpp MinresM23, csrc/spmv_dict.hip
    spmv_dict_scaled_kernel): M3 of iteration k — beta_new, the normalisation, the Givens rotation, p, x, the convergence test
    (minres.rs:120-168) — is not launched; the SpMV of iteration k + 1 multiplies by the un-normalised v_new scaled in its gathers
    and M3's element-wise work rides with M2 of iteration k + 1.  Two launches per iteration instead of three, and everything the
    solve returns must be bit for bit what the three-launch iteration returns: iteration count, residual, x — to convergence, for
    fixed iteration counts (tol = 0), with x0 != 0, for every poll interval (the M3 before a poll is launched on its own and writes
    the raw v back normalised) and with a trace (no deferral at all)."""
    ctx = sa.default_ctx(0)
    kind, indptr, cols, data, rhs, knob = _m3_cases()[name]()
    n = indptr.size - 1
    rng = np.random.default_rng(9)
    cplx = np.dtype(data.dtype).kind == "c"
    rhs2 = (rng.uniform(-1, 1, n) + (1j * rng.uniform(-1, 1, n) if cplx else 0)).astype(data.dtype)
    x0 = (rng.uniform(-1, 1, n) + (1j * rng.uniform(-1, 1, n) if cplx else 0)).astype(data.dtype)
    cls = sa.MinRes if kind == "minres" else sa.CSMinRes
    out = {}
    try:
        ctx.set("spmv_dict", knob)
        A = sa.HipCsr.new((n, n), indptr, cols, data)
        assert A.stream_format()[0] in (1, 2)
        for fuse in (1, 0):
            ctx.set("spmv_fuse", fuse)
            res = []
            for poll in (16, 1, 2, 5, 64):
                ctx.set("poll", poll)
                for b, start, max_iter, tol in ((rhs, None, 4000, 1e-10), (rhs2, x0, 4000, 1e-9), (rhs2, None, 37, 0.0), (rhs, None, 1, 0.0),
                                                (rhs2, None, 2, 0.0), (rhs, None, 3, 0.0), (rhs, None, 4000, 0.3), (rhs, None, 64, 0.0), (rhs, None, 65, 0.0)):
                    s = cls.new(A, n); s.set_profile(True)
                    x = np.zeros(n, dtype=data.dtype) if start is None else start.copy()
                    try:
                        its, rr = s.solve(b, x, max_iter, tol); st = "ok"
                    except sa.error.InsufficientIterNum as e:
                        its, rr, st = e.iters, None, "insufficient"
                    prof = s.profile()
                    res.append((st, its, rr, bits(x).copy(), prof["fused_k2"], prof["steps"]))
            ctx.set("poll", 16)
            # traced: one poll per iteration, M3 always on its own — the trace itself must not change
            s = cls.new(A, n); s.set_trace(40)
            x = np.zeros(n, dtype=data.dtype)
            try:
                s.solve(rhs, x, 30, 0.0)
            except sa.error.InsufficientIterNum:
                pass
            res.append(("trace", 0, None, bits(s.trace()).copy(), 0, 0))
            out[fuse] = res
        fused_any = 0
        for a, b in zip(out[1], out[0]):
            assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], (a[:3], b[:3])
            assert np.array_equal(a[3], b[3]), "x (or the trace) differs between the two- and the three-launch iteration"
            assert b[4] == 0 and a[5] == b[5]              # the same number of SpMV launches either way
            fused_any += a[4]
        assert fused_any > 0
        ref = (oracle.minres if kind == "minres" else oracle.csminres)(indptr, cols, data, rhs, np.zeros(n, dtype=data.dtype), 4000, 1e-10)
        assert (ref.status == 0) == (out[1][0][0] == "ok")
        if ref.status == 0:
            assert abs(ref.its - out[1][0][1]) <= max(2, ref.its // 20)          # (reduction order: the count of a 2800-iteration solve moves by a few per cent)
            assert np.max(np.abs(out[1][0][3].view(data.dtype) - ref.x)) < 1e-6 * max(1.0, float(np.max(np.abs(ref.x))))
    finally:
        ctx.set("spmv_dict", -1); ctx.set("spmv_fuse", -1); ctx.set("poll", 16)


def test_fused_spmv_input_randomised(sa, oracle):
    """The three-launch iteration against the five-launch one on random chain-capable grids (line lengths with seams in different
    positions, plane sizes that drift against the 128-row block grid by different amounts), random right-hand sides, initial
    guesses, iteration limits, tolerances and poll intervals: outcome, iteration count, residual and x bit for bit; the solution
    against the oracle's where the solve converges."""
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    rng = np.random.default_rng(2024)
    try:
        ctx.set("spmv_chain", 1); ctx.set("spmv_tile", 1)
        for case in range(10):
            nx = int(rng.integers(75, 126)) * 2; ny = int(rng.integers(60, 91)); nz = int(rng.integers(18, 27))
            ip, ix, d, rhs1 = gen.poisson3d(nx, ny, nz)
            n = rhs1.size
            A = sa.HipCsr.new((n, n), ip, ix, d)
            assert A.chain_plan()[0] >= 64, (nx, ny, nz, A.chain_plan())
            rhs = rhs1 if case % 3 == 0 else rng.uniform(-1, 1, n)
            x0 = rng.uniform(-1, 1, n) if case % 4 == 1 else np.zeros(n)
            max_iter = int(rng.choice([1, 2, 3, 17, 64, 400, 900]))
            tol = float(rng.choice([0.0, 1e-3, 1e-10]))
            poll = int(rng.integers(1, 17))
            ctx.set("poll", poll)
            got = []
            for fuse in (1, 0):
                ctx.set("spmv_fuse", fuse)
                s = sa.BiCGStab.new(A, n); s.set_profile(True)
                x = x0.copy()
                try:
                    its, rr = s.solve(rhs, x, max_iter, tol); st = "ok"
                except sa.error.InsufficientIterNum as e:
                    its, rr, st = e.iters, None, "insufficient"
                except sa.error.BreakDown as e:
                    its, rr, st = e.its, None, "breakdown"
                got.append((st, its, rr, bits(x).copy(), s.profile()["fused_k4"]))
            a, b = got
            assert a[:3] == b[:3], (case, nx, ny, nz, max_iter, tol, poll, a[:3], b[:3])
            assert np.array_equal(a[3], b[3]), (case, nx, ny, nz, max_iter, tol, poll)
            assert a[4] >= 1 and b[4] == 0
            if a[0] == "ok" and tol == 1e-10:
                ref = oracle.bicgstab(ip, ix, d, rhs, x0, max_iter, tol)
                assert ref.status == oracle.OK
                assert np.max(np.abs(a[3].view(np.float64) - ref.x)) <= 1e-7 * max(1.0, float(np.max(np.abs(ref.x))))
    finally:
        ctx.set("spmv_chain", -1); ctx.set("spmv_tile", -1); ctx.set("spmv_fuse", -1); ctx.set("poll", 16)


def test_plane_streaming_chains_policy(sa, oracle):
    """Automatic policy: chains only where they fill the chip (about one segment of >= 6 tiles per workgroup) — a 6 M-row
    500 x 200 x 60 grid (46 MiB vectors: tiles wanted, 48 chains cut into 7 segments each) qualifies; small grids keep the tile
    plan under spmv_tile = 1; with spmv_chain = 0 at LAUNCH time a handle that has chains multiplies through its tile plan
    instead (same y)."""
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    try:
        ctx.set("spmv_tile", 1)
        ip, ix, d, rhs = gen.poisson3d(160, 128, 24)
        n = rhs.size
        A = sa.HipCsr.new((n, n), ip, ix, d)
        assert A.chain_plan() == (0, 0, 0, 0) and A.tile_plan()[0] >= 8       # 18 segments would leave most of the chip idle
        ctx.set("spmv_tile", -1)
        ip, ix, d, rhs = gen.poisson3d(500, 200, 60)
        n = rhs.size
        A = sa.HipCsr.new((n, n), ip, ix, d)
        cp = A.chain_plan()
        assert cp[0] >= 2500 and cp[1] >= 256 and cp[0] >= 6 * cp[1] and A.tile_plan() == (0, 0, 0), cp
        x = rand_vec(n, np.float64, 3)
        ref = oracle.spmv(ip, ix, d, x)
        y = np.zeros(n); A.mul_vec(x, y)
        assert np.array_equal(bits(y), bits(ref))
        ctx.set("spmv_chain", 0)                                               # launch-time: the same handle through its tiles
        assert A.chain_plan() == (0, 0, 0, 0) and A.tile_plan()[0] >= 8
        y = np.zeros(n); A.mul_vec(x, y)
        assert np.array_equal(bits(y), bits(ref))
    finally:
        ctx.set("spmv_chain", -1); ctx.set("spmv_tile", -1)


def test_lds_window_tiles_policy_and_fallbacks(sa, oracle):
    """Automatic policy: cache-resident matrices keep the per-block kernel (no plan); patterns the kernel is not built for
    (near slots beyond the window, unsorted far / near order, more than 8 slots) get no plan under spmv_tile = 1 either and
    stay bit-exact; a preconditioned solve (the double dot's operand is NOT the input vector) agrees with the per-block one."""
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    ip, ix, d, rhs = gen.poisson3d(160, 128, 12)
    n = rhs.size
    A = sa.HipCsr.new((n, n), ip, ix, d)
    assert A.tile_plan() == (0, 0, 0)                                      # automatic: 2 MB vectors
    try:
        ctx.set("spmv_tile", 1)
        # two far bands on each side (nx = 1600 > the wide window too): shape (7, 2, 2) is not built
        ip2, ix2, d2, _ = gen.poisson3d(1600, 20, 20)
        n2 = ip2.size - 1
        A2 = sa.HipCsr.new((n2, n2), ip2, ix2, d2)
        assert A2.stream_format()[0] == 2 and A2.tile_plan() == (0, 0, 0)
        x2 = rand_vec(n2, np.float64, 5); y2 = np.zeros(n2); A2.mul_vec(x2, y2)
        assert np.array_equal(bits(y2), bits(oracle.spmv(ip2, ix2, d2, x2)))
        # Jacobi-preconditioned BiCGStab: K4's operand differs from its input
        A = sa.HipCsr.new((n, n), ip, ix, d)
        assert A.tile_plan()[0] >= 8
        diag = np.full(n, 6.0)
        out = []
        for tile in (1, 0):
            ctx.set("spmv_tile", tile)          # the launch honours the knob at run time too
            s = sa.BiCGStab.new(A, n); sol = np.zeros(n)
            its, res = s.precond_solve(sa.DiagPrecond.new(diag), rhs, sol, 2000, 1e-10)
            assert np.max(np.abs(sol - 1.0)) < 1e-7
            out.append(its)
        assert abs(out[0] - out[1]) <= max(3, out[0] // 10)
    finally:
        ctx.set("spmv_tile", -1)


@pytest.mark.parametrize("name", list(_tile_cases()))
def test_lds_window_tiles_with_a_value_per_entry(sa, oracle, name):
    """The same tiles on the OFFSET-CODE stream (spmv_tile_off_kernel): the patterns of `test_lds_window_tiles_bit_identical`
    with a different value in every entry (no value dictionary, so the handle multiplies with one-byte offset codes + the
    values).  A block's values are streamed through the wavefront's LDS slice; rows behind a short seam row start earlier.
    y bit-identical to the oracle and to the per-block kernel; odd and even first entries of blocks both occur (seam rows
    shift the parity), the last block of the matrix stays outside the tiles."""
    ctx = sa.default_ctx(0)
    indptr, cols, data, rhs, exact = _tile_cases()[name]()
    n = indptr.size - 1
    rng = np.random.default_rng(11)
    vals = rng.uniform(0.5, 1.5, data.size) * np.where(data == 0, 1.0, np.sign(data))
    x = rand_vec(n, np.float64, 78)
    ref = oracle.spmv(indptr, cols, vals, x)
    e = oracle.conj_dot(x, ref)
    plans = {}
    try:
        for tile in (0, 1):
            ctx.set("spmv_tile", tile)
            A = sa.HipCsr.new((n, n), indptr, cols, vals)
            assert A.stream_format()[0] == 1
            plans[tile] = A.tile_plan()
            y = np.full(n, 9.0)
            A.mul_vec(x, y)
            bad = np.flatnonzero(y != ref)
            assert bad.size == 0, (tile, bad[:8], bad.size)
            assert np.array_equal(bits(y), bits(ref))
            y2 = np.full(n, -3.0)
            d = A.mul_vec_dot(x, y2)
            assert np.array_equal(bits(y2), bits(ref))
            assert abs(d - e) <= 1e-12 * max(1.0, float(np.sum(np.abs(x * ref))))
        assert plans[0] == (0, 0, 0)
        nt, ntb, nob = plans[1]
        nb = (n + 127) // 128
        assert nt >= 8 and ntb == 32 * nt and ntb + nob == nb, (plans, nb)
        assert ntb >= 0.3 * nb, (plans, nb)
        # inside a solve (the fused double dot; Jacobi so that its operand is not the input vector, then plain)
        diag = vals[np.flatnonzero(cols == np.repeat(np.arange(n), np.diff(indptr)))]
        if diag.size == n:
            b = oracle.spmv(indptr, cols, vals, np.ones(n))
            dom = np.abs(diag) * 4.0 + 8.0                                   # make it solvable: a dominant diagonal
            vals2 = vals.copy(); vals2[np.flatnonzero(cols == np.repeat(np.arange(n), np.diff(indptr)))] = dom
            b = oracle.spmv(indptr, cols, vals2, np.ones(n))
            A2 = sa.HipCsr.new((n, n), indptr, cols, vals2)
            assert A2.tile_plan()[0] >= 8
            for jac in (True, False):
                s = sa.BiCGStab.new(A2, n); sol = np.zeros(n)
                if jac:
                    its, res = s.precond_solve(sa.DiagPrecond.new(dom), b, sol, 500, 1e-11)
                else:
                    its, res = s.solve(b, sol, 500, 1e-11)
                assert np.max(np.abs(sol - 1.0)) < 1e-8, (jac, its, res)
    finally:
        ctx.set("spmv_tile", -1)


def _banded_with_defects(rng, n, offs, vals_per_diag, n_defects):
    """CSR of the band matrix with diagonals `offs` (sorted), truncated at the matrix ends, with `n_defects` random rows
    damaged: one entry removed, or the row replaced by a single diagonal entry, or a pair of adjacent rows both damaged."""
    offs = np.asarray(offs, dtype=np.int64)
    rows = np.arange(n, dtype=np.int64)
    cols = rows[:, None] + offs[None, :]
    ok = (cols >= 0) & (cols < n)
    vals = np.broadcast_to(np.asarray(vals_per_diag, dtype=np.float64)[None, :], cols.shape).copy()
    pick = rng.choice(n, size=n_defects, replace=False)
    for r in pick:
        kind = rng.integers(0, 3)
        for rr in ((r, r + 1) if kind == 2 and r + 1 < n else (r,)):
            have = np.flatnonzero(ok[rr])
            if have.size < 2:
                continue
            if kind == 1:
                d = int(np.flatnonzero(offs == 0)[0]) if 0 in offs else int(have[0])
                ok[rr, :] = False; ok[rr, d] = True; vals[rr, d] = 1.0
            else:
                ok[rr, rng.choice(have)] = False
    cnt = ok.sum(axis=1)
    indptr = np.zeros(n + 1, dtype=np.int64); np.cumsum(cnt, out=indptr[1:])
    return indptr.astype(np.int32), cols[ok].astype(np.int32), vals[ok]


@pytest.mark.parametrize("seed", range(6))
def test_lds_window_tiles_randomised_bands(sa, oracle, seed):
    """Random band patterns of every tile shape the kernels are built for — (3,0,0) (3,1,1) (5,0,0) (5,1,1) (7,0,0) (7,1,1):
    near diagonals anywhere within +-510, far ones beyond — on 100-300 k rows, with random damaged rows (an entry missing, a
    row reduced to its diagonal with a value of its own, adjacent pairs of those) that cut the runs of the pattern in random
    places, in both streams (constant diagonals: pair codes; a value per entry: offset codes).  Tile plan or not, whatever the
    plan builder made of the runs: y bit-identical to the oracle, with and without the fused dot."""
    rng = np.random.default_rng(1000 + seed)
    ctx = sa.default_ctx(0)
    shapes = [(3, 0, 0), (3, 1, 1), (5, 0, 0), (5, 1, 1), (7, 0, 0), (7, 1, 1)]
    ul, fl, fh = shapes[seed % 6]
    n = int(rng.integers(100_000, 300_000))
    near = sorted(set([0] + [int(v) for v in rng.integers(-510, 511, size=4 * ul)]))
    nn = ul - fl - fh
    while len(near) > nn:
        near.pop(int(rng.integers(0, len(near))))
    if 0 not in near:
        near[len(near) // 2] = 0
    near = sorted(set(near))
    assert len(near) <= nn
    far_lo = [-int(rng.integers(2_000, 40_000))] if fl else []
    far_hi = [int(rng.integers(2_000, 40_000))] if fh else []
    offs = far_lo + near + far_hi
    diag_vals = rng.integers(-3, 4, size=len(offs)).astype(np.float64); diag_vals[diag_vals == 0] = 2.0
    indptr, cols, data = _banded_with_defects(rng, n, offs, diag_vals, n_defects=int(rng.integers(0, 60)))
    x = rand_vec(n, np.float64, 500 + seed)
    try:
        ctx.set("spmv_tile", 1)
        for stream in ("pair", "offsets"):
            vals = data if stream == "pair" else data * rng.uniform(0.5, 1.5, data.size)
            ref = oracle.spmv(indptr, cols, vals, x)
            A = sa.HipCsr.new((n, n), indptr, cols, vals)
            assert A.stream_format()[0] == (2 if stream == "pair" else 1), (stream, A.stream_format())
            y = np.full(n, 7.0); A.mul_vec(x, y)
            bad = np.flatnonzero(y != ref)
            assert bad.size == 0, (stream, offs, A.tile_plan(), bad[:6], bad.size)
            assert np.array_equal(bits(y), bits(ref))
            y2 = np.full(n, 7.0); d = A.mul_vec_dot(x, y2)
            assert np.array_equal(bits(y2), bits(ref))
            e = oracle.conj_dot(x, ref)
            assert abs(d - e) <= 1e-12 * max(1.0, float(np.sum(np.abs(x * ref))))
            if len(near) == nn and len(offs) == ul:
                assert A.tile_plan()[0] >= 8, (stream, offs, A.tile_plan())      # long runs survive ~60 damaged rows in 100 k+
    finally:
        ctx.set("spmv_tile", -1)


@pytest.mark.parametrize("stream", ["pair", "offsets"])
def test_lds_window_tiles_unaligned_device_vectors(sa, oracle, stream):
    """x, y (and the dot operand) as device views that start 8 bytes into an allocation: the tile kernels' 16-byte window / far /
    operand loads and y stores then sit on 8-byte boundaries only.  Same bits."""
    import torch
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    ip, ix, d, _ = gen.poisson3d(160, 128, 12)
    n = ip.size - 1
    if stream == "offsets":
        d = d * np.random.default_rng(5).uniform(0.5, 1.5, d.size)
    xh = rand_vec(n, np.float64, 91)
    ref = oracle.spmv(ip, ix, d, xh)
    dev = torch.device("cuda", 0)
    try:
        ctx.set("spmv_tile", 1)
        A = sa.HipCsr.new((n, n), ip, ix, d)
        assert A.tile_plan()[0] >= 8 and A.stream_format()[0] == (2 if stream == "pair" else 1)
        for off in (1, 3):
            bx = torch.zeros(n + 8, dtype=torch.float64, device=dev); by = torch.full((n + 8,), 5.0, dtype=torch.float64, device=dev)
            x = bx[off:off + n]; y = by[off:off + n]
            x.copy_(torch.from_numpy(xh))
            assert x.data_ptr() % 16 == 8
            A.mul_vec_unchecked(x, y)
            assert np.array_equal(bits(y.cpu().numpy()), bits(ref)), off
            assert float(by[off - 1]) == 5.0 and float(by[off + n]) == 5.0          # nothing written outside the view
            y.fill_(0.0)
            dd = A.mul_vec_dot_unchecked(x, y)
            assert np.array_equal(bits(y.cpu().numpy()), bits(ref))
            e = oracle.conj_dot(xh, ref)
            assert abs(dd - e) <= 1e-12 * max(1.0, float(np.sum(np.abs(xh * ref))))
    finally:
        ctx.set("spmv_tile", -1)


@pytest.mark.parametrize("name", ["p3_800x64x8", "p2_dirichlet_1500"])
def test_lds_window_tiles_wide_window(sa, oracle, name):
    """Grids whose lines are 511 to 1534 rows long: the pair-code tile kernel takes a window of half-width 1536 (the +-nx columns
    are near again: one staged window instead of a far load per row pair each); the offset-code stream keeps the per-block
    kernel there.  y bit-identical either way."""
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    if name == "p3_800x64x8":
        indptr, cols, data, _ = gen.poisson3d(800, 64, 8)
    else:
        indptr, cols, data = gen.grid_laplacian_dirichlet(1500, 1500)
    n = indptr.size - 1
    x = rand_vec(n, np.float64, 123)
    try:
        ctx.set("spmv_tile", 1)
        for stream in ("pair", "offsets"):
            vals = data if stream == "pair" else data * np.random.default_rng(3).uniform(0.5, 1.5, data.size)
            ref = oracle.spmv(indptr, cols, vals, x)
            A = sa.HipCsr.new((n, n), indptr, cols, vals)
            assert A.stream_format()[0] == (2 if stream == "pair" else 1)
            nt = A.tile_plan()[0]
            # (offset codes: window 512 only — the 3-D grid's +-800 would be two far bands a side, not built; the 2-D grid's +-1500 are its one far band)
            assert (nt >= 8) if (stream == "pair" or name == "p2_dirichlet_1500") else (nt == 0), (stream, A.tile_plan())
            y = np.full(n, 3.0); A.mul_vec(x, y)
            assert np.array_equal(bits(y), bits(ref)), stream
            y2 = np.zeros(n); d = A.mul_vec_dot(x, y2)
            assert np.array_equal(bits(y2), bits(ref))
            e = oracle.conj_dot(x, ref)
            assert abs(d - e) <= 1e-12 * max(1.0, float(np.sum(np.abs(x * ref))))
    finally:
        ctx.set("spmv_tile", -1)
