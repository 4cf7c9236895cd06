"""Matrix ingest: MatrixMarket coordinate files <-> CSR arrays.

The reference's tests hint at this format (`sprs::io::write_matrix_market`,
tests/test_complex_solve.rs:14).  Host-side only; the arrays go to `HipCsr.new`.
Within-row order after reading is ascending column — what `sprs::TriMat::to_csr` produces."""
import numpy as np


def read_matrix_market(path):
    """-> (shape, indptr:int32, indices:int32, data:float64|complex128).  Coordinate format;
    fields real / integer / complex / pattern; symmetries general / symmetric / hermitian /
    skew-symmetric (off-diagonal entries are mirrored).  Duplicate entries are summed."""
    with open(path, "r") as f:
        header = f.readline().strip().lower().split()
        if len(header) < 5 or header[0] != "%%matrixmarket" or header[1] != "matrix":
            raise ValueError("not a MatrixMarket matrix file")
        fmt, field, sym = header[2], header[3], header[4]
        if fmt != "coordinate":
            raise ValueError("only the coordinate (sparse) format is supported")
        line = f.readline()
        while line.startswith("%") or not line.strip():
            line = f.readline()
        nrows, ncols, nnz = (int(t) for t in line.split())
        ncol_txt = {"real": 3, "integer": 3, "complex": 4, "pattern": 2}[field]
        raw = np.loadtxt(f, dtype=np.float64, ndmin=2) if nnz else np.zeros((0, ncol_txt))
    if raw.shape[0] != nnz or (nnz and raw.shape[1] != ncol_txt):
        raise ValueError("malformed MatrixMarket body")
    r = raw[:, 0].astype(np.int64) - 1
    c = raw[:, 1].astype(np.int64) - 1
    if field == "complex":
        v = raw[:, 2] + 1j * raw[:, 3]
    elif field == "pattern":
        v = np.ones(nnz)
    else:
        v = raw[:, 2].copy()
    if sym != "general":
        off = r != c
        mv = {"symmetric": v[off], "hermitian": np.conj(v[off]), "skew-symmetric": -v[off]}[sym]
        r, c, v = np.concatenate([r, c[off]]), np.concatenate([c, r[:nnz][off]]), np.concatenate([v, mv])
    if r.size and (r.min() < 0 or r.max() >= nrows or c.min() < 0 or c.max() >= ncols):
        raise ValueError("index out of range")
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    if r.size:                                  # sum duplicates (TriMat semantics)
        first = np.ones(r.size, dtype=bool)
        first[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
        grp = np.cumsum(first) - 1
        vs = np.zeros(int(grp[-1]) + 1, dtype=v.dtype)
        np.add.at(vs, grp, v)
        r, c, v = r[first], c[first], vs
    indptr = np.zeros(nrows + 1, dtype=np.int64)
    np.add.at(indptr, r + 1, 1)
    np.cumsum(indptr, out=indptr)
    return (nrows, ncols), indptr.astype(np.int32), c.astype(np.int32), v


def write_matrix_market(path, shape, indptr, indices, data):
    """CSR -> MatrixMarket coordinate general (real or complex), 17 significant digits."""
    data = np.asarray(data)
    cplx = np.iscomplexobj(data)
    rows = np.repeat(np.arange(shape[0]), np.diff(np.asarray(indptr, dtype=np.int64)))
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s general\n" % ("complex" if cplx else "real"))
        f.write("%d %d %d\n" % (shape[0], shape[1], data.size))
        for i, j, v in zip(rows, np.asarray(indices), data):
            if cplx:
                f.write("%d %d %.17g %.17g\n" % (i + 1, j + 1, v.real, v.imag))
            else:
                f.write("%d %d %.17g\n" % (i + 1, j + 1, v))
