"""Device-side (torch) twins of the big generators in gen.py, so BASELINE-size matrices
(cfg 5: 50 M rows, 349 M nnz) are built directly in HBM instead of crossing PCIe.
torch is plumbing here (device memory + a few tensor ops); tests check these against gen.py."""
import torch


def poisson3d(nx, ny, nz, z0=0, z1=None, device="cuda"):
    """gen.poisson3d on `device`: CSR row block for planes [z0, z1) of the 7-point Poisson
    matrix (diag +6, neighbours -1, truncated at the faces), GLOBAL int32 column indices,
    rhs = A*1.  Returns (indptr:int32, indices:int32, data:f64, rhs:f64)."""
    if z1 is None:
        z1 = nz
    plane = nx * ny
    assert nx * ny * nz < 2**31
    g = torch.arange(z0 * plane, z1 * plane, dtype=torch.int64, device=device)
    x = g % nx
    y = (g // nx) % ny
    z = g // plane
    offs = torch.tensor([-plane, -nx, -1, 0, 1, nx, plane], dtype=torch.int64, device=device)
    ok = torch.stack([z > 0, y > 0, x > 0, torch.ones_like(g, dtype=torch.bool), x < nx - 1, y < ny - 1, z < nz - 1], dim=1)
    del x, y, z
    cnt = ok.sum(dim=1)
    indptr = torch.zeros(g.numel() + 1, dtype=torch.int64, device=device)
    torch.cumsum(cnt, 0, out=indptr[1:])
    rhs = 6.0 - (cnt - 1).to(torch.float64)
    cand = (g[:, None] + offs[None, :])
    indices = cand[ok].to(torch.int32)
    del cand, g
    vals = torch.tensor([-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0], dtype=torch.float64, device=device)
    data = vals[None, :].expand(ok.shape[0], 7)[ok].contiguous()
    return indptr.to(torch.int32), indices, data, rhs


def grid_laplacian_dirichlet(rows, cols, device="cuda"):
    """gen.grid_laplacian_dirichlet + gen.dirichlet_rhs on `device` (square grids, as upstream).
    Returns (indptr, indices, data, rhs, diag)."""
    assert rows == cols
    g = torch.arange(rows * cols, dtype=torch.int64, device=device)
    i = g // cols
    j = g % cols
    border = (i == 0) | (i + 1 == rows) | (j == 0) | (j + 1 == cols)
    offs = torch.tensor([-rows, -1, 0, 1, rows], dtype=torch.int64, device=device)
    ok = (~border)[:, None].expand(-1, 5).clone()
    ok[:, 2] = True
    cnt = ok.sum(dim=1)
    indptr = torch.zeros(g.numel() + 1, dtype=torch.int64, device=device)
    torch.cumsum(cnt, 0, out=indptr[1:])
    indices = (g[:, None] + offs[None, :])[ok].to(torch.int32)
    vals = torch.tensor([1.0, 1.0, -4.0, 1.0, 1.0], dtype=torch.float64, device=device)[None, :].expand(g.numel(), 5).clone()
    vals[border, 2] = 1.0
    data = vals[ok].contiguous()
    rhs = torch.where(border, (i + j).to(torch.float64), torch.zeros((), dtype=torch.float64, device=device))
    diag = torch.where(border, 1.0, -4.0).to(torch.float64)
    return indptr.to(torch.int32), indices, data, rhs, diag
