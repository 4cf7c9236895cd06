"""Device-side (torch) twins of the big generators in gen.py, so BASELINE-size matrices
(cfg 5: 50 M rows, 349 M nnz) are built directly in HBM instead of crossing PCIe.
torch is plumbing here (device memory + a few tensor ops); tests check these against gen.py."""
import torch


def _splitmix64_keys(seed, keys, stream):
    """gen.splitmix64_keys on int64 tensors: two's-complement wrap-around = arithmetic mod 2^64; logical right
    shifts are emulated with a mask."""
    def c(v):   # uint64 constant as int64
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)
    base = c((seed & 0xFFFFFFFFFFFFFFFF) + stream * 0xD1B54A32D192ED03)
    z = (keys + 1) * c(0x9E3779B97F4A7C15) + base
    z = (z ^ lsr(z, 30)) * c(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * c(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def _uniform_keys(seed, keys, stream):
    u = ((_splitmix64_keys(seed, keys, stream) >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / 9007199254740992.0)
    return -1.0 + 2.0 * u


def poisson3d(nx, ny, nz, z0=0, z1=None, device="cuda", values="poisson", seed=0x5052534F4C5645):
    """gen.poisson3d on `device`: CSR row block for planes [z0, z1) of the 7-point Poisson
    matrix (diag +6, neighbours -1, truncated at the faces), GLOBAL int32 column indices,
    rhs = A*1.  values="random": gen.poisson3d(values="random") — bit-identical values (tests/test_generators.py).
    Returns (indptr:int32, indices:int32, data:f64, rhs:f64)."""
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
    cand = (g[:, None] + offs[None, :])
    indices = cand[ok].to(torch.int32)
    del cand
    if values == "random":
        slot = torch.arange(7, dtype=torch.int64, device=device)
        vals = _uniform_keys(seed, g[:, None] * 7 + slot[None, :], 50)
        vals = torch.where(ok, vals, torch.zeros((), dtype=torch.float64, device=device))
        vals[:, 3] = 0.0
        # the row's |.| sum in slot order, like numpy's axis-1 sum of 7 elements (pairwise == sequential below 8 terms)
        acc = torch.zeros(g.numel(), dtype=torch.float64, device=device)
        for j in range(7):
            acc = acc + vals[:, j].abs()
        vals[:, 3] = 1.0 + acc
        rhs = torch.zeros(g.numel(), dtype=torch.float64, device=device)
        for j in range(7):
            rhs = rhs + vals[:, j]
        data = vals[ok].contiguous()
        del vals, acc
    else:
        rhs = 6.0 - (cnt - 1).to(torch.float64)
        vals = torch.tensor([-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0], dtype=torch.float64, device=device)
        data = vals[None, :].expand(ok.shape[0], 7)[ok].contiguous()
    del g
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
