"""Row partition of a CSR matrix over ranks and the halo-exchange plan derived from it.

Host-side logic of the multi-GPU path (no reference analogue — the reference is
single-process).  Pure index arithmetic, written once against a tiny array-module shim so the
same code runs on numpy arrays (CPU tests, world_size-2 gloo) and on torch CUDA tensors (the
50 M-row bench: 43 M column indices per rank are localised in HBM, not on the host).

Vocabulary: rank r owns rows/entries [starts[r], starts[r+1]).  Its EXTENDED x vector is
    [ owned entries (n_local) | entries needed from peer p0 | from peer p1 | ... ]
with peers in ascending rank order and, per peer, the needed global indices ascending.
"""
import numpy as np


def row_starts(n, world):
    """Balanced contiguous row ranges: starts[r] = floor(r*n/world)."""
    return np.array([(r * n) // world for r in range(world + 1)], dtype=np.int64)


def slab_starts(nz, plane, world):
    """z-slab partition of an nx*ny*nz grid (plane = nx*ny rows): whole planes per rank."""
    return np.array([((r * nz) // world) * plane for r in range(world + 1)], dtype=np.int64)


class _NP:
    @staticmethod
    def unique(a): return np.unique(a)
    @staticmethod
    def searchsorted(a, v): return np.searchsorted(a, v)
    @staticmethod
    def where(c, a, b): return np.where(c, a, b)
    @staticmethod
    def to_numpy(a): return np.asarray(a)
    @staticmethod
    def as_i32(a): return a.astype(np.int32)
    @staticmethod
    def as_i64(a): return a.astype(np.int64)
    @staticmethod
    def from_numpy_like(a, like): return a


class _TORCH:
    def __init__(self):
        import torch
        self.t = torch
    def unique(self, a): return self.t.unique(a)          # sorted ascending
    def searchsorted(self, a, v): return self.t.searchsorted(a, v)
    def where(self, c, a, b): return self.t.where(c, a, b)
    def to_numpy(self, a): return a.cpu().numpy()
    def as_i32(self, a): return a.to(self.t.int32)
    def as_i64(self, a): return a.to(self.t.int64)
    def from_numpy_like(self, a, like): return self.t.from_numpy(np.ascontiguousarray(a)).to(like.device)


def _xp(a):
    return _NP if isinstance(a, np.ndarray) else _TORCH()


def localize(col_global, starts, rank):
    """Renumber the GLOBAL column indices of rank's row block into its extended-local numbering.

    Returns (col_ext:int32 same array type, needed: {peer: sorted unique GLOBAL indices (numpy int64)},
             recv_off: numpy int64 [n_peers+1], peers: list[int]).
    """
    xp = _xp(col_global)
    r0, r1 = int(starts[rank]), int(starts[rank + 1])
    n_local = r1 - r0
    cg = xp.as_i64(col_global)
    remote = (cg < r0) | (cg >= r1)
    uniq = xp.unique(cg[remote])                       # ascending global ids => grouped by owner rank
    uniq_np = np.asarray(xp.to_numpy(uniq), dtype=np.int64)
    owner = np.searchsorted(np.asarray(starts, dtype=np.int64), uniq_np, side="right") - 1
    peers = sorted(set(owner.tolist()))
    needed, recv_off = {}, [0]
    for p in peers:
        sel = uniq_np[owner == p]
        needed[p] = sel
        recv_off.append(recv_off[-1] + sel.size)
    # position of each remote column inside `uniq` IS its offset in the halo tail (peers ascending,
    # indices ascending within a peer == global ascending order)
    if uniq_np.size:
        pos = xp.searchsorted(uniq, cg)
        # clamp for the owned entries (their searchsorted result is meaningless but must be in range)
        col_ext = xp.where(remote, pos + n_local, cg - r0)
    else:
        col_ext = cg - r0
    return xp.as_i32(col_ext), needed, np.asarray(recv_off, dtype=np.int64), peers


def send_plan(all_needed, starts, rank):
    """From every rank's `needed` dict (gathered over the ranks, e.g. all_gather_object), the
    entries THIS rank must pack for each peer: (peers, send_off int64 [n+1], send_idx int32 local)."""
    r0 = int(starts[rank])
    peers, off, idx = [], [0], []
    for q, need in enumerate(all_needed):
        if q == rank or rank not in need:
            continue
        loc = np.asarray(need[rank], dtype=np.int64) - r0
        peers.append(q)
        idx.append(loc.astype(np.int32))
        off.append(off[-1] + loc.size)
    send_idx = np.concatenate(idx) if idx else np.zeros(0, np.int32)
    return peers, np.asarray(off, dtype=np.int64), send_idx


def merge_peers(recv_peers, recv_off, send_peers, send_off, send_idx):
    """One peer list for both directions (a peer may only send or only receive):
    returns (peers, send_off, send_idx, recv_off) aligned on `peers`."""
    peers = sorted(set(recv_peers) | set(send_peers))
    s_off, r_off, s_idx = [0], [0], []
    for p in peers:
        if p in send_peers:
            k = send_peers.index(p)
            s_idx.append(send_idx[send_off[k]:send_off[k + 1]])
            s_off.append(s_off[-1] + int(send_off[k + 1] - send_off[k]))
        else:
            s_off.append(s_off[-1])
        if p in recv_peers:
            k = recv_peers.index(p)
            r_off.append(r_off[-1] + int(recv_off[k + 1] - recv_off[k]))
        else:
            r_off.append(r_off[-1])
    sidx = np.concatenate(s_idx).astype(np.int32) if s_idx else np.zeros(0, np.int32)
    return peers, np.asarray(s_off, np.int64), sidx, np.asarray(r_off, np.int64)


def build_plan(col_global, starts, rank, all_gather_object):
    """Full setup for one rank.  `all_gather_object(obj) -> list` is the only communication
    (torch.distributed.all_gather_object under gloo or nccl).
    Returns dict(col_ext, n_local, n_ext, peers, send_off, send_idx, recv_off)."""
    col_ext, needed, recv_off, recv_peers = localize(col_global, starts, rank)
    all_needed = all_gather_object(needed)
    send_peers, send_off, send_idx = send_plan(all_needed, starts, rank)
    peers, s_off, s_idx, r_off = merge_peers(recv_peers, recv_off, send_peers, send_off, send_idx)
    n_local = int(starts[rank + 1] - starts[rank])
    return dict(col_ext=col_ext, n_local=n_local, n_ext=n_local + int(r_off[-1]), peers=peers,
                send_off=s_off, send_idx=s_idx, recv_off=r_off)


def allgather_plan(col_global, starts, rank):
    """The literal north_star exchange: every rank all-gathers every x slice.  Column indices are renumbered into
    the gathered vector [rank 0 slice | rank 1 slice | ...] whose slices are padded to `slice` = max rows per rank
    (rounded up to 2 so every slice stays 16-byte aligned).  Returns dict(col_ext, n_local, slice, mode)."""
    xp = _xp(col_global)
    st = np.asarray(starts, dtype=np.int64)
    slice_ = int(np.max(np.diff(st)))
    slice_ += slice_ & 1
    cg = xp.as_i64(col_global)
    if isinstance(col_global, np.ndarray):
        owner = np.searchsorted(st, cg, side="right") - 1
        col = owner * slice_ + (cg - st[owner])
    else:
        t = xp.t
        stt = t.from_numpy(st).to(cg.device)
        owner = t.searchsorted(stt, cg, right=True) - 1
        col = owner * slice_ + (cg - stt[owner])
    return dict(col_ext=xp.as_i32(col), n_local=int(st[rank + 1] - st[rank]), slice=slice_, mode="allgather")
