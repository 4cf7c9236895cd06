"""Deep fuzz of every SpMV stream / kernel / knob combination against the oracle's fold (bit-exact y), far longer than
the test suite's 200 matrices: structured matrices that hit the uniform-block paths (offset-code and pair-code), the
equal-length-row shortcut of the plain kernel, the XCD-period schedule, ragged and empty rows, all four scalar types.
  usage (GPU box): python scripts/fuzz_spmv.py [seconds] [seed]      -> prints a summary, exits 1 on the first mismatch"""
import itertools
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sprsolve_amd as sa            # noqa: E402
from oracle import oracle           # noqa: E402

rng = np.random.default_rng(12345)     # re-seeded by run()
DT = [np.float64, np.complex128, np.float32, np.complex64]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _stencil_rows_vectorised(n, offs, kind, line, seam_kind):
    """The row loop of make() for kinds 5 / 6 (no per-row random draws) as array operations: the same (indptr, cols, vals)
    — checked against the loop by tests/test_fuzz_generators.py — at a hundredth of the time for the 70-160 k-row matrices
    that reach the LDS-window tiles."""
    r = np.arange(n, dtype=np.int64)[:, None]
    c = r + offs[None, :].astype(np.int64)
    keep = (c >= 0) & (c < n)
    if kind == 6 and line > 0:
        cnt = keep.sum(axis=1)
        rank = np.cumsum(keep, axis=1) - 1                       # position of an entry among its row's kept ones
        many = cnt > 1
        ph = (r[:, 0] % line)
        if seam_kind == 2:
            sel = many & ((ph == 0) | (ph == line - 1))
            want = np.where(r[:, 0] % 3 != 0, cnt // 2, 0)
            keep = np.where(sel[:, None], keep & (rank == want[:, None]), keep)
        else:
            last = many & (ph == line - 1)
            first = many & (ph == 0) & ~last
            if seam_kind:
                keep = np.where(last[:, None], keep & (rank != (cnt - 1)[:, None]), keep)
                keep = np.where(first[:, None], keep & (rank != 0), keep)
            else:
                keep = np.where(last[:, None], keep & (c != r + 1), keep)
                keep = np.where(first[:, None], keep & (c != r - 1), keep)
    cnt = keep.sum(axis=1)
    indptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(cnt, out=indptr[1:])
    cols = c[keep].astype(np.int32)
    rr = np.broadcast_to(r, c.shape)[keep]
    vals = (cols - rr) * 0.375 + np.where(np.repeat(cnt, cnt) > 1, 1.0, 7.5)
    return indptr, cols, vals


def make(n, kinds=None, vectorised=True):
    kind = rng.integers(0, 7) if kinds is None else int(rng.choice(kinds))
    nb = int(rng.integers(1, 9))
    offs = np.unique(np.concatenate([[0], rng.integers(-min(n - 1, 40), min(n - 1, 40) + 1, size=nb)]))
    if kind >= 5 and n > 8:         # stencil-like: runs of consecutive columns (column triples), at most 8 per row
        o0 = int(rng.integers(-min(n - 1, 40), min(n - 1, 40) - 3))
        offs = np.unique(np.concatenate([offs[:int(rng.integers(0, 4))], o0 + np.arange(int(rng.integers(3, 6)))]))[:8]
    if kind == 3 and n > 600:       # far band (period schedule, plane-like)
        offs = np.unique(np.concatenate([offs, [-(n // 5), n // 5]]))
    if kinds is not None:           # big stencil-like matrices for the LDS-window tiles: at most 8 slots, at most one far band a side
        near = [int(o) for o in offs if abs(o) <= 40][:6]
        offs = np.unique(np.array(([-(n // int(rng.integers(4, 9)))] if rng.uniform() < 0.7 else []) + near +
                                  ([n // int(rng.integers(4, 9))] if rng.uniform() < 0.7 else [])))
    line = int(rng.integers(130, 700)) if kind == 6 else 0
    seam_kind = int(rng.integers(0, 3))
    rows = []
    ragged_lo, ragged_hi = (int(rng.integers(0, n)), int(rng.integers(0, n)))
    ragged_lo, ragged_hi = min(ragged_lo, ragged_hi), max(ragged_lo, ragged_hi)
    if kinds is not None and vectorised:
        return _stencil_rows_vectorised(n, offs, kind, line, seam_kind)
    for r in range(n):
        c = r + offs
        c = c[(c >= 0) & (c < n)]
        if kind in (1, 4) and ragged_lo <= r < ragged_hi:
            c = c[rng.uniform(size=c.size) < 0.6]
        if kind == 2 and rng.uniform() < 0.02:
            c = c[:0]
        if kind == 6 and line > 0:        # truncated lines: the last row of a line lacks its largest offset, the first its smallest
            if seam_kind == 2:            # ... or Dirichlet rows: the two rows at a seam hold one entry (not always the diagonal)
                if r % line in (0, line - 1) and c.size > 1:
                    c = c[[int(c.size // 2) if r % 3 else 0]]
            elif r % line == line - 1 and c.size > 1:
                c = c[:-1] if seam_kind else c[c != r + 1]
            elif r % line == 0 and c.size > 1:
                c = c[1:] if seam_kind else c[c != r - 1]
        rows.append(c)
    indptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum([len(c) for c in rows], out=indptr[1:])
    cols = np.concatenate(rows).astype(np.int32) if indptr[-1] else np.zeros(0, np.int32)
    if kind >= 5:                   # one value per offset: interior blocks are uniform; single-entry rows carry a value of their own
        vals = (np.concatenate([(c - r) * 0.375 + (1.0 if c.size > 1 else 7.5) for r, c in enumerate(rows)])) if indptr[-1] else np.zeros(0)
    elif rng.uniform() < 0.5:
        vals = np.array([1.0, -1.0, 0.5, 2.0, -3.25, 6.0])[rng.integers(0, 6, cols.size)]      # value dictionary
    else:
        vals = rng.uniform(-1, 1, cols.size)                                                 # none
    return indptr, cols, vals


def run(budget=60.0, seed=12345, max_matrices=None, big_prob=0.04, small=True, verbose=True):
    """Fuzz for `budget` seconds or `max_matrices` matrices.  big_prob: share of 70-160 k-row stencil-like f64 matrices (the
    LDS-window tile plans); small=False skips the full knob product of the small matrices.  Returns a summary dict
    (`mismatch` must be None)."""
    global rng
    rng = np.random.default_rng(seed)
    ctx = sa.default_ctx(0)
    t_end = time.time() + budget
    count = combos = tiled = 0
    mismatch = None
    t_say = time.time() + 60.0
    try:
        while time.time() < t_end and (max_matrices is None or count < max_matrices) and mismatch is None:
            if time.time() > t_say:                      # a sign of life a minute (a silent GPU job is taken for a hung one)
                print("... %d matrices, %d combinations so far" % (count, combos), flush=True)
                t_say = time.time() + 60.0
            big = rng.uniform() < big_prob          # now and then a matrix long enough for tiles (knob spmv_tile), f64
            if big:
                n = int(rng.integers(70_000, 160_000))
                indptr, cols, vals = make(n, kinds=(5, 6))
                if rng.uniform() < 0.5:
                    vals = vals * rng.uniform(0.5, 1.5, vals.size)           # a value per entry: the offset-code tiles
            else:
                n = int(rng.choice([rng.integers(1, 400), rng.integers(400, 6000), rng.integers(6000, 40000)]))
                indptr, cols, vals = make(n)
            if indptr[-1] == 0:
                continue
            dtype = np.float64 if big else DT[int(rng.integers(0, 4))]
            d = vals.astype(dtype)
            if np.dtype(dtype).kind == "c":
                d = d * (1 + 0.25j)
            x = rng.uniform(-1, 1, n).astype(dtype)
            if np.dtype(dtype).kind == "c":
                x = x + 1j * rng.uniform(-1, 1, n).astype(x.real.dtype)
            ref = oracle.spmv(indptr, cols, d, x)
            grid = (itertools.product((1, 2), (1,), (1,), (1,), (0, 1), (1,), (0, 1), (1,), (0, 1)) if big else
                    itertools.product((0, 1, 2), (0, 1), (0, 1), (0, 1), (0, 1), (0, 1), (0, 1), (0, 1), (-1,)) if small else
                    itertools.product((0, 1, 2), (1,), (1,), (1,), (0,), (1,), (1,), (1,), (-1,)))
            for knob, wide, uni, eq, period, tri, seam, wl, tile in grid:
                if (knob != 2 and (wide or period)) and not big and small:
                    continue
                if small and ((knob == 0 and uni) or (knob != 0 and eq == 0) or (not tri and not wide) or (not seam and not (wide and uni)) or (knob != 0 and wl == 0)):
                    continue
                for k, v in (("spmv_dict", knob), ("spmv_wide", wide), ("spmv_uniform", uni), ("spmv_eqrows", eq), ("spmv_period", period),
                             ("spmv_triple", tri), ("spmv_seam", seam), ("spmv_wideload", wl), ("spmv_tile", tile)):
                    ctx.set(k, v)
                A = sa.HipCsr.new((n, n), indptr, cols, d)
                y = np.full(n, 7.0, dtype=dtype)
                A.mul_vec(x, y)
                y2 = np.zeros(n, dtype=dtype)
                A.mul_vec_dot(x, y2)
                combos += 1
                tiled += A.tile_plan()[0] > 0
                if not (np.array_equal(bits(y), bits(ref)) and np.array_equal(bits(y2), bits(ref))):
                    mismatch = dict(text="MISMATCH n=%d dtype=%s knobs dict=%d wide=%d uniform=%d eqrows=%d period=%d triple=%d seam=%d wideload=%d tile=%d stream=%s plan=%s bad=%d" % (
                        n, np.dtype(dtype).name, knob, wide, uni, eq, period, tri, seam, wl, tile, A.stream_format(), A.tile_plan(), int(np.sum(bits(y) != bits(ref)))),
                        arrays=dict(indptr=indptr, cols=cols, d=d, x=x))
                    break
            count += 1
    finally:
        for k in ("spmv_dict", "spmv_wide", "spmv_uniform", "spmv_eqrows", "spmv_wideload", "spmv_period", "spmv_triple", "spmv_seam", "spmv_tile"):
            ctx.set(k, -1)
    return dict(matrices=count, combos=combos, tiled=int(tiled), mismatch=mismatch)


if __name__ == "__main__":
    r = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    if r["mismatch"]:
        import os
        print(r["mismatch"]["text"])
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez("gpurun_out/fuzz_fail.npz", **r["mismatch"]["arrays"])
        sys.exit(1)
    print("fuzz ok: %d matrices, %d (matrix, knob) combinations (%d of them through LDS-window tiles), all y bit-identical to the reference fold"
          % (r["matrices"], r["combos"], r["tiled"]))
