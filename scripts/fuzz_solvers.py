"""Randomised solver-level parity: BiCGStab / MINRES / CSMINRES (plain and Jacobi, fused and literal mode) on small
random systems of all four scalar types against the oracle's restatement of the reference recurrences — outcome
(Ok / InsufficientIterNum / BreakDown / InvalidPreconditioner), iteration count, residual and solution.
Sizes straddle the 64-row / 128-row block edges; degenerate inputs (zero rhs, one row, max_iter 0 and 1, loose and
unreachable tolerances, non-zero initial guess) are mixed in.
  usage (GPU box): python scripts/fuzz_solvers.py [seconds] [seed]     -> summary; exits 1 on the first hard mismatch"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sprsolve_amd as sa            # noqa: E402
from oracle import oracle           # noqa: E402
from sprsolve_amd import error as E  # noqa: E402

rng = np.random.default_rng(2024)      # re-seeded by run()
SIZES = [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 1000]


def cplx(shape, dtype):
    if np.dtype(dtype).kind == "c":
        return (rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)).astype(dtype)
    return rng.uniform(-1, 1, shape).astype(dtype)


def banded(n, dtype, symmetric, complex_symmetric):
    """Strictly diagonally dominant band with random half-bandwidth; symmetric (real / Hermitian-free complex-symmetric)
    on request.  CSR with sorted columns."""
    hbw = int(rng.integers(0, min(n, 5)))
    dense = {}
    for k in range(1, hbw + 1):
        up = cplx(n - k, dtype)
        lo = up if (symmetric or complex_symmetric) else cplx(n - k, dtype)
        for i in range(n - k):
            dense[(i, i + k)] = up[i]
            dense[(i + k, i)] = lo[i]
    rowsum = np.zeros(n)
    for (i, j), v in dense.items():
        rowsum[i] += abs(v)
    for i in range(n):
        d = 1.0 + rowsum[i] + rng.uniform(0, 1)
        dense[(i, i)] = np.dtype(dtype).type(d * (1 + (0.3j if complex_symmetric else 0)) if np.dtype(dtype).kind == "c" else d)
    keys = sorted(dense)
    indptr = np.zeros(n + 1, dtype=np.int32)
    for i, _ in keys:
        indptr[i + 1] += 1
    np.cumsum(indptr, out=indptr)
    cols = np.array([j for _, j in keys], dtype=np.int32)
    data = np.array([dense[k] for k in keys], dtype=dtype)
    return indptr, cols, data


def outcome_gpu(fn):
    try:
        its, res = fn()
        return oracle.OK, its, res
    except E.InsufficientIterNum as e:
        return oracle.INSUFFICIENT_ITER, e.iters, None
    except E.BreakDown as e:
        return oracle.BREAKDOWN, e.its, None
    except E.InvalidPreconditioner:
        return oracle.INVALID_PRECOND, -1, None


def run(budget=60.0, seed=2024, max_cases=None, mode_weights=None, sizes=None, verbose=True):
    """Fuzz for `budget` seconds or `max_cases` solves (whichever ends first; max_cases alone makes the run deterministic).
    mode_weights: relative weights of the six input modes (0 plain, 1 zero rhs, 2 tiny max_iter, 3 loose tolerance = converges
    within its first iterations, 4 non-zero x0, 5 plain); sizes: the size table.  Returns a summary dict; `hard` must be 0."""
    global rng
    rng = np.random.default_rng(seed)
    SZ = list(sizes) if sizes is not None else SIZES
    mw = np.asarray(mode_weights if mode_weights is not None else [1, 1, 1, 1, 1, 1], dtype=float)
    mw = mw / mw.sum()
    t_end = time.time() + budget
    cases = hard = soft = 0
    by_status = {}
    fail = None
    t_say = time.time() + 60.0
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        if verbose and time.time() > t_say:                 # a sign of life a minute (a silent GPU job is taken for a hung one)
            print("... %d solves so far, %d hard, %d soft" % (cases, hard, soft), flush=True)
            t_say = time.time() + 60.0
        dtype = [np.float64, np.complex128, np.float32, np.complex64][int(rng.integers(0, 4))]
        is_c = np.dtype(dtype).kind == "c"
        single = np.dtype(dtype).itemsize in (4, 8) and np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.complex64))
        kinds = ["bicgstab", "minres"] + (["csminres"] if is_c else [])
        kind = kinds[int(rng.integers(0, len(kinds)))]
        n = int(SZ[int(rng.integers(0, len(SZ)))]) if rng.uniform() < 0.7 else int(rng.integers(1, 400))
        # MINRES wants a symmetric (real) / Hermitian matrix: use real symmetric data even for complex T; CSMINRES complex-symmetric
        if kind == "minres":
            ip, ix, d = banded(n, np.float64 if not single else np.float32, True, False)
            d = d.astype(dtype)
        else:
            ip, ix, d = banded(n, dtype, False, kind == "csminres")
        mode = int(rng.choice(6, p=mw))
        rhs = cplx(n, dtype)
        x0 = np.zeros(n, dtype=dtype)
        max_iter, tol = 400, (1e-4 if single else 1e-10)
        if mode == 1:
            rhs = np.zeros(n, dtype=dtype)                  # rhs = 0: the early return (x = 0, Ok(0, 0))
        elif mode == 2:
            max_iter = int(rng.integers(0, 3))              # InsufficientIterNum (or a lucky Ok)
        elif mode == 3:
            tol = (0.5, 0.3, 0.1)[int(rng.integers(0, 3))]                                       # loose: Ok after very few iterations
        elif mode == 4:
            x0 = cplx(n, dtype)                             # non-zero initial guess
        use_pc = kind != "csminres" and rng.uniform() < 0.5
        pdiag = None
        if use_pc:
            pdiag = np.abs(np.array([d[ip[i]:ip[i + 1]][ix[ip[i]:ip[i + 1]] == i][0] for i in range(n)])).astype(
                np.float32 if single else np.float64)       # positive real diagonal (MINRES needs an SPD preconditioner)
        ref = getattr(oracle, kind)(ip, ix, d, rhs, x0, max_iter, tol, **({"precond_diag": pdiag} if use_pc else {}))
        A = sa.HipCsr.new((n, n), ip, ix, d)
        cls = {"bicgstab": sa.BiCGStab, "minres": sa.MinRes, "csminres": sa.CSMinRes}[kind]
        P = sa.DiagPrecond.new(pdiag, t_dtype=dtype) if use_pc else None
        for smode in ("fused", "literal"):
            s = cls.new(A, n); s.set_mode(smode)
            x = x0.copy()
            st, its, res = outcome_gpu((lambda: s.precond_solve(P, rhs, x, max_iter, tol)) if use_pc else (lambda: s.solve(rhs, x, max_iter, tol)))
            cases += 1
            by_status[st] = by_status.get(st, 0) + 1
            tag = "%s %s n=%d pc=%d mode=%d %s" % (kind, np.dtype(dtype).name, n, int(use_pc), mode, smode)
            scale = max(1.0, float(np.max(np.abs(ref.x))) if n else 1.0)
            xtol = (5e-3 if single else 1e-7) * scale
            if st != ref.status:
                # an outcome may legitimately flip only at a knife edge: convergence test within rounding of the tolerance
                knife = ref.status in (oracle.OK, oracle.INSUFFICIENT_ITER) and st in (oracle.OK, oracle.INSUFFICIENT_ITER) and mode == 2
                if not knife:
                    # ... or where the recurrence itself sits on an edge that the summation ORDER decides (a 3 x 3 system in f32 whose
                    # Krylov space is exhausted: beta^2 = 5e-9 against eps on one order, convergence on the other): the oracle with
                    # its reductions in the GPU kernels' order (oracle/krylov_tmpl.h) then takes the GPU's branch
                    oracle.set_reduction_order("gpu", int(sa.default_ctx(0).get("grid")))
                    try:
                        ref_g = getattr(oracle, kind)(ip, ix, d, rhs, x0, max_iter, tol, **({"precond_diag": pdiag} if use_pc else {}))
                    finally:
                        oracle.set_reduction_order("reference")
                    knife = ref_g.status == st and (st != oracle.OK or abs(its - ref_g.its) <= max(2, ref_g.its // 8))
                if knife:
                    soft += 1
                    continue
                print("HARD outcome mismatch:", tag, "gpu", st, its, "oracle", ref)
                hard += 1
                break
            if st == oracle.OK:
                bad_its = abs(its - ref.its) > max(2, ref.its // 8)
                bad_x = float(np.max(np.abs(x - ref.x))) > xtol if n else False
                if bad_its or bad_x:
                    print("HARD result mismatch:", tag, "gpu its %d res %.3e" % (its, res), "oracle", ref,
                          "max|dx| %.3e" % float(np.max(np.abs(x - ref.x))))
                    hard += 1
                    break
                if its != ref.its:
                    soft += 1
            elif st == oracle.INSUFFICIENT_ITER and its != ref.its:
                print("HARD its mismatch (InsufficientIterNum):", tag, its, ref)
                hard += 1
                break
        if hard:
            fail = dict(indptr=ip, cols=ix, d=d, rhs=rhs, x0=x0)
            break
    return dict(cases=cases, hard=hard, soft=soft, by_status={int(k): v for k, v in sorted(by_status.items())}, fail=fail)


if __name__ == "__main__":
    r = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    if r["hard"]:
        import os
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez("gpurun_out/fuzz_solver_fail.npz", **r["fail"])
        sys.exit(1)
    print("solver fuzz ok: %d solves (both modes), outcomes %s, %d differed from the oracle by an iteration or two (rounding), 0 hard mismatches"
          % (r["cases"], r["by_status"], r["soft"]))
