"""A bounded, seeded slice of the two fuzzers (scripts/fuzz_solvers.py, scripts/fuzz_spmv.py) inside the suite the driver
runs.  Round 3's MINRES race — the converging launch's workgroup 0 publishing "converged" before late workgroups of the same
launch had read the status word — was found by the long fuzz runs and by no committed test: this slice biases the solver
fuzz towards solves that converge within their first iterations (all three solvers, fused and literal, four scalar types,
plain and Jacobi) and the SpMV fuzz towards 70-160 k-row stencil-like matrices whose launches go through LDS-window tile
plans (both compressed streams, seam / period / tile knobs).  0 hard mismatches against the oracle, y bit-identical."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_solver_fuzz_slice_early_convergence(oracle):
    F = _load("fuzz_solvers")
    # mode 3 (tolerance 0.5 / 0.3 / 0.1: Ok within the first iterations) eight times as likely as each other mode
    r = F.run(budget=14.0, seed=404, mode_weights=[1, 1, 1, 8, 1, 1])
    assert r["hard"] == 0, r
    assert r["cases"] >= 300, r                 # ~1.5-2 k solves on an MI355X box; a floor that still means something
    assert r["by_status"].get(0, 0) >= r["cases"] // 2


def test_solver_fuzz_slice_default_mix(oracle):
    F = _load("fuzz_solvers")
    r = F.run(budget=6.0, seed=77)
    assert r["hard"] == 0 and r["cases"] >= 100, r


def test_spmv_fuzz_slice_tile_plans(oracle):
    F = _load("fuzz_spmv")
    r = F.run(budget=14.0, seed=9, big_prob=1.0)
    assert r["mismatch"] is None, r["mismatch"] and r["mismatch"]["text"]
    assert r["tiled"] >= 100, r                 # (matrix, knob) combinations whose SpMV ran through a tile plan; 2 SpMVs each


def test_spmv_fuzz_slice_small_matrices(oracle):
    F = _load("fuzz_spmv")
    r = F.run(budget=6.0, seed=10, big_prob=0.0)
    assert r["mismatch"] is None, r["mismatch"] and r["mismatch"]["text"]
    assert r["combos"] >= 300, r
