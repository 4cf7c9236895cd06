"""The vectorised big-matrix generator of scripts/fuzz_spmv.py (what makes ~10^2 tile-plan matrices fit the GPU suite's fuzz
slice) against the row loop it replaces: identical (indptr, cols, vals) for the same random draws."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_vectorised_stencil_rows_equal_the_row_loop():
    F = _load("fuzz_spmv")
    seen = set()
    for seed in range(40):
        n = 3000 + 517 * seed
        F.rng = np.random.default_rng(seed)
        a = F.make(n, kinds=(5, 6), vectorised=False)
        F.rng = np.random.default_rng(seed)
        b = F.make(n, kinds=(5, 6), vectorised=True)
        for u, v in zip(a, b):
            assert u.dtype == v.dtype and np.array_equal(u, v), seed
        seen.add((int(a[0][-1]) // n))
    assert len(seen) >= 3          # several row lengths were drawn
