"""CSMINRES — mirror of the reference's src/cs_minres.rs (complex-symmetric systems)."""
from . import _lib
from ._solver import _SolverBase


class CSMinRes(_SolverBase):
    """`CSMinRes::new(&A, size)` (cs_minres.rs:19); Saunders process."""
    KIND = _lib.SOLVER_CSMINRES
    NAME = "csminres"

    def solve(self, rhs, x, max_iter, tol):
        """cs_minres.rs:29-158."""
        return self._solve(None, rhs, x, max_iter, tol, False)
