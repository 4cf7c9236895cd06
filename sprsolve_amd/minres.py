"""MINRES — mirror of the reference's src/minres.rs."""
from . import _lib
from ._solver import _SolverBase


class MinRes(_SolverBase):
    """`MinRes::new(&A, size)` (minres.rs:21).  Real symmetric / complex Hermitian systems."""
    KIND = _lib.SOLVER_MINRES
    NAME = "minres"

    def solve(self, rhs, x, max_iter, tol):
        """minres.rs:31-172.  `iters` is 0-based, as upstream (minres.rs:166)."""
        return self._solve(None, rhs, x, max_iter, tol, False)

    def precond_solve(self, precond, rhs, x, max_iter, tol):
        """minres.rs:178-341."""
        return self._solve(precond, rhs, x, max_iter, tol, True)
