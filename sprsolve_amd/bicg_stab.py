"""BiCGStab — mirror of the reference's src/bicg_stab.rs."""
from . import _lib
from ._solver import _SolverBase


class BiCGStab(_SolverBase):
    """`BiCGStab::new(&A, size)` (bicg_stab.rs:25); the recurrence runs in C++ on device-resident
    vectors and scalars (sprsolve_amd/csrc/krylov.hip)."""
    KIND = _lib.SOLVER_BICGSTAB
    NAME = "bicgstab"

    def solve(self, rhs, x, max_iter, tol):
        """bicg_stab.rs:35-200.  Returns (iters, relative residual); raises SolverError."""
        return self._solve(None, rhs, x, max_iter, tol, False)

    def precond_solve(self, precond, rhs, x, max_iter, tol):
        """bicg_stab.rs:204-366 (right-preconditioned)."""
        return self._solve(precond, rhs, x, max_iter, tol, True)
