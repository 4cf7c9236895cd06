"""SolverError / SolveResult of the reference (src/error.rs:3-22), 1:1 with the C status codes."""
from . import _lib


class SolverError(Exception):
    """Base of the reference's `SolverError` enum (src/error.rs:7)."""
    status = None


class IncompatibleMatrixFormat(SolverError):
    """src/error.rs:9-10 — message strings of src/bicg_stab.rs:44-53."""

    def __init__(self, msg):
        super().__init__("Incompatible input matrix format: %s" % msg)
        self.msg = msg


class ZeorDiagonalElem(SolverError):
    """src/error.rs:12-13 (raised by GaussSeidel.solve, gauss_seidel.rs:72-78; spelling as upstream)."""

    def __init__(self, row):
        super().__init__("Matrix has zero diagonal element at %d" % row)
        self.row = row


class InsufficientIterNum(SolverError):
    """src/error.rs:15-16."""

    def __init__(self, n):
        super().__init__("Insufficient interation #: %d" % n)
        self.iters = n


class BreakDown(SolverError):
    """src/error.rs:18-19."""

    def __init__(self, its):
        super().__init__("Solver break down: its #%d" % its)
        self.its = its


class InvalidPreconditioner(SolverError):
    """src/error.rs:21-22."""

    def __init__(self, msg):
        super().__init__("Invalid preconditioner: %s" % msg)
        self.msg = msg


class DimensionMismatch(Exception):
    """The reference's `panic!("Dimension mismatch")` (src/mat.rs:50-52, src/precond.rs:39-41)."""


class BackendError(RuntimeError):
    """HIP / RCCL failure inside the library (no reference analogue)."""


def check(status, ctx=None):
    """Generic status -> exception for non-solver entry points."""
    if status == _lib.OK:
        return
    if status == _lib.DIM_MISMATCH:
        raise DimensionMismatch("Dimension mismatch")
    if status == _lib.INVALID_ARGUMENT:
        raise ValueError("sprsolve_hip: invalid argument")
    if status == _lib.NOT_SQUARE:
        raise IncompatibleMatrixFormat("Not a square matrix")
    if status == _lib.NOT_CSR:
        raise IncompatibleMatrixFormat("Not in CSR format")
    detail = ""
    if ctx is not None and status >= _lib.ERR_HIP:
        detail = ": " + (_lib.lib().sprs_last_error(ctx) or b"").decode(errors="replace")
    raise BackendError("sprsolve_hip status %d (%s)%s" % (
        status, _lib.lib().sprs_status_str(status).decode(), detail))


def solve_result(status, its, res, ctx=None):
    """Map a solver status to the reference's `SolveResult<(usize, T::Real)>`."""
    if status == _lib.OK:
        return its, res
    if status == _lib.INCOMPATIBLE_RHS_SIZE:
        raise IncompatibleMatrixFormat("Input vec dimension doesn't match the matrix size")
    if status == _lib.INCOMPATIBLE_X_SIZE:
        raise IncompatibleMatrixFormat("Input and output vec dimension do not match")
    if status == _lib.INSUFFICIENT_ITER:
        raise InsufficientIterNum(its)
    if status == _lib.BREAKDOWN:
        raise BreakDown(its)
    if status == _lib.INVALID_PRECOND:
        raise InvalidPreconditioner("beta_%d [%r] is not positive" % (its, res))
    if status == _lib.ZERO_DIAGONAL:
        raise ZeorDiagonalElem(its)
    check(status, ctx)
