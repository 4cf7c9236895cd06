"""sprsolve_amd — MI355X-native (gfx950, HIP) backend for sprsolve's Krylov hot path.

Public surface mirrors the reference crate's (src/lib.rs:15-19): `BiCGStab`, `MinRes`,
`CSMinRes`, `GaussSeidel`, `MatVecMul`, `precond`, `vecalg`, `error`.  All compute runs in
libsprsolve_hip.so (hand-written HIP, C ABI in include/sprsolve_hip.h); importing the
package does not load it, using it does, and there is no CPU fallback.
"""
from . import error, precond, vecalg  # noqa: F401
from .bicg_stab import BiCGStab  # noqa: F401
from .cs_minres import CSMinRes  # noqa: F401
from .gauss_seidel import GaussSeidel  # noqa: F401
from .device import Context, DevVec, default_ctx  # noqa: F401
from .mat import HipCsr, MatVecMul  # noqa: F401
from .minres import MinRes  # noqa: F401
from .precond import DiagPrecond  # noqa: F401

__version__ = "0.1.0"
