"""simplexmethod_amd — MI355X-native dense-LP hot path (vertex enumeration + simplex pivot).

The product is the C-ABI library built from csrc/ (include/simplexmethod_amd.h) and the
C++ host classes in host/ that mirror the reference's Symmetrical / Canonical / Solver /
EnumerationSolver.  This Python package is plumbing for tests and bench.py: a ctypes
binding (capi) and the one-process-per-GPU enumeration driver (dist).
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
