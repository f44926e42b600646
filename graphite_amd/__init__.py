"""graphite_amd — MI355X-native hot path of sfu-rsl/graphite (BAL bundle adjustment).

The product is ``libgraphite_mi355x.so`` (HIP kernels + C-ABI, see
``include/graphite_mi355x.h``); this package is the thin Python harness used by
tests and ``bench.py``.  Nothing here imports ``oracle/``.
"""
from . import _lib, synth  # noqa: F401
from .bal import (BalProblem, dense_cholesky_solve, SOLVER_PCG, SOLVER_PCG_IDENTITY, SOLVER_PCG_SCHUR, SOLVER_PCG_SCHUR_IMPLICIT, SOLVER_DENSE_SCHUR,  # noqa: F401
                  LOSS_DEFAULT, LOSS_HUBER)

__all__ = ["BalProblem", "synth", "SOLVER_PCG", "SOLVER_PCG_IDENTITY", "SOLVER_PCG_SCHUR", "SOLVER_PCG_SCHUR_IMPLICIT", "SOLVER_DENSE_SCHUR", "dense_cholesky_solve",
           "LOSS_DEFAULT", "LOSS_HUBER"]
