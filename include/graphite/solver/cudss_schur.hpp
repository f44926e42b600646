// graphite/solver/cudss_schur.hpp (reference path): cudssSchurSolver<T,S> is the direct solve of the reduced camera
// system (solver/cudss_schur.hpp:146-234) = the role of EigenSchurLDLTSolver here (sparse Schur complement + the
// MFMA Cholesky of libgraphite_mi355x.so).
#pragma once
#include "cudss.hpp"
namespace graphite {
template <typename T, typename S> class cudssSchurSolver : public EigenSchurLDLTSolver<T, S> {
public:
  explicit cudssSchurSolver(const cudssSolverOptions & = {}) {}
};
} // namespace graphite
