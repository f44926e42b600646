// graphite/solver/cudss.hpp (reference path): cudssSolver<T,S> is the direct solve of the FULL damped system
// (solver/cudss.hpp:183-256).  cuDSS does not exist on ROCm; the same role is played by the direct solver of
// solve.hpp (EigenLDLTSolver: eliminated vertices first = Schur reduction + back-substitution, the reduced system on the
// MFMA tile Cholesky of libgraphite_mi355x.so; bundle-adjustment graphs go to the engine's nested-dissection sparse Cholesky).
#pragma once
#include "../solve.hpp"
namespace graphite {
struct cudssSolverOptions { int64_t hybrid_memory = 0; }; // solver/cudss.hpp:19-27; accepted, unused
template <typename T, typename S> class cudssSolver : public EigenLDLTSolver<T, S> {
public:
  explicit cudssSolver(const cudssSolverOptions & = {}) {}
};
} // namespace graphite
