// ORACLE / CPU baseline — test infrastructure only (see bal_model.hpp header).
//
// Simplicial up-looking sparse LDL^T of a symmetric matrix given by its UPPER
// triangle in scalar CSC — the algorithm family of
// Eigen::SimplicialLDLT<SparseMatrix, Upper> that the reference's
// EigenLDLTSolver / EigenSchurLDLTSolver call on ONE host thread
// (/root/reference/src/eigen_solver.cpp:10-29, solver/eigen.hpp:49-98).
// Eigen 3.4 is an un-vendored dependency (CMakeLists.txt:25) and absent here;
// its SimplicialLDLT is itself a restatement of T. Davis' LDL package
// (elimination tree + per-row reach, "Algorithm 849"), which is what is
// restated below.  Eigen orders with AMD by default; this restatement takes a
// caller-supplied permutation (the BAL drivers pass "points first, then
// cameras by reverse Cuthill-McKee", which is what minimum-degree orderings
// produce on bundle-adjustment Hessians).  Absolute solution values of the
// reference's Eigen solvers are "parity unpinned" (SURVEY §8c): its tests only
// assert full-vs-Schur agreement at 1e-8 (tests/schur.cu:285-288), which
// tests/test_oracle_solvers.py re-asserts on this code.
#pragma once
#include <cstdint>
#include <vector>

namespace gro {

struct SparseLDLT {
  int64_t n = 0;
  std::vector<int64_t> Lp, Li, parent, Lnz, perm, iperm;
  std::vector<double> Lx, D;
  // permuted upper-triangular pattern
  std::vector<int64_t> Cp, Ci, Cmap; // Cmap: index into the caller's value array
  bool ok = false;

  // Ap/Ai: upper-triangular CSC pattern (row <= col), perm: new->old (may be empty).
  void analyze(int64_t n_, const int64_t *Ap, const int64_t *Ai, const std::vector<int64_t> &perm_) {
    n = n_;
    perm = perm_;
    if (perm.empty()) { perm.resize(n); for (int64_t i = 0; i < n; ++i) perm[i] = i; }
    iperm.assign(n, 0);
    for (int64_t i = 0; i < n; ++i) iperm[perm[i]] = i;
    // C = upper(P A P^T): entry (i,j) of A (i<=j) goes to (min,max) of permuted ids.
    std::vector<int64_t> cnt(n + 1, 0);
    for (int64_t j = 0; j < n; ++j)
      for (int64_t p = Ap[j]; p < Ap[j + 1]; ++p) {
        const int64_t a = iperm[Ai[p]], b = iperm[j];
        cnt[(a > b ? a : b) + 1]++;
      }
    Cp.assign(n + 1, 0);
    for (int64_t j = 0; j < n; ++j) Cp[j + 1] = Cp[j] + cnt[j + 1];
    Ci.assign(Cp[n], 0);
    Cmap.assign(Cp[n], 0);
    std::vector<int64_t> w(Cp.begin(), Cp.end() - 1);
    for (int64_t j = 0; j < n; ++j)
      for (int64_t p = Ap[j]; p < Ap[j + 1]; ++p) {
        const int64_t a = iperm[Ai[p]], b = iperm[j];
        const int64_t col = a > b ? a : b, row = a > b ? b : a;
        const int64_t q = w[col]++;
        Ci[q] = row;
        Cmap[q] = p;
      }
    // symbolic: elimination tree and column counts
    parent.assign(n, -1);
    Lnz.assign(n, 0);
    std::vector<int64_t> flag(n, -1);
    for (int64_t k = 0; k < n; ++k) {
      flag[k] = k;
      for (int64_t p = Cp[k]; p < Cp[k + 1]; ++p) {
        int64_t i = Ci[p];
        if (i < k) {
          for (; flag[i] != k; i = parent[i]) {
            if (parent[i] == -1) parent[i] = k;
            Lnz[i]++;
            flag[i] = k;
          }
        }
      }
    }
    Lp.assign(n + 1, 0);
    for (int64_t k = 0; k < n; ++k) Lp[k + 1] = Lp[k] + Lnz[k];
    Li.assign(Lp[n], 0);
    Lx.assign(Lp[n], 0.0);
    D.assign(n, 0.0);
  }

  template <typename T> bool factorize(const T *Ax) {
    std::vector<double> Y(n, 0.0);
    std::vector<int64_t> pattern(n), flag(n, -1), lnz(n, 0);
    ok = true;
    for (int64_t k = 0; k < n; ++k) {
      int64_t top = n;
      flag[k] = k;
      Y[k] = 0.0;
      for (int64_t p = Cp[k]; p < Cp[k + 1]; ++p) {
        int64_t i = Ci[p];
        if (i <= k) {
          Y[i] += static_cast<double>(Ax[Cmap[p]]);
          int64_t len = 0;
          for (; flag[i] != k; i = parent[i]) {
            pattern[len++] = i;
            flag[i] = k;
          }
          while (len > 0) pattern[--top] = pattern[--len];
        }
      }
      D[k] = Y[k];
      Y[k] = 0.0;
      for (; top < n; ++top) {
        const int64_t i = pattern[top];
        const double yi = Y[i];
        Y[i] = 0.0;
        const int64_t p2 = Lp[i] + lnz[i];
        for (int64_t p = Lp[i]; p < p2; ++p) Y[Li[p]] -= Lx[p] * yi;
        const double l_ki = yi / D[i];
        D[k] -= l_ki * yi;
        Li[p2] = k;
        Lx[p2] = l_ki;
        lnz[i]++;
      }
      if (D[k] == 0.0) { ok = false; return false; }
    }
    return true;
  }

  template <typename T> bool solve(const T *b, T *x) const {
    if (!ok) return false;
    std::vector<double> y(n);
    for (int64_t i = 0; i < n; ++i) y[i] = static_cast<double>(b[perm[i]]);
    for (int64_t j = 0; j < n; ++j)
      for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) y[Li[p]] -= Lx[p] * y[j];
    for (int64_t j = 0; j < n; ++j) y[j] /= D[j];
    for (int64_t j = n - 1; j >= 0; --j)
      for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) y[j] -= Lx[p] * y[Li[p]];
    for (int64_t i = 0; i < n; ++i) x[perm[i]] = static_cast<T>(y[i]);
    return true;
  }
};

} // namespace gro
