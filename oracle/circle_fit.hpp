// ORACLE — test infrastructure only (see bal_model.hpp header): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use anything under oracle/.
//
// BASELINE configs[0]: the reference's examples/circle.cu problem — n 2-d vertices, one UNARY factor each with the residual
// |p|^2 - R^2 (circle.cu:38-68: error and jacobian of CircleFactorTraits), the last vertex fixed (circle.cu:133), the factor
// of vertex 2 switched off (circle.cu:136) — optimised by optimizer::levenberg_marquardt (optimizer/levenberg_marquardt.hpp:
// 110-242) with the inner solver either
//   * EigenLDLTSolver (solver/eigen.hpp:49-98; src/eigen_solver.cpp:10-29: Eigen::SimplicialLDLT<Upper> of the damped H in
//     scalar CSC — restated by sparse_ldlt.hpp), or
//   * PCGSolver with the IdentityPreconditioner, as circle.cu:139-140 configures it (solver/pcg.hpp:61-232).
// Pipeline per linearisation as Graph::linearize (graph.hpp:236-290): error, Jacobian, chi2 / rho', column scales
// 1 / (eps + sqrt(diag(J^T rho' P J))) in double, J scaled in place, b = -J^T rho' P r; Hessian::update_values + apply_damping
// (hessian.hpp:136-176: d + mu clamp(d, 1e-6, 1e32) or d + mu); Graph::apply_update with the scales (graph.hpp:292-300).
// Ordering (graph.hpp:100-149): active vertices take columns in vertex order; a vertex is active iff it is not fixed and an
// active factor touches it (active.hpp:18-21) — vertex n - 1 (fixed) and vertex 2 (its only factor is off) have no column.
// Precision matrix: identity (add_factor(..., nullptr, ...): factor.hpp:373-412); loss: DefaultLoss.
//
// One deliberate difference, the same one include/graphite/core.hpp makes: chi2 is summed over the ACTIVE factors.  The
// reference's chi2 kernel indexes the first `active_count` factors instead of `active_indices` (ops/chi2.hpp:36-43), which
// on this graph counts the switched-off factor 2 and drops factor n - 1 ("reference quirks").
#pragma once
#include "sparse_ldlt.hpp"
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace gro {

struct CircleOracle {
  size_t n = 0;
  double R = 0;
  std::vector<double> p, p_backup;       // [n][2]
  std::vector<uint8_t> fixed, factor_on; // per vertex / per factor
  bool scale_system = true;
  // linearisation
  std::vector<int64_t> col;              // first scalar column of vertex i, -1: none
  size_t dim = 0;
  std::vector<double> r, J, dchi2, scales, b, H, Hdiag0; // J: [n][2] (scaled); H: [n_active][4] column-major 2 x 2 blocks
  double damping = 0; bool damping_identity = false;
  int last_pcg_iters = 0;
  SparseLDLT ldlt; bool ldlt_ready = false;
  std::vector<int64_t> csc_p, csc_i; std::vector<double> csc_x;

  CircleOracle(size_t n_, double R_, const double *pts) : n(n_), R(R_), p(pts, pts + 2 * n_), fixed(n_, 0), factor_on(n_, 1) {}

  void initialize() { // Graph::initialize_optimization
    col.assign(n, -1);
    dim = 0;
    for (size_t i = 0; i < n; ++i)
      if (!fixed[i] && factor_on[i]) { col[i] = (int64_t)dim; dim += 2; }
    ldlt_ready = false;
  }
  void compute_error() { // ops/error.hpp:253-323 on circle.cu:47-52
    r.assign(n, 0.0);
    for (size_t f = 0; f < n; ++f) if (factor_on[f]) r[f] = p[2 * f] * p[2 * f] + p[2 * f + 1] * p[2 * f + 1] - R * R;
  }
  double chi2() { // ops/chi2.hpp:10-44 with DefaultLoss (loss.hpp:15-24): rho(x) = x, rho' = 1; factor.hpp:551-557
    dchi2.assign(n, 1.0);
    double s = 0;
    for (size_t f = 0; f < n; ++f) if (factor_on[f]) s += r[f] * r[f];
    return s;
  }
  void linearize() { // graph.hpp:236-290
    compute_error();
    J.assign(2 * n, 0.0);
    for (size_t f = 0; f < n; ++f)
      if (factor_on[f] && col[f] >= 0) { J[2 * f] = 2 * p[2 * f]; J[2 * f + 1] = 2 * p[2 * f + 1]; } // circle.cu:60-67; fixed vertices: skipped (ops/linearize.hpp:24)
    (void)chi2();
    scales.assign(dim, 1.0);
    if (scale_system) {
      for (size_t f = 0; f < n; ++f)
        if (factor_on[f] && col[f] >= 0)
          for (int k = 0; k < 2; ++k) scales[col[f] + k] = 1.0 / (std::numeric_limits<double>::epsilon() + std::sqrt(J[2 * f + k] * J[2 * f + k] * dchi2[f]));
      for (size_t f = 0; f < n; ++f)
        if (factor_on[f] && col[f] >= 0)
          for (int k = 0; k < 2; ++k) J[2 * f + k] *= scales[col[f] + k]; // ops/linearize.hpp:142-180
    }
    b.assign(dim, 0.0);
    for (size_t f = 0; f < n; ++f)
      if (factor_on[f] && col[f] >= 0)
        for (int k = 0; k < 2; ++k) b[col[f] + k] -= J[2 * f + k] * dchi2[f] * r[f]; // ops/linearize.hpp:240-303
  }
  void hessian_update_values() { // hessian.hpp:290-307, ops/hessian.hpp:10-78
    H.assign(2 * dim, 0.0);
    for (size_t f = 0; f < n; ++f)
      if (factor_on[f] && col[f] >= 0) {
        double *B = &H[2 * (size_t)col[f]];
        for (int c = 0; c < 2; ++c) for (int rw = 0; rw < 2; ++rw) B[rw + 2 * c] += dchi2[f] * J[2 * f + rw] * J[2 * f + c];
      }
    Hdiag0.assign(dim, 0.0);
    for (size_t v = 0; v * 2 < dim; ++v) { Hdiag0[2 * v] = H[4 * v]; Hdiag0[2 * v + 1] = H[4 * v + 3]; } // backup_diagonal, hessian.hpp:102
  }
  static double clamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
  void apply_damping(double mu, bool identity) { // hessian.hpp:136-176
    damping = mu; damping_identity = identity;
    for (size_t v = 0; v * 2 < dim; ++v)
      for (int k = 0; k < 2; ++k) {
        const double d = Hdiag0[2 * v + k];
        H[4 * v + 3 * k] = identity ? d + mu : d + mu * clamp(d, 1.0e-6, 1.0e32);
      }
  }
  bool solve_ldlt(double *x) { // solver/eigen.hpp:76-98 on csc_utils.hpp:74-193 (upper triangle, scalar CSC)
    csc_p.assign(dim + 1, 0); csc_i.clear(); csc_x.clear();
    for (size_t v = 0; v * 2 < dim; ++v) {
      csc_i.push_back(2 * v); csc_x.push_back(H[4 * v]);
      csc_p[2 * v + 1] = (int64_t)csc_i.size();
      csc_i.push_back(2 * v); csc_x.push_back(H[4 * v + 2]);
      csc_i.push_back(2 * v + 1); csc_x.push_back(H[4 * v + 3]);
      csc_p[2 * v + 2] = (int64_t)csc_i.size();
    }
    if (!ldlt_ready) { ldlt.analyze((int64_t)dim, csc_p.data(), csc_i.data(), {}); ldlt_ready = true; }
    if (!ldlt.factorize(csc_x.data())) return false;
    return ldlt.solve(b.data(), x);
  }
  // y = (J^T rho' J + mu D) v, the products of ops/product.hpp:195,405 with the scaled Jacobians
  void apply_operator(const std::vector<double> &v, const std::vector<double> &diag, std::vector<double> &y) const {
    y.assign(dim, 0.0);
    for (size_t f = 0; f < n; ++f)
      if (factor_on[f] && col[f] >= 0) {
        const double u = (J[2 * f] * v[col[f]] + J[2 * f + 1] * v[col[f] + 1]) * dchi2[f];
        y[col[f]] += J[2 * f] * u; y[col[f] + 1] += J[2 * f + 1] * u;
      }
    for (size_t k = 0; k < dim; ++k) y[k] += damping * (damping_identity ? 1.0 : diag[k]) * v[k];
  }
  bool solve_pcg(double *x, int max_iter, double tol, double rejection_ratio) { // solver/pcg.hpp:61-232, IdentityPreconditioner
    last_pcg_iters = 0;
    std::vector<double> xs(dim, 0.0), rr(b), diag(dim), y(dim), z(dim), pd(dim), v2, xb(dim);
    for (size_t k = 0; k < dim; ++k) diag[k] = clamp(Hdiag0[k], 1.0e-6, 1.0e32); // :88-103
    auto dot = [&](const std::vector<double> &a, const std::vector<double> &c) { double s = 0; for (size_t k = 0; k < dim; ++k) s += a[k] * c[k]; return s; };
    double rnorm = std::sqrt(dot(rr, rr)), scale = 1.0 / rnorm;
    for (size_t k = 0; k < dim; ++k) z[k] = scale * rr[k]; // y = r / |r|, z = M^-1 y = y
    pd = z;
    double rz = dot(rr, z), rz_0 = std::numeric_limits<double>::infinity();
    for (int k = 0; k < max_iter; ++k) {
      if (rz == 0) break;
      apply_operator(pd, diag, v2);
      const double alpha = rz / dot(pd, v2);
      xb = xs;
      for (size_t i = 0; i < dim; ++i) { xs[i] += alpha * pd[i]; rr[i] -= alpha * v2[i]; }
      rnorm = std::sqrt(dot(rr, rr)); scale = 1.0 / rnorm;
      for (size_t i = 0; i < dim; ++i) z[i] = scale * rr[i];
      const double rz_new = dot(rr, z);
      last_pcg_iters = k + 1;
      if (std::abs(rz_new) > rejection_ratio * rz_0 || std::isnan(rz_new)) { xs = xb; break; }
      rz_0 = std::min(rz_0, std::abs(rz_new));
      const double beta = rz_new / rz;
      rz = rz_new;
      for (size_t i = 0; i < dim; ++i) pd[i] = z[i] + beta * pd[i];
      if (std::abs(rz_new) < tol) break;
    }
    for (size_t k = 0; k < dim; ++k) x[k] = xs[k];
    return true;
  }
  void backup() { p_backup = p; }
  void revert() { p = p_backup; }
  void apply_update(const double *dx) { // graph.hpp:292-300, ops/update.hpp:11-31 on circle.cu's Point update (plain addition)
    for (size_t i = 0; i < n; ++i) if (col[i] >= 0) for (int k = 0; k < 2; ++k) p[2 * i + k] += dx[col[i] + k] * scales[col[i] + k];
  }
  // optimizer/levenberg_marquardt.hpp:110-242; solver 0: EigenLDLTSolver, 1: PCGSolver + IdentityPreconditioner.
  // Returns the number of iterations run; trace[0 .. it]: chi2 after each, lambda likewise.
  int levenberg_marquardt(int solver, int iterations, double initial_damping, bool use_identity, int pcg_max_iter, double pcg_tol, double pcg_rej,
                          double *chi2_trace, double *lambda_trace, int *accepted, int *pcg_iterations) {
    initialize();
    double mu = initial_damping, nu = 2;
    linearize();
    hessian_update_values();
    double chi2v = chi2();
    std::vector<double> dx(dim, 0.0);
    chi2_trace[0] = chi2v; lambda_trace[0] = mu;
    int it = 0;
    *accepted = 0; *pcg_iterations = 0;
    bool run = true;
    for (int i = 0; i < iterations && run; ++i) {
      apply_damping(mu, use_identity);
      const bool solve_ok = solver == 0 ? solve_ldlt(dx.data()) : solve_pcg(dx.data(), pcg_max_iter, pcg_tol, pcg_rej);
      *pcg_iterations += solver == 0 ? 0 : last_pcg_iters;
      backup();
      apply_update(dx.data());
      compute_error();
      double new_chi2 = chi2();
      if (!solve_ok) new_chi2 = std::numeric_limits<double>::max();
      double denom = 1.0;
      if (solve_ok) { denom = 0; for (size_t k = 0; k < dim; ++k) denom += dx[k] * (mu * dx[k] + b[k]); denom += 1.0e-3; } // compute_rho :20-47
      const double rho = (chi2v - new_chi2) / denom;
      if (solve_ok && std::isfinite(new_chi2) && rho > 0) {
        double alpha = 1.0 - std::pow(2.0 * rho - 1.0, 3);
        alpha = std::max(std::min(alpha, 2.0 / 3.0), 1.0 / 3.0);
        mu *= alpha; nu = 2;
        linearize();
        hessian_update_values();
        ++*accepted;
      } else {
        revert();
        compute_error(); (void)chi2();
        mu *= nu; nu *= 2;
        new_chi2 = chi2v;
      }
      chi2v = new_chi2;
      ++it;
      chi2_trace[it] = chi2v; lambda_trace[it] = mu;
      if (!std::isfinite(mu)) run = false;
      if (rho == 0) break;
    }
    return it;
  }
};

} // namespace gro
