// graphite_mi355x.hpp — host-side C++17 mirror of the reference's interface for the BAL hot path.
//
// Thin, header-only layer over the C-ABI (graphite_mi355x.h).  Class and method names, argument
// meaning and error behaviour follow the reference objects a BAL program touches
// (file:line relative to /root/reference):
//   graphite::optimizer::LevenbergMarquardtOptions / levenberg_marquardt
//                                            optimizer/levenberg_marquardt.hpp:52-98, 110-242
//   graphite::Solver<T,S> (4 methods)        solver/solver.hpp:12-25
//   graphite::PCGSolver / PCGSchurSolver     solver/pcg.hpp:35-40, solver/pcg_schur.hpp:42-47
//   graphite::BlockJacobiPreconditioner, IdentityPreconditioner, BlockJacobiSchurPreconditioner
//   graphite::Graph<T,S> methods used by the optimiser   graph.hpp:92-318
//   graphite::StreamPool                     stream.hpp:7-24 (kept for source compatibility)
// The reference's generic trait-templated descriptors (user-defined vertices/factors) are NOT
// provided here: the graph type is BalGraph<T>, whose constructor takes the arrays that
// examples/bal.cu:55-141 feeds into CameraDescriptor / PointDescriptor / ReprojectionError.
// Like the reference: `bool` returns, diagnostics on std::cerr, non-owning raw pointers between
// options -> solver -> preconditioner, no exceptions except for invalid construction.
#pragma once
#include "graphite_mi355x.h"
#include <cstdint>
#include <iomanip>
#include <iostream>
#include <stdexcept>
#include <vector>

namespace graphite {

// stream.hpp:7-24 — the MI355X engine issues everything on one stream; the pool is accepted and ignored.
class StreamPool {
public:
  explicit StreamPool(size_t = 1) {}
  void sync_all() {}
  void sync_n(size_t) {}
};

template <typename T, int E> struct DefaultLoss {};                       // loss.hpp:15-30
template <typename T, int E> struct HuberLoss { T delta = 100; HuberLoss() = default; explicit HuberLoss(T d) : delta(d) {} };

template <typename T> constexpr gr_dtype dtype_of() {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "float or double");
  return sizeof(T) == 8 ? GR_F64 : GR_F32;
}

// The device-resident BAL graph: Graph<T,S> + CameraDescriptor + PointDescriptor + ReprojectionError.
template <typename T, typename S = T> class BalGraph {
  static_assert(sizeof(T) == sizeof(S), "mixed precision graphs are not provided (DESIGN.md §6)");
  gr_bal_problem *p_ = nullptr;
  size_t nc_, np_, no_;
public:
  // cameras: 9 per camera [r t f k1 k2], points: 3 per point, observations: 2 per factor;
  // factor i connects camera cam_idx[i] and point pt_idx[i] (bal.cu:96-109: add_factor({cam, pt + Nc}, obs)).
  BalGraph(const std::vector<T> &cameras, const std::vector<T> &points, const std::vector<T> &observations,
           const std::vector<int32_t> &cam_idx, const std::vector<int32_t> &pt_idx, int device = 0, void *stream = nullptr)
      : nc_(cameras.size() / 9), np_(points.size() / 3), no_(cam_idx.size()) {
    const gr_status st = gr_bal_create(&p_, dtype_of<T>(), (int64_t)nc_, (int64_t)np_, (int64_t)no_, cameras.data(), points.data(),
                                       observations.data(), cam_idx.data(), pt_idx.data(), device, stream);
    if (st != GR_OK) throw std::runtime_error(std::string("BalGraph: ") + gr_last_error_string());
  }
  BalGraph(const BalGraph &) = delete;
  BalGraph &operator=(const BalGraph &) = delete;
  ~BalGraph() { gr_bal_destroy(p_); }
  gr_bal_problem *handle() const { return p_; }
  bool ok(gr_status st) const { if (st != GR_OK) std::cerr << "graphite-mi355x: " << gr_last_error_string() << std::endl; return st == GR_OK; }

  size_t get_hessian_dimension() const { return 9 * nc_ + 3 * np_; }          // graph.hpp:47
  size_t get_num_block_columns() const { return nc_ + np_; }                   // graph.hpp:51
  size_t get_elimination_block_column() const { return nc_; }                  // graph.hpp:90 (points are eliminated)
  size_t num_cameras() const { return nc_; }
  size_t num_points() const { return np_; }
  size_t num_observations() const { return no_; }

  bool initialize_optimization(uint8_t /*level*/ = 0) { return true; }          // done at construction
  bool build_structure() { return true; }
  void scale_system(bool enable) { ok(gr_bal_set_scale_system(p_, enable)); }   // graph.hpp:331
  // VertexDescriptor::set_fixed (vertex.hpp:262-264) for whole descriptors at once: byte masks over cameras / points
  // (nullptr = none of that kind); a fixed vertex keeps its value through the optimisation
  void set_fixed(const unsigned char *camera_fixed, const unsigned char *point_fixed) { ok(gr_bal_set_fixed(p_, camera_fixed, point_fixed)); }
  // Graph<double, float>: Jacobian entries in fp32 (bal.cu --precision FP64-FP32)
  void set_jacobian_precision_f32(bool on) { ok(gr_bal_set_jacobian_precision(p_, on ? GR_F32 : GR_F64)); }
  template <int E> void set_loss(const DefaultLoss<T, E> &) { ok(gr_bal_set_loss(p_, GR_LOSS_DEFAULT, 0.0)); }
  template <int E> void set_loss(const HuberLoss<T, E> &l) { ok(gr_bal_set_loss(p_, GR_LOSS_HUBER, (double)l.delta)); }

  void linearize(StreamPool &) { ok(gr_bal_linearize(p_)); }                    // graph.hpp:236
  void linearize() { ok(gr_bal_linearize(p_)); }
  void compute_error() {}                                                       // folded into chi2()
  T chi2() { double c = 0; ok(gr_bal_chi2(p_, &c)); return (T)c; }              // graph.hpp:212-225
  void backup_parameters() { ok(gr_bal_backup_parameters(p_)); }                // graph.hpp:302
  void revert_parameters() { ok(gr_bal_revert_parameters(p_)); }                // graph.hpp:311
  void apply_update(const T *delta_x, StreamPool &) { ok(gr_bal_apply_update(p_, delta_x)); }  // graph.hpp:292
  void apply_update(const T *delta_x) { ok(gr_bal_apply_update(p_, delta_x)); }

  std::vector<T> get_b() { return get(GR_GET_B); }                              // graph.hpp:55
  std::vector<T> get_jacobian_scales() { return get(GR_GET_SCALES); }           // graph.hpp:66
  std::vector<T> get(gr_bal_array which) {
    int64_t n = 0;
    ok(gr_bal_get(p_, which, nullptr, &n));
    std::vector<T> v((size_t)n);
    ok(gr_bal_get(p_, which, v.data(), &n));
    return v;
  }
  // the reference updates user-owned vertices in place (vertex.hpp:65); here they are read back
  void read_back(std::vector<T> &cameras, std::vector<T> &points) {
    cameras.resize(9 * nc_); points.resize(3 * np_);
    ok(gr_bal_get_params(p_, cameras.data(), points.data()));
  }
  void set_vertices(const std::vector<T> &cameras, const std::vector<T> &points) { ok(gr_bal_set_params(p_, cameras.data(), points.data())); }
};

// ---- preconditioners (tags carried to the solver; the work happens inside the library) ----------
template <typename T, typename S = T> struct Preconditioner { virtual ~Preconditioner() = default; virtual bool identity() const = 0; };
template <typename T, typename S = T> struct IdentityPreconditioner : Preconditioner<T, S> { bool identity() const override { return true; } };
template <typename T, typename S = T> struct BlockJacobiPreconditioner : Preconditioner<T, S> { bool identity() const override { return false; } };
template <typename T, typename S = T> struct SchurPreconditioner { virtual ~SchurPreconditioner() = default; };
template <typename T, typename S = T> struct BlockJacobiSchurPreconditioner : SchurPreconditioner<T, S> {};

// ---- Solver<T,S> (solver/solver.hpp:12-25) -----------------------------------------------------
template <typename T, typename S = T> class Solver {
public:
  virtual ~Solver() = default;
  virtual void set_damping_factor(BalGraph<T, S> *graph, T damping_factor, const bool use_identity, StreamPool &streams) = 0;
  virtual void update_structure(BalGraph<T, S> *graph, StreamPool &streams) = 0;
  virtual void update_values(BalGraph<T, S> *graph, StreamPool &streams) = 0;
  virtual bool solve(BalGraph<T, S> *graph, T *delta_x, StreamPool &streams) = 0;
  virtual gr_solver kind() const = 0;
  virtual void pcg_parameters(int &max_iter, double &tol, double &rejection_ratio) const = 0;
};

template <typename T, typename S> class CApiSolver : public Solver<T, S> {
protected:
  gr_solver kind_;
  size_t max_iter_;
  T tol_, rejection_ratio_;
  int last_iterations_ = 0;
public:
  CApiSolver(gr_solver kind, size_t max_iter, T tol, T rejection_ratio)
      : kind_(kind), max_iter_(max_iter), tol_(tol), rejection_ratio_(rejection_ratio) {}
  void set_damping_factor(BalGraph<T, S> *g, T mu, const bool use_identity, StreamPool &) override { g->ok(gr_bal_solver_set_damping(g->handle(), kind_, (double)mu, use_identity)); }
  void update_structure(BalGraph<T, S> *g, StreamPool &) override { g->ok(gr_bal_solver_update_structure(g->handle(), kind_)); }
  void update_values(BalGraph<T, S> *g, StreamPool &) override { g->ok(gr_bal_solver_update_values(g->handle(), kind_)); }
  bool solve(BalGraph<T, S> *g, T *delta_x, StreamPool &) override {
    return g->ok(gr_bal_solver_solve(g->handle(), kind_, (int)max_iter_, (double)tol_, (double)rejection_ratio_, delta_x, &last_iterations_));
  }
  int last_iterations() const { return last_iterations_; }
  gr_solver kind() const override { return kind_; }
  void pcg_parameters(int &m, double &t, double &r) const override { m = (int)max_iter_; t = (double)tol_; r = (double)rejection_ratio_; }
};

// PCGSolver(max_iter, tol, rejection_ratio, Preconditioner*)   solver/pcg.hpp:35-40
template <typename T, typename S = T> class PCGSolver : public CApiSolver<T, S> {
public:
  PCGSolver(size_t max_iter, T tol, T rejection_ratio, Preconditioner<T, S> *preconditioner)
      : CApiSolver<T, S>(preconditioner && preconditioner->identity() ? GR_SOLVER_PCG_IDENTITY : GR_SOLVER_PCG, max_iter, tol, rejection_ratio) {
    if (!preconditioner) throw std::invalid_argument("PCGSolver: preconditioner is null");
  }
};
// PCGSchurSolver(max_iter, tol, rejection_ratio, SchurPreconditioner*)   solver/pcg_schur.hpp:42-47
template <typename T, typename S = T> class PCGSchurSolver : public CApiSolver<T, S> {
public:
  PCGSchurSolver(size_t max_iter, T tol, T rejection_ratio, SchurPreconditioner<T, S> *preconditioner)
      : CApiSolver<T, S>(GR_SOLVER_PCG_SCHUR, max_iter, tol, rejection_ratio) {
    if (!preconditioner) throw std::invalid_argument("PCGSchurSolver: preconditioner is null");
  }
};

// PCGSchurSolver with S applied implicitly (Hpp - Hpl Hll^-1 Hpl^T, never formed): same iterates, no
// Pi-sized product list; the choice for Venice/Final-sized graphs (mode 2b)
template <typename T, typename S = T> class PCGImplicitSchurSolver : public CApiSolver<T, S> {
public:
  PCGImplicitSchurSolver(size_t max_iter, T tol, T rejection_ratio, SchurPreconditioner<T, S> *preconditioner)
      : CApiSolver<T, S>(GR_SOLVER_PCG_SCHUR_IMPLICIT, max_iter, tol, rejection_ratio) {
    if (!preconditioner) throw std::invalid_argument("PCGImplicitSchurSolver: preconditioner is null");
  }
};
// Direct solve of the reduced camera system: EigenSchurLDLTSolver (solver/eigen_schur.hpp:16-110) and
// cudssSchurSolver (solver/cudss_schur.hpp:29-236) both map to the dense MFMA Cholesky of S.
template <typename T, typename S = T> class EigenSchurLDLTSolver : public CApiSolver<T, S> {
public:
  EigenSchurLDLTSolver() : CApiSolver<T, S>(GR_SOLVER_DENSE_SCHUR, 0, T(0), T(0)) {}
};
struct cudssSolverOptions { int64_t hybrid_memory = 0; }; // accepted, unused (solver/cudss.hpp:19-27)
// EigenLDLTSolver (solver/eigen.hpp:49-98) / cudssSolver (solver/cudss.hpp:29-262) factorise the full damped H.  On a
// two-type bundle-adjustment graph that is the same linear step as eliminating the points and factorising S (block
// Gaussian elimination), so both map to the engine's direct Schur solve.
template <typename T, typename S = T> class EigenLDLTSolver : public CApiSolver<T, S> {
public:
  EigenLDLTSolver() : CApiSolver<T, S>(GR_SOLVER_DENSE_SCHUR, 0, T(0), T(0)) {}
};
template <typename T, typename S = T> class cudssSolver : public CApiSolver<T, S> {
public:
  explicit cudssSolver(const cudssSolverOptions & = {}) : CApiSolver<T, S>(GR_SOLVER_DENSE_SCHUR, 0, T(0), T(0)) {}
};
template <typename T, typename S = T> class cudssSchurSolver : public CApiSolver<T, S> {
public:
  explicit cudssSchurSolver(const cudssSolverOptions & = {}) : CApiSolver<T, S>(GR_SOLVER_DENSE_SCHUR, 0, T(0), T(0)) {}
};

namespace optimizer {

// optimizer/levenberg_marquardt.hpp:52-98
template <typename T, typename S = T> class LevenbergMarquardtOptions {
public:
  LevenbergMarquardtOptions() = default;
  Solver<T, S> *solver = nullptr;
  size_t iterations = 10;
  double initial_damping = 1e-4;
  uint8_t optimization_level = 0;
  bool verbose = false;
  bool *stop_flag = nullptr;
  bool use_identity = false;
  StreamPool *streams = nullptr;
  bool validate() const {
    if (solver == nullptr) { if (verbose) std::cerr << "Levenberg-Marquardt options invalid: solver is null" << std::endl; return false; }
    if (streams == nullptr) { if (verbose) std::cerr << "Levenberg-Marquardt options invalid: streams is null" << std::endl; return false; }
    return true;
  }
};

// optimizer/levenberg_marquardt.hpp:110-242.  The loop runs inside the library (one call); with
// options->verbose the reference's iteration table (:153-163, :216-221) is printed afterwards from
// the recorded traces (per-iteration wall times are not recorded: the Time column shows the mean).
// stop_flag is handed to the library, which polls it after every iteration as the reference does (:233-238).
namespace detail {
template <typename T, typename S>
bool lm_call(BalGraph<T, S> *graph, LevenbergMarquardtOptions<T, S> *options, gr_lm_stats *stats_out, bool early_stop) {
  if (!options->validate()) {
    if (options->verbose) std::cerr << "Levenberg-Marquardt options invalid" << std::endl;
    return false;
  }
  gr_lm_options o{};
  o.solver = options->solver->kind();
  o.iterations = (int32_t)options->iterations;
  o.initial_damping = options->initial_damping;
  o.use_identity = options->use_identity;
  o.early_stop = early_stop ? 1 : 0;
  o.stop_flag = reinterpret_cast<const volatile unsigned char *>(options->stop_flag);
  int m; double t, r;
  options->solver->pcg_parameters(m, t, r);
  o.pcg_max_iter = m; o.pcg_tol = t; o.pcg_rejection_ratio = r;
  gr_lm_stats st{};
  std::vector<double> chi2(options->iterations + 1), lambda(options->iterations + 1);
  if (!graph->ok(gr_bal_levenberg_marquardt(graph->handle(), &o, &st, chi2.data(), lambda.data()))) return false;
  if (options->verbose) {
    std::cout << std::setprecision(12) << std::setw(18) << "Iteration" << std::setw(24) << "Initial Chi2" << std::setw(24)
              << "Current Chi2" << std::setw(24) << "Lambda" << std::setw(24) << "Time" << std::setw(24) << "Total Time" << std::endl;
    std::cout << std::string(138, '-') << std::endl;
    const double per_it = st.iterations_run ? st.loop_seconds / st.iterations_run : 0.0;
    const int prec = early_stop ? 4 : 12, w0 = early_stop ? 10 : 18, w = early_stop ? 16 : 24; // the rows of :216-221 / :382-387
    for (int i = 0; i < st.iterations_run; ++i)
      std::cout << std::setprecision(prec) << std::setw(w0) << i << std::setw(w) << chi2[i] << std::setw(w) << chi2[i + 1]
                << std::setw(w) << lambda[i + 1] << std::setw(w) << per_it << std::setw(w) << st.setup_seconds + per_it * (i + 1) << std::endl;
  }
  if (stats_out) *stats_out = st;
  return st.ok != 0;
}
} // namespace detail

template <typename T, typename S>
bool levenberg_marquardt(BalGraph<T, S> *graph, LevenbergMarquardtOptions<T, S> *options, gr_lm_stats *stats_out = nullptr) {
  return detail::lm_call(graph, options, stats_out, false);
}
// optimizer/levenberg_marquardt.hpp:255-418: the same iteration, leaving early once three consecutive accepted
// steps each lowered chi2 by less than 0.1 % (:404-414)
template <typename T, typename S>
bool levenberg_marquardt2(BalGraph<T, S> *graph, LevenbergMarquardtOptions<T, S> *options, gr_lm_stats *stats_out = nullptr) {
  return detail::lm_call(graph, options, stats_out, true);
}

} // namespace optimizer
} // namespace graphite
