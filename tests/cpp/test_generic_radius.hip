// Generic-layer plumbing test on the smallest useful graph (BASELINE configs[0]: unary factors on 2-d vertices, the
// problem of the reference's examples/circle.cu — which itself is compiled UNMODIFIED as build/ref_examples/circle):
// n points, one factor each pulling the point onto the circle |p| = R, one vertex fixed, one factor switched off.
// One binary covers what the reference exercises through separate builds:
//   test_generic_radius <n> <manual|auto> <lm|lm2> [pcg|eigen] [start-points file]
//     manual / auto : Traits::jacobian  vs  dual-number differentiation of Traits::error   (differentiation.hpp)
//     lm / lm2      : levenberg_marquardt vs the early-termination variant                 (levenberg_marquardt.hpp:100-418)
//     pcg / eigen   : PCGSolver + IdentityPreconditioner as examples/circle.cu:139-140 configures it  vs  EigenLDLTSolver
//                     (solver/eigen.hpp:49-98) — the "eigen_solver path" BASELINE configs[0] names
//     file          : n lines "x y" (17 significant digits) instead of the built-in start, so that the oracle
//                     (oracle/circle_fit.hpp) starts from the same bits (tests/test_generic_api.py writes graphite_amd.synth.make_circle)
// Start points are a fixed function of the index (no random device), so every run prints the same table.
#include <cmath>
#include <graphite/optimizer/levenberg_marquardt.hpp>
#include <graphite/preconditioner/identity.hpp>
#include <graphite/solver/pcg.hpp>
#include <graphite/solver/eigen.hpp>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

namespace graphite {

template <typename T> struct Xy { T x, y; };

template <typename T> struct XyTraits {
  static constexpr size_t dimension = 2;
  using Vertex = Xy<T>;
  template <typename P> d_fn static void parameters(const Vertex &v, P *out) { out[0] = P(v.x); out[1] = P(v.y); }
  d_fn static void update(Vertex &v, const T *step) { v.x += step[0]; v.y += step[1]; }
};
template <typename T, typename S> using XyDescriptor = VertexDescriptor<T, S, XyTraits<T>>;

// residual |p|^2 - R^2 with R as the observation; Mode picks who differentiates it
template <typename T, typename S, typename Mode> struct OnCircleTraits {
  static constexpr size_t dimension = 1;
  using VertexDescriptors = std::tuple<XyDescriptor<T, S>>;
  using Observation = T;
  using Data = Empty;
  using Loss = DefaultLoss<T, 1>;
  using Differentiation = Mode;
  template <typename D> d_fn static void error(const D *p, const T &R, D *e) { e[0] = p[0] * p[0] + p[1] * p[1] - D(R * R); }
  template <typename J, size_t I> d_fn static void jacobian(const Xy<T> &p, const T &, J *jac) { jac[0] = J(2 * p.x); jac[1] = J(2 * p.y); }
};

template <typename Mode> static int fit(size_t n, bool early_stop, bool eigen, const char *start_file) {
  using T = double;
  using Factor = FactorDescriptor<T, T, OnCircleTraits<T, T, Mode>>;
  const T R = 4.0;
  const size_t first_id = 10; // vertex ids are the user's, not positions
  managed_vector<Xy<T>> pts(n);
  std::vector<Xy<T>> start(n);
  for (size_t i = 0; i < n; ++i) {
    // one point per quadrant in turn, within 0.3 rad of the diagonal: Marquardt's diag(H) damping divides by 4 x^2 and 4 y^2,
    // so a start on an axis sends that point off tangentially (the reference's random starts hit that now and then);
    // |p| in [R - 0.45, R + 0.45]
    const T ang = 0.25 * M_PI + 0.5 * M_PI * (T)i + 0.3 * std::sin(2.1 * (T)i + 0.4), rad = R + 0.45 * std::sin(1.3 * (T)i + 0.2);
    pts[i] = start[i] = Xy<T>{rad * std::cos(ang), rad * std::sin(ang)};
  }
  if (start_file) {
    std::ifstream in(start_file);
    for (size_t i = 0; i < n; ++i) {
      T x, y;
      if (!(in >> x >> y)) { std::cerr << "cannot read " << n << " points from " << start_file << std::endl; return 2; }
      pts[i] = start[i] = Xy<T>{x, y};
    }
  }
  Graph<T, T> graph;
  XyDescriptor<T, T> vertices;
  vertices.reserve(n);
  graph.add_descriptor(&vertices);
  for (size_t i = 0; i < n; ++i) vertices.add_vertex(first_id + i, &pts[i]);
  Factor factors(&vertices);
  factors.reserve(n);
  graph.add_descriptor(&factors);
  const DefaultLoss<T, 1> loss;
  for (size_t i = 0; i < n; ++i) factors.add_factor({first_id + i}, R, nullptr, Empty(), loss);
  const size_t fixed = n - 1, muted = n > 2 ? 2 : 0;
  vertices.set_fixed(first_id + fixed, true); // vertex.hpp:262
  factors.set_active(muted, 0x1);             // factor.hpp:385: any non-zero bit outside the optimisation level's mask

  IdentityPreconditioner<T, T> precond;
  PCGSolver<T, T> pcg_solver(50, 1e-20, 10.0, &precond);
  EigenLDLTSolver<T, T> eigen_solver;
  StreamPool streams(1);
  optimizer::LevenbergMarquardtOptions<T, T> opt;
  opt.solver = eigen ? static_cast<Solver<T, T> *>(&eigen_solver) : static_cast<Solver<T, T> *>(&pcg_solver);
  opt.initial_damping = 1e-6;
  opt.iterations = 100;
  opt.optimization_level = 0;
  opt.verbose = true;
  opt.streams = &streams;
  std::cout << "FACTORS " << factors.internal_count() << " ENGINE_HANDOVERS_BEFORE " << optimizer::engine_handover_count() << std::endl;
  if (early_stop) optimizer::levenberg_marquardt2<T, T>(&graph, &opt);
  else optimizer::levenberg_marquardt<T, T>(&graph, &opt);

  int bad = 0;
  for (size_t i = 0; i < n; ++i) {
    const Xy<T> &p = *vertices.get_vertex(first_id + i);
    const T r = std::hypot(p.x, p.y);
    std::cout << std::setprecision(17) << "POINT " << i << " " << p.x << " " << p.y << " RADIUS " << r << std::endl;
    if (i == fixed || i == muted) bad += !(p.x == start[i].x && p.y == start[i].y); // untouched, bit for bit
    else bad += !(std::abs(r - R) < 1e-6);
  }
  std::cout << std::setprecision(17) << "FINAL_CHI2 " << graph.chi2() << std::endl;
  std::cout << (bad ? "FAILED" : "OK") << " (" << bad << " failures)" << std::endl;
  return bad != 0;
}

} // namespace graphite

int main(int argc, char **argv) {
  if (argc < 4) { std::cerr << "usage: test_generic_radius <n> <manual|auto> <lm|lm2> [pcg|eigen] [start-points file]" << std::endl; return 2; }
  (void)hipSetDevice(0);
  const size_t n = std::stoul(argv[1]);
  const bool early = std::string(argv[3]) == "lm2";
  const bool eigen = argc > 4 && std::string(argv[4]) == "eigen";
  const char *start_file = argc > 5 ? argv[5] : nullptr;
  if (std::string(argv[2]) == "auto") return graphite::fit<graphite::DifferentiationMode::Auto>(n, early, eigen, start_file);
  return graphite::fit<graphite::DifferentiationMode::Manual>(n, early, eigen, start_file);
}
