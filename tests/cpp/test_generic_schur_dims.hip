// The compile-time forms of the generic Schur kernels (sparse.hpp: k_schur_mul_fixed, k_schur_hpl_*<D, DL>, k_schur_vec<D>) against
// the any-dimension kernels on a graph with 6-d "poses" and 3-d eliminated "landmarks" (the SE(3) + 3-d point shape; 9 / 3 and
// 3 / 2 are covered by the bundle-adjustment and planar-SLAM tests).  The factor is a smooth made-up function — only the
// dimensions matter.  tests/test_generic_api.py runs this binary twice, with and without GRAPHITE_SCHUR_MUL_GENERIC=1, and
// compares the printed numbers.
//   usage: test_generic_schur_dims <pcg-schur|eigen-schur>
#include <cmath>
#include <graphite/optimizer/levenberg_marquardt.hpp>
#include <graphite/preconditioner/block_jacobi_schur.hpp>
#include <graphite/solver/eigen_schur.hpp>
#include <graphite/solver/pcg_schur.hpp>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

namespace graphite {
template <int N> struct Arr {
  double v[N];
  hd_fn double operator()(int i) const { return v[i]; }
  hd_fn double &operator()(int i) { return v[i]; }
};
template <int N> struct ArrTraits {
  static constexpr size_t dimension = N;
  using Vertex = Arr<N>;
  template <typename P> d_fn static void parameters(const Vertex &x, P *p) { for (int i = 0; i < N; ++i) p[i] = P(x(i)); }
  d_fn static void update(Vertex &x, const double *d) { for (int i = 0; i < N; ++i) x(i) += d[i]; }
};
using Pose6 = VertexDescriptor<double, double, ArrTraits<6>>;
using Point3 = VertexDescriptor<double, double, ArrTraits<3>>;

// e = l + a x (b . l) t  -  z   with a = pose[0:3], t = pose[3:6]: bilinear in (pose, landmark), 3 residuals
struct SeesTraits {
  static constexpr size_t dimension = 3;
  using VertexDescriptors = std::tuple<Pose6, Point3>;
  using Observation = Arr<3>;
  using Data = Empty;
  using Loss = DefaultLoss<double, 3>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *p, const D *l, const Observation &z, D *e) {
    const D s = p[0] * l[0] + p[1] * l[1] + p[2] * l[2];
    for (int i = 0; i < 3; ++i) e[i] = l[i] + p[3 + i] * s + p[i] * D(0.1) - D(z(i));
  }
};
// a prior on every pose keeps the problem well posed
struct PriorTraits {
  static constexpr size_t dimension = 6;
  using VertexDescriptors = std::tuple<Pose6>;
  using Observation = Arr<6>;
  using Data = Empty;
  using Loss = DefaultLoss<double, 6>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *p, const Observation &z, D *e) { for (int i = 0; i < 6; ++i) e[i] = p[i] - D(z(i)); }
};
} // namespace graphite

int main(int argc, char **argv) {
  using namespace graphite;
  if (argc < 2) { std::cerr << "usage: test_generic_schur_dims <pcg-schur|eigen-schur>" << std::endl; return 2; }
  (void)hipSetDevice(0);
  const size_t np = 40, nl = 300;
  managed_vector<Arr<6>> poses(np);
  managed_vector<Arr<3>> lms(nl);
  auto f = [](double x) { return std::sin(12.9898 * x) * 0.5; }; // a fixed "noise"
  for (size_t i = 0; i < np; ++i) for (int k = 0; k < 6; ++k) poses[i](k) = 0.3 * std::cos(0.7 * i + k) + (k >= 3 ? 0.5 : 0.0);
  for (size_t j = 0; j < nl; ++j) for (int k = 0; k < 3; ++k) lms[j](k) = 2.0 * std::sin(0.37 * j + 1.3 * k) + 0.1 * k;
  Graph<double, double> graph;
  Pose6 pd; Point3 ld;
  pd.reserve(np); ld.reserve(nl);
  graph.add_descriptor(&pd); graph.add_descriptor(&ld);
  for (size_t i = 0; i < np; ++i) pd.add_vertex(i, &poses[i]);
  for (size_t j = 0; j < nl; ++j) ld.add_vertex(np + j, &lms[j]);
  ld.set_eliminate(true);
  FactorDescriptor<double, double, SeesTraits> sees(&pd, &ld);
  FactorDescriptor<double, double, PriorTraits> prior(&pd);
  graph.add_descriptor(&sees); graph.add_descriptor(&prior);
  const DefaultLoss<double, 3> l3; const DefaultLoss<double, 6> l6;
  for (size_t j = 0; j < nl; ++j)
    for (int q = 0; q < 5; ++q) { // every landmark is seen by five poses; some poses see many landmarks
      const size_t i = (7 * j + 11 * q * q + q) % np;
      bool dup = false;
      for (int q2 = 0; q2 < q; ++q2) dup = dup || (7 * j + 11 * q2 * q2 + q2) % np == i;
      if (dup) continue;
      Arr<3> z;
      for (int k = 0; k < 3; ++k) z(k) = lms[j](k) + 0.05 * f(j + 3.1 * i + k);
      sees.add_factor({i, np + j}, z, nullptr, Empty(), l3);
    }
  for (size_t i = 0; i < np; ++i) {
    Arr<6> z;
    for (int k = 0; k < 6; ++k) z(k) = poses[i](k) + 0.02 * f(i + 0.9 * k);
    prior.add_factor({i}, z, nullptr, Empty(), l6);
  }
  BlockJacobiSchurPreconditioner<double, double> bjs;
  std::unique_ptr<Solver<double, double>> solver;
  if (std::string(argv[1]) == "pcg-schur") solver.reset(new PCGSchurSolver<double, double>(60, 1e-14, 1e6, &bjs));
  else solver.reset(new EigenSchurLDLTSolver<double, double>());
  StreamPool streams(1);
  optimizer::LevenbergMarquardtOptions<double, double> opt;
  opt.solver = solver.get(); opt.initial_damping = 1e-3; opt.iterations = 6; opt.verbose = false; opt.streams = &streams;
  optimizer::levenberg_marquardt<double, double>(&graph, &opt);
  std::cout << std::setprecision(15) << "CHI2 " << graph.chi2() << std::endl;
  for (size_t i : {size_t(0), size_t(17), np - 1}) { std::cout << "POSE " << i; for (int k = 0; k < 6; ++k) std::cout << " " << poses[i](k); std::cout << std::endl; }
  for (size_t j : {size_t(0), size_t(123), nl - 1}) { std::cout << "LM " << j; for (int k = 0; k < 3; ++k) std::cout << " " << lms[j](k); std::cout << std::endl; }
  solver.reset();
  return 0;
}
