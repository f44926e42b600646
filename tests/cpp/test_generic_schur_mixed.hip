// Generic-layer Schur elimination on a graph that is NOT bundle adjustment: 2-D SLAM with poses (x, y, theta;
// dimension 3), landmarks (x, y; dimension 2, set_eliminate), a unary prior, pose-pose odometry factors and
// pose-landmark observations (landmark seen in the pose frame), all by automatic differentiation.
// The reduced solvers must reproduce the full direct solve: EigenSchurLDLTSolver == EigenLDLTSolver up to rounding,
// PCGSchurSolver (run to convergence) to the PCG tolerance.
//   usage: test_generic_schur_mixed            prints one line per solver: name, final chi2, pose 3, landmark 2
#include <cmath>
#include <graphite/optimizer/levenberg_marquardt.hpp>
#include <graphite/preconditioner/block_jacobi_schur.hpp>
#include <graphite/solver/eigen.hpp>
#include <graphite/solver/eigen_schur.hpp>
#include <graphite/solver/pcg_schur.hpp>
#include <iomanip>
#include <iostream>
#include <random>
#include <vector>

namespace graphite {
template <int N> struct VecN {
  double v[N];
  hd_fn double operator()(int i) const { return v[i]; }
  hd_fn double &operator()(int i) { return v[i]; }
};
template <int N> struct VecNTraits {
  static constexpr size_t dimension = N;
  using Vertex = VecN<N>;
  template <typename P> d_fn static void parameters(const Vertex &x, P *p) { for (int i = 0; i < N; ++i) p[i] = P(x(i)); }
  d_fn static void update(Vertex &x, const double *d) { for (int i = 0; i < N; ++i) x(i) += d[i]; }
};
using PoseDescriptor = VertexDescriptor<double, double, VecNTraits<3>>;
using LandmarkDescriptor = VertexDescriptor<double, double, VecNTraits<2>>;

struct PriorTraits { // pose - observation
  static constexpr size_t dimension = 3;
  using VertexDescriptors = std::tuple<PoseDescriptor>;
  using Observation = VecN<3>;
  using Data = Empty;
  using Loss = DefaultLoss<double, 3>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *p, const Observation &o, D *e) { for (int i = 0; i < 3; ++i) e[i] = p[i] - D(o(i)); }
};
struct OdometryTraits { // pose b expressed in the frame of pose a, minus the measured relative motion
  static constexpr size_t dimension = 3;
  using VertexDescriptors = std::tuple<PoseDescriptor, PoseDescriptor>;
  using Observation = VecN<3>;
  using Data = Empty;
  using Loss = DefaultLoss<double, 3>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *a, const D *b, const Observation &o, D *e) {
    const D c = cos(a[2]), s = sin(a[2]), dx = b[0] - a[0], dy = b[1] - a[1];
    e[0] = c * dx + s * dy - D(o(0));
    e[1] = c * dy - s * dx - D(o(1));
    e[2] = b[2] - a[2] - D(o(2));
  }
};
struct SightingTraits { // landmark in the pose frame
  static constexpr size_t dimension = 2;
  using VertexDescriptors = std::tuple<PoseDescriptor, LandmarkDescriptor>;
  using Observation = VecN<2>;
  using Data = Empty;
  using Loss = HuberLoss<double, 2>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *p, const D *l, const Observation &o, D *e) {
    const D c = cos(p[2]), s = sin(p[2]), dx = l[0] - p[0], dy = l[1] - p[1];
    e[0] = c * dx + s * dy - D(o(0));
    e[1] = c * dy - s * dx - D(o(1));
  }
};
} // namespace graphite

int main() {
  using namespace graphite;
  (void)hipSetDevice(0);
  constexpr int NP = 12, NL = 30;
  std::mt19937 gen(11);
  std::normal_distribution<double> noise(0.0, 0.05), big(0.0, 0.3);
  std::vector<VecN<3>> true_pose(NP);
  std::vector<VecN<2>> true_lm(NL);
  for (int i = 0; i < NP; ++i) { const double a = 2 * M_PI * i / NP; true_pose[i] = {{5 * std::cos(a), 5 * std::sin(a), a + M_PI / 2}}; }
  std::uniform_real_distribution<double> box(-8.0, 8.0);
  for (auto &l : true_lm) l = {{box(gen), box(gen)}};
  auto rel = [](const VecN<3> &a, double x, double y, double &ox, double &oy) {
    const double c = std::cos(a(2)), s = std::sin(a(2));
    ox = c * (x - a(0)) + s * (y - a(1)); oy = c * (y - a(1)) - s * (x - a(0));
  };
  std::vector<VecN<3>> odo(NP - 1);
  for (int i = 0; i + 1 < NP; ++i) {
    double ox, oy; rel(true_pose[i], true_pose[i + 1](0), true_pose[i + 1](1), ox, oy);
    odo[i] = {{ox + noise(gen), oy + noise(gen), true_pose[i + 1](2) - true_pose[i](2) + 0.2 * noise(gen)}};
  }
  struct Sight { int p, l; VecN<2> o; };
  std::vector<Sight> sights;
  for (int l = 0; l < NL; ++l)
    for (int k = 0; k < 4; ++k) { // every landmark seen from 4 poses
      const int p = (l * 5 + k * 3) % NP;
      double ox, oy; rel(true_pose[p], true_lm[l](0), true_lm[l](1), ox, oy);
      sights.push_back({p, l, {{ox + noise(gen), oy + noise(gen)}}});
    }
  std::vector<VecN<3>> pose0(NP);
  std::vector<VecN<2>> lm0(NL);
  for (int i = 0; i < NP; ++i) pose0[i] = {{true_pose[i](0) + big(gen), true_pose[i](1) + big(gen), true_pose[i](2) + 0.1 * big(gen)}};
  for (int l = 0; l < NL; ++l) lm0[l] = {{true_lm[l](0) + big(gen), true_lm[l](1) + big(gen)}};

  int failures = 0;
  double ref_chi2 = 0;
  std::vector<double> ref_vals;
  for (int which = 0; which < 3; ++which) {
    managed_vector<VecN<3>> poses(NP);
    managed_vector<VecN<2>> lms(NL);
    for (int i = 0; i < NP; ++i) poses[i] = pose0[i];
    for (int l = 0; l < NL; ++l) lms[l] = lm0[l];
    Graph<double, double> graph;
    PoseDescriptor pd; LandmarkDescriptor ld;
    graph.add_descriptor(&ld); graph.add_descriptor(&pd); // eliminated descriptor added FIRST: columns must still come last
    for (int i = 0; i < NP; ++i) pd.add_vertex(100 + i, &poses[i]);
    for (int l = 0; l < NL; ++l) ld.add_vertex(1000 + l, &lms[l]);
    ld.set_eliminate(true);
    FactorDescriptor<double, double, PriorTraits> prior(&pd);
    FactorDescriptor<double, double, OdometryTraits> odom(&pd, &pd);
    FactorDescriptor<double, double, SightingTraits> sight(&pd, &ld);
    graph.add_descriptor(&prior); graph.add_descriptor(&odom); graph.add_descriptor(&sight);
    prior.add_factor({100}, true_pose[0]);
    for (int i = 0; i + 1 < NP; ++i) odom.add_factor({100u + (size_t)i, 101u + (size_t)i}, odo[i]);
    const HuberLoss<double, 2> huber(1.0);
    for (auto &s : sights) sight.add_factor({100u + (size_t)s.p, 1000u + (size_t)s.l}, s.o, nullptr, Empty(), huber);

    BlockJacobiSchurPreconditioner<double, double> bjs;
    EigenLDLTSolver<double, double> full;
    EigenSchurLDLTSolver<double, double> direct;
    PCGSchurSolver<double, double> pcg(200, 1e-24, 1e30, &bjs);
    Solver<double, double> *solver = which == 0 ? (Solver<double, double> *)&full : which == 1 ? (Solver<double, double> *)&direct : (Solver<double, double> *)&pcg;
    StreamPool streams(1);
    optimizer::LevenbergMarquardtOptions<double, double> options;
    options.solver = solver; options.iterations = 12; options.initial_damping = 1e-4; options.streams = &streams;
    const bool ok = optimizer::levenberg_marquardt<double, double>(&graph, &options);
    const double chi2 = graph.chi2();
    std::vector<double> vals;
    for (int i = 0; i < NP; ++i) for (int k = 0; k < 3; ++k) vals.push_back(poses[i](k));
    for (int l = 0; l < NL; ++l) for (int k = 0; k < 2; ++k) vals.push_back(lms[l](k));
    const char *name = which == 0 ? "eigen" : which == 1 ? "eigen-schur" : "pcg-schur";
    std::cout << std::setprecision(15) << name << " ok=" << ok << " hessian=" << graph.get_hessian_dimension() << " pose_dim=" << graph.get_pose_dimension()
              << " chi2=" << chi2 << " pose3=(" << poses[3](0) << "," << poses[3](1) << "," << poses[3](2) << ") lm2=(" << lms[2](0) << "," << lms[2](1) << ")" << std::endl;
    failures += graph.get_hessian_dimension() != 3 * NP + 2 * NL || graph.get_pose_dimension() != 3 * NP;
    if (which == 0) { ref_chi2 = chi2; ref_vals = vals; failures += !(chi2 < 5.0); }
    else {
      const double tol = which == 1 ? 1e-9 : 1e-6;
      failures += !(std::abs(chi2 - ref_chi2) <= tol * ref_chi2);
      double worst = 0;
      for (size_t k = 0; k < vals.size(); ++k) worst = std::max(worst, std::abs(vals[k] - ref_vals[k]));
      std::cout << "  max |parameter - eigen| = " << worst << std::endl;
      failures += !(worst < 1e-5); // the runs stop (rho == 0) at rounding-dependent trips inside the converged basin
    }
  }
  std::cout << (failures ? "FAILED" : "OK") << " (" << failures << " failures)" << std::endl;
  return failures != 0;
}
