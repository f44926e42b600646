// A planar pose graph on the reference's generic API: ONE vertex descriptor (SE(2) poses, dimension 3), ONE binary factor descriptor
// whose two slots are that same descriptor (between-factors, error dimension 3, a full 3 x 3 information matrix per factor, optional
// Huber loss) — the SLAM back-end shape the reference's README names, which BASELINE configs[0] calls a "2D pose-graph".  The BAL
// clients exercise camera / landmark graphs; this one exercises what they cannot: both vertices of a factor NON-eliminated and of the
// same type, vertex-side sums that meet in one descriptor from both slots, precision matrices that are not diagonal.
//   usage: test_pose_graph <file> <pcg|pcg-identity> <iterations> <manual|auto> [pcg iterations] [pcg tolerance] [out file]
//   file: graphite_amd.synth.write_pose_graph;  out file: final poses, 17 digits
// Prints the LM table (levenberg_marquardt.hpp:216-221), FINAL_CHI2 and the per-iteration time; tests/test_generic_api.py compares
// with oracle/pose_graph.py.
#include <chrono>
#include <fstream>
#include <graphite/optimizer/levenberg_marquardt.hpp>
#include <graphite/preconditioner/block_jacobi.hpp>
#include <graphite/preconditioner/identity.hpp>
#include <graphite/solver/pcg.hpp>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

namespace graphite {

template <typename T> struct Pose2 { T x, y, th; };
template <typename T> struct Rel2 { T x, y, th; };

template <typename T> struct Pose2Traits {
  static constexpr size_t dimension = 3;
  using Vertex = Pose2<T>;
  template <typename P> d_fn static void parameters(const Vertex &v, P *p) { p[0] = P(v.x); p[1] = P(v.y); p[2] = P(v.th); }
  d_fn static void update(Vertex &v, const T *d) { v.x += d[0]; v.y += d[1]; v.th += d[2]; }
};
template <typename T, typename S> using Pose2Descriptor = VertexDescriptor<T, S, Pose2Traits<T>>;

// e = [R_i^T (t_j - t_i) - m_t ; th_j - th_i - m_th]
template <typename T, typename S, typename Mode, template <typename, int> class LossT> struct Between2Traits {
  static constexpr size_t dimension = 3;
  using VertexDescriptors = std::tuple<Pose2Descriptor<T, S>, Pose2Descriptor<T, S>>;
  using Observation = Rel2<T>;
  using Data = Empty;
  using Loss = LossT<T, 3>;
  using Differentiation = Mode;
  template <typename D> d_fn static void error(const D *a, const D *b, const Observation &m, D *e) {
    const D c = cos(a[2]), s = sin(a[2]);
    const D dx = b[0] - a[0], dy = b[1] - a[1];
    e[0] = c * dx + s * dy - D(m.x);
    e[1] = c * dy - s * dx - D(m.y);
    e[2] = b[2] - a[2] - D(m.th);
  }
  template <typename J, size_t I> d_fn static void jacobian(const Pose2<T> &a, const Pose2<T> &b, const Observation &, J *jac) {
    const T c = cos(a.th), s = sin(a.th), dx = b.x - a.x, dy = b.y - a.y;
    if constexpr (I == 0) { // 3 x 3 column-major: columns x_i, y_i, th_i
      jac[0] = J(-c); jac[1] = J(s); jac[2] = J(0);
      jac[3] = J(-s); jac[4] = J(-c); jac[5] = J(0);
      jac[6] = J(-s * dx + c * dy); jac[7] = J(-c * dx - s * dy); jac[8] = J(-1);
    } else {
      jac[0] = J(c); jac[1] = J(-s); jac[2] = J(0);
      jac[3] = J(s); jac[4] = J(c); jac[5] = J(0);
      jac[6] = J(0); jac[7] = J(0); jac[8] = J(1);
    }
  }
};

template <typename Mode, template <typename, int> class LossT> static int run(int argc, char **argv) {
  using T = double;
  using Factor = FactorDescriptor<T, T, Between2Traits<T, T, Mode, LossT>>;
  std::ifstream in(argv[1]);
  size_t n = 0, nf = 0;
  double delta = 0;
  if (!(in >> n >> nf >> delta)) { std::cerr << "cannot read " << argv[1] << std::endl; return 2; }
  const std::string solver = argv[2];
  const size_t iterations = std::stoul(argv[3]);
  const size_t pcg_it = argc > 5 ? std::stoul(argv[5]) : 10;
  const double pcg_tol = argc > 6 ? std::stod(argv[6]) : 1.0;
  managed_vector<Pose2<T>> poses(n);
  std::vector<int> fixed(n);
  for (size_t i = 0; i < n; ++i) { T x, y, th; in >> x >> y >> th >> fixed[i]; poses[i] = Pose2<T>{x, y, th}; }
  Graph<T, T> graph;
  Pose2Descriptor<T, T> vd;
  vd.reserve(n);
  graph.add_descriptor(&vd);
  for (size_t i = 0; i < n; ++i) vd.add_vertex(i, &poses[i], fixed[i] != 0);
  Factor fd(&vd, &vd);
  fd.reserve(nf);
  graph.add_descriptor(&fd);
  for (size_t f = 0; f < nf; ++f) {
    size_t i, j;
    T mx, my, mth, P[9];
    in >> i >> j >> mx >> my >> mth;
    for (int k = 0; k < 9; ++k) in >> P[k];
    if (!in) { std::cerr << "short file at factor " << f << std::endl; return 2; }
    if constexpr (std::is_same<LossT<T, 3>, HuberLoss<T, 3>>::value) fd.add_factor({i, j}, Rel2<T>{mx, my, mth}, P, Empty(), HuberLoss<T, 3>(delta));
    else fd.add_factor({i, j}, Rel2<T>{mx, my, mth}, P, Empty(), DefaultLoss<T, 3>());
  }
  BlockJacobiPreconditioner<T, T> bj;
  IdentityPreconditioner<T, T> ident;
  PCGSolver<T, T> pcg(pcg_it, pcg_tol, 5.0, solver == "pcg-identity" ? static_cast<Preconditioner<T, T> *>(&ident) : static_cast<Preconditioner<T, T> *>(&bj));
  StreamPool streams(1);
  optimizer::LevenbergMarquardtOptions<T, T> opt;
  opt.solver = &pcg;
  opt.initial_damping = 1e-4;
  opt.iterations = iterations;
  opt.optimization_level = 0;
  opt.verbose = true;
  opt.streams = &streams;
  std::cout << "POSES " << n << " FACTORS " << fd.internal_count() << std::endl;
  const auto t0 = std::chrono::steady_clock::now();
  optimizer::levenberg_marquardt<T, T>(&graph, &opt);
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::cout << std::setprecision(17) << "FINAL_CHI2 " << graph.chi2() << std::endl;
  std::cout << "ENGINE_HANDOVERS " << optimizer::engine_handover_count() << " ENGINE_MODEL_HANDOVERS " << optimizer::engine_model_handover_count() << std::endl;
  std::cout << std::setprecision(6) << "LM_SECONDS " << sec << " PER_ITERATION_US " << 1e6 * sec / (double)std::max<size_t>(iterations, 1) << std::endl;
  if (argc > 7) {
    std::ofstream out(argv[7]);
    out << std::setprecision(17);
    for (size_t i = 0; i < n; ++i) out << poses[i].x << " " << poses[i].y << " " << poses[i].th << "\n";
  }
  return 0;
}

} // namespace graphite

int main(int argc, char **argv) {
  if (argc < 5) { std::cerr << "usage: test_pose_graph <file> <pcg|pcg-identity> <iterations> <manual|auto|manual-huber> [pcg iterations] [pcg tolerance] [out file]" << std::endl; return 2; }
  (void)hipSetDevice(0);
  const std::string mode = argv[4];
  using namespace graphite;
  if (mode == "auto") return run<DifferentiationMode::Auto, DefaultLoss>(argc, argv);
  if (mode == "manual-huber") return run<DifferentiationMode::Manual, HuberLoss>(argc, argv);
  return run<DifferentiationMode::Manual, DefaultLoss>(argc, argv);
}
