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
#include <graphite/solver/eigen.hpp>
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

// unary prior on a pose: e = x - m, Jacobian = identity (a SECOND factor descriptor on the same vertex descriptor: the optional PRIORS section of the file)
template <typename T, typename S> struct Prior2Traits {
  static constexpr size_t dimension = 3;
  using VertexDescriptors = std::tuple<Pose2Descriptor<T, S>>;
  using Observation = Rel2<T>;
  using Data = Empty;
  using Loss = DefaultLoss<T, 3>;
  using Differentiation = DifferentiationMode::Manual;
  template <typename D> d_fn static void error(const D *a, const Observation &m, D *e) { e[0] = a[0] - D(m.x); e[1] = a[1] - D(m.y); e[2] = a[2] - D(m.th); }
  template <typename J, size_t I> d_fn static void jacobian(const Pose2<T> &, const Observation &, J *jac) {
    for (int k = 0; k < 9; ++k) jac[k] = J(k % 4 == 0 ? 1 : 0);
  }
};

template <typename T, typename Mode, template <typename, int> class LossT> static int run(int argc, char **argv) {
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
  // optional second factor descriptor: unary priors
  FactorDescriptor<T, T, Prior2Traits<T, T>> pd(&vd);
  {
    std::string word;
    size_t np = 0;
    if (in >> word >> np && word == "PRIORS") {
      pd.reserve(np);
      graph.add_descriptor(&pd);
      for (size_t q = 0; q < np; ++q) {
        size_t i;
        T mx, my, mth, P[9];
        in >> i >> mx >> my >> mth;
        for (int k = 0; k < 9; ++k) in >> P[k];
        if (!in) { std::cerr << "short file at prior " << q << std::endl; return 2; }
        pd.add_factor({i}, Rel2<T>{mx, my, mth}, P, Empty(), DefaultLoss<T, 3>());
      }
      std::cout << "PRIORS " << np << std::endl;
    }
  }
  // POSE_DEACTIVATE=k: every k-th factor gets activity level 1 (inactive at optimisation level 0: factor.hpp:419-431, active.hpp:11-21);
  // POSE_EXTRA_VERTICES=m: m vertices no factor touches are added behind the poses (they take no part in the optimisation)
  if (getenv("POSE_DEACTIVATE")) { const size_t k = (size_t)std::max(2, atoi(getenv("POSE_DEACTIVATE"))); for (size_t f = 0; f < nf; f += k) fd.set_active(f, 1); }
  managed_vector<Pose2<T>> extra(getenv("POSE_EXTRA_VERTICES") ? (size_t)atoi(getenv("POSE_EXTRA_VERTICES")) : 0);
  for (size_t i = 0; i < extra.size(); ++i) { extra[i] = Pose2<T>{T(100 + i), T(-3), T(0.5)}; vd.add_vertex(n + 10 + i, &extra[i], false); }
  BlockJacobiPreconditioner<T, T> bj;
  IdentityPreconditioner<T, T> ident;
  PCGSolver<T, T> pcg(pcg_it, pcg_tol, 5.0, solver == "pcg-identity" ? static_cast<Preconditioner<T, T> *>(&ident) : static_cast<Preconditioner<T, T> *>(&bj));
  EigenLDLTSolver<T, T> ldlt; // solver "eigen": the direct solve of the whole damped system (solver/eigen.hpp:49-98)
  StreamPool streams(1);
  optimizer::LevenbergMarquardtOptions<T, T> opt;
  opt.solver = solver == "eigen" ? static_cast<Solver<T, T> *>(&ldlt) : static_cast<Solver<T, T> *>(&pcg);
  opt.initial_damping = 1e-4;
  opt.iterations = iterations;
  opt.optimization_level = 0;
  opt.verbose = true;
  opt.use_identity = getenv("POSE_IDENTITY_DAMPING") && atoi(getenv("POSE_IDENTITY_DAMPING")) != 0; // mu I instead of mu clamp(diag)
  opt.streams = &streams;
  bool stop = getenv("POSE_STOP") && atoi(getenv("POSE_STOP")) != 0; // already raised: the loop ends after its first iteration (levenberg_marquardt.hpp:232)
  if (getenv("POSE_STOP")) opt.stop_flag = &stop;
  const bool lm2 = getenv("POSE_LM2") && atoi(getenv("POSE_LM2")) != 0;
  std::cout << "POSES " << n << " FACTORS " << fd.internal_count() << std::endl;
  // POSE_REPEAT=n: the optimiser is called n times on the same graph FROM THE SAME START (the poses are put back in between) and the
  // LAST call is the one reported: what a process that optimises again and again pays per call (no first-use costs)
  const int repeat = getenv("POSE_REPEAT") ? std::max(1, atoi(getenv("POSE_REPEAT"))) : 1;
  std::vector<Pose2<T>> start(poses.begin(), poses.end());
  double sec = 0;
  for (int rep = 0; rep < repeat; ++rep) {
    if (rep) { std::copy(start.begin(), start.end(), poses.begin()); std::cout << "REPEAT " << rep << std::endl; }
    // POSE_MUTATE=k: before the LAST call pose k is fixed too (a structure change between optimiser calls: what was cached for the graph no longer holds)
    if (rep && rep == repeat - 1 && getenv("POSE_MUTATE")) vd.set_fixed((size_t)atoi(getenv("POSE_MUTATE")), true);
    const auto t0 = std::chrono::steady_clock::now();
    if (lm2) optimizer::levenberg_marquardt2<T, T>(&graph, &opt); else optimizer::levenberg_marquardt<T, T>(&graph, &opt);
    sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  std::cout << std::setprecision(17) << "FINAL_CHI2 " << graph.chi2() << std::endl;
  if (solver == "eigen") std::cout << "SPARSE_FACTORISATION " << (ldlt.uses_sparse_factorisation() ? 1 : 0) << std::endl;
  std::cout << "ENGINE_HANDOVERS " << optimizer::engine_handover_count() << " ENGINE_MODEL_HANDOVERS " << optimizer::engine_model_handover_count()
            << " POSE_ENGINE_HANDOVERS " << optimizer::pose_engine_handover_count() << std::endl;
  if (optimizer::pose_engine_handover_count())
    std::cout << std::setprecision(6) << "ENGINE_SETUP_SECONDS " << optimizer::engine_last_setup_seconds() << " ENGINE_LOOP_SECONDS " << optimizer::engine_last_loop_seconds()
              << " ENGINE_LOOP_PER_ITERATION_US " << 1e6 * optimizer::engine_last_loop_seconds() / std::max(1, optimizer::engine_last_iterations()) << std::endl;
  std::cout << std::setprecision(6) << "LM_SECONDS " << sec << " PER_ITERATION_US " << 1e6 * sec / (double)std::max<size_t>(iterations, 1) << std::endl;
  if (argc > 7) {
    std::ofstream out(argv[7]);
    out << std::setprecision(17);
    for (size_t i = 0; i < n; ++i) out << poses[i].x << " " << poses[i].y << " " << poses[i].th << "\n";
  }
  return 0;
}


// A 6-dimensional toy (tangent 6, error 6, dense 6 x 6 information matrices, dual-number Jacobians): the dimensions of an SE(3) graph
// through the engine's templates — 96-byte direction records, 6 x 6 block inverses.  No file, no oracle: the graph is generated here and
// the test compares the engine with the generic kernels (GRAPHITE_POSE_ENGINE=0).   usage: test_pose_graph vec6 <n> <iterations> <x> [out]
template <typename T> struct V6 { T v[6]; };
template <typename T> struct M6 { T m[6]; };
template <typename T> struct V6Traits {
  static constexpr size_t dimension = 6;
  using Vertex = V6<T>;
  template <typename P> d_fn static void parameters(const Vertex &x, P *p) { for (int i = 0; i < 6; ++i) p[i] = P(x.v[i]); }
  d_fn static void update(Vertex &x, const T *d) { for (int i = 0; i < 6; ++i) x.v[i] += d[i]; }
};
template <typename T, typename S> using V6Descriptor = VertexDescriptor<T, S, V6Traits<T>>;
template <typename T, typename S> struct Between6Traits {
  static constexpr size_t dimension = 6;
  using VertexDescriptors = std::tuple<V6Descriptor<T, S>, V6Descriptor<T, S>>;
  using Observation = M6<T>;
  using Data = Empty;
  using Loss = DefaultLoss<T, 6>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *a, const D *b, const Observation &m, D *e) {
    for (int i = 0; i < 6; ++i) e[i] = (b[i] - a[i]) + D(T(0.3)) * sin(b[(i + 1) % 6] - a[(i + 1) % 6]) * cos(a[(i + 2) % 6]) - D(m.m[i]);
  }
};
static int run_vec6(int argc, char **argv) {
  using T = double;
  const size_t n = std::stoul(argv[2]), iterations = std::stoul(argv[3]);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
  managed_vector<V6<T>> xs(n);
  std::vector<V6<T>> truth(n);
  for (size_t i = 0; i < n; ++i) for (int k = 0; k < 6; ++k) { truth[i].v[k] = 0.05 * (double)i * (k + 1) + rnd(); xs[i].v[k] = truth[i].v[k] + 0.2 * rnd(); }
  xs[0] = truth[0];
  Graph<T, T> graph;
  V6Descriptor<T, T> vd;
  vd.reserve(n);
  graph.add_descriptor(&vd);
  for (size_t i = 0; i < n; ++i) vd.add_vertex(i, &xs[i], i == 0);
  FactorDescriptor<T, T, Between6Traits<T, T>> fd(&vd, &vd);
  graph.add_descriptor(&fd);
  size_t nf = 0;
  for (size_t i = 0; i < n; ++i)
    for (int e = 0; e < 3; ++e) {
      const size_t j = e == 0 ? i + 1 : (size_t)((rnd() + 0.5) * (double)n);
      if (j >= n || j == i) continue;
      M6<T> m;
      for (int k = 0; k < 6; ++k) { // the model's value at the truth + noise
        const double d1 = truth[j].v[(k + 1) % 6] - truth[i].v[(k + 1) % 6];
        m.m[k] = (truth[j].v[k] - truth[i].v[k]) + 0.3 * std::sin(d1) * std::cos(truth[i].v[(k + 2) % 6]) + 0.02 * rnd();
      }
      T L[36], P[36];
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) L[r * 6 + c] = c < r ? 0.4 * rnd() : (c == r ? 1.0 + 0.5 * (rnd() + 0.5) : 0.0);
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { T s = 0; for (int k = 0; k < 6; ++k) s += L[r * 6 + k] * L[c * 6 + k]; P[r * 6 + c] = s; }
      for (int r = 0; r < 6; ++r) for (int c = r + 1; c < 6; ++c) P[c * 6 + r] = P[r * 6 + c]; // bit-symmetric
      fd.add_factor({i, j}, m, P, Empty(), DefaultLoss<T, 6>());
      ++nf;
    }
  BlockJacobiPreconditioner<T, T> bj;
  PCGSolver<T, T> pcg(20, 1e-10, 5.0, &bj);
  StreamPool streams(1);
  optimizer::LevenbergMarquardtOptions<T, T> opt;
  opt.solver = &pcg; opt.initial_damping = 1e-4; opt.iterations = iterations; opt.optimization_level = 0; opt.verbose = true; opt.streams = &streams;
  std::cout << "POSES " << n << " FACTORS " << nf << std::endl;
  optimizer::levenberg_marquardt<T, T>(&graph, &opt);
  std::cout << std::setprecision(17) << "FINAL_CHI2 " << graph.chi2() << std::endl;
  std::cout << "ENGINE_HANDOVERS " << optimizer::engine_handover_count() << " ENGINE_MODEL_HANDOVERS " << optimizer::engine_model_handover_count()
            << " POSE_ENGINE_HANDOVERS " << optimizer::pose_engine_handover_count() << std::endl;
  if (argc > 5) {
    std::ofstream out(argv[5]);
    out << std::setprecision(17);
    for (size_t i = 0; i < n; ++i) { for (int k = 0; k < 6; ++k) out << xs[i].v[k] << " "; out << "\n"; }
  }
  return 0;
}

} // namespace graphite

int main(int argc, char **argv) {
  if (argc < 5) { std::cerr << "usage: test_pose_graph <file> <pcg|pcg-identity> <iterations> <manual|auto|manual-huber> [pcg iterations] [pcg tolerance] [out file]" << std::endl; return 2; }
  (void)hipSetDevice(0);
  const std::string mode = argv[4];
  using namespace graphite;
  if (std::string(argv[1]) == "vec6") return run_vec6(argc, argv);
  if (mode == "auto") return run<double, DifferentiationMode::Auto, DefaultLoss>(argc, argv);
  if (mode == "manual-huber") return run<double, DifferentiationMode::Manual, HuberLoss>(argc, argv);
  if (mode == "manual-f32") return run<float, DifferentiationMode::Manual, DefaultLoss>(argc, argv);
  return run<double, DifferentiationMode::Manual, DefaultLoss>(argc, argv);
}
