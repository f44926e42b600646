// The engine's kernels instantiated on USER traits (include/graphite/engine_model.hpp): graphs that are NOT the library's
// built-in camera model — or that carry per-factor precision matrices / losses / constraint data — optimised by
// levenberg_marquardt on the gr_bal engine.  tests/test_engine_model.py compares the printed chi2 traces with the CPU oracle
// and with the generic kernels (GRAPHITE_GENERIC_ONLY=1) and checks which path ran.
//   usage: test_engine_model <bal file> <pcg|pcg-identity|pcg-schur|eigen-schur> <iterations> <bal|weighted|k3|pinhole> [stored|dynamic] [fp64|fp32|mixed]
//   bal      : the reprojection factor of docs/markdown/main.md:230-262 (Manual Jacobian), identity precision, DefaultLoss
//   weighted : the same factor with a closed-form Jacobian, every factor with its own 2 x 2 information matrix and its own Huber delta (factor.hpp:373-412)
//   k3       : a sixth-order radial term whose coefficient is per-factor constraint data (Traits::Data), dual-number Jacobian
//   pinhole  : 6-dof pose [angle-axis, t] x 3-d point -> pixel, intrinsics in the constraint data
#include <fstream>
#include <graphite/optimizer/levenberg_marquardt.hpp>
#include <graphite/preconditioner/block_jacobi.hpp>
#include <graphite/preconditioner/identity.hpp>
#include <graphite/preconditioner/block_jacobi_schur.hpp>
#include <graphite/solver/eigen_schur.hpp>
#include <graphite/solver/pcg.hpp>
#include <graphite/solver/pcg_schur.hpp>
#include <iostream>
#include <memory>
#include <string>

namespace graphite {

template <typename T, int N> struct Vec {
  T v[N];
  hd_fn T operator()(int i) const { return v[i]; }
  hd_fn T &operator()(int i) { return v[i]; }
};
template <typename T> using Pixel = Vec<T, 2>;

template <typename T, int N> struct VecTraits {
  static constexpr size_t dimension = N;
  using Vertex = Vec<T, N>;
  template <typename P> d_fn static void parameters(const Vertex &x, P *p) { for (int i = 0; i < N; ++i) p[i] = P(x(i)); }
  d_fn static void update(Vertex &x, const T *d) { for (int i = 0; i < N; ++i) x(i) += d[i]; }
};
template <typename T, typename S, int N> using VecDescriptor = VertexDescriptor<T, S, VecTraits<T, N>>;

// P = R(r) X + t, angle-axis rotation as a Rodrigues matrix; theta == 0 -> identity
template <typename D, typename T> d_fn void transform(const D *pose, const D *pt, D *P) {
  const D rx = pose[0], ry = pose[1], rz = pose[2];
  const D theta2 = rx * rx + ry * ry + rz * rz;
  if (theta2 > D(T(0))) {
    const D theta = sqrt(theta2);
    const D ax = rx / theta, ay = ry / theta, az = rz / theta;
    const D s = sin(theta), c = cos(theta), k = D(T(1)) - c;
    const D R[9] = {k * ax * ax + c,      k * ax * ay - s * az, k * ax * az + s * ay,
                    k * ax * ay + s * az, k * ay * ay + c,      k * ay * az - s * ax,
                    k * ax * az - s * ay, k * ay * az + s * ax, k * az * az + c};
    for (int i = 0; i < 3; ++i) P[i] = R[3 * i] * pt[0] + R[3 * i + 1] * pt[1] + R[3 * i + 2] * pt[2] + pose[3 + i];
  } else {
    for (int i = 0; i < 3; ++i) P[i] = pt[i] + pose[3 + i];
  }
}
// Snavely camera with an optional sixth-order radial term
template <typename D, typename T> d_fn void reprojection(const D *cam, const D *pt, const Pixel<T> &obs, T k3, D *err) {
  D P[3];
  transform<D, T>(cam, pt, P);
  const D px = -P[0] / P[2], py = -P[1] / P[2];
  const D r2 = px * px + py * py;
  const D d = D(T(1)) + cam[7] * r2 + cam[8] * r2 * r2 + D(k3) * r2 * r2 * r2;
  err[0] = cam[6] * d * px - D(obs(0));
  err[1] = cam[6] * d * py - D(obs(1));
}

// closed-form derivative of `reprojection` (k3 = 0) with respect to slot I (0: camera, 9 columns; 1: point, 3 columns), column-major 2 x d:
// err = f d p - obs, p = -(P0, P1) / P2, P = R(w) X + t.  With M = d err / d P (2 x 3) and N = M R:
//   d err / d X = N,  d err / d t = M,  d err / d w: row = ((u . w) w + (R u - u) x w) / theta^2 with u = X x n  (n = the row of N; theta = 0: u)
//   — the derivative of a rotated vector with respect to its rotation vector, d (R X) / d w = -R [X]x (w w^T + (R^T - I) [w]x) / theta^2 —
//   d err / d f = d p,  d err / d k1 = f r^2 p,  d err / d k2 = f r^4 p
template <typename T, typename Sj, size_t I> d_fn void reprojection_jacobian(const Vec<T, 9> &cam, const Vec<T, 3> &pt, Sj *jac) {
  const T wx = cam(0), wy = cam(1), wz = cam(2), X0 = pt(0), X1 = pt(1), X2 = pt(2);
  const T theta2 = wx * wx + wy * wy + wz * wz;
  T R[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
  if (theta2 > T(0)) {
    const T theta = sqrt(theta2), ax = wx / theta, ay = wy / theta, az = wz / theta;
    const T s = sin(theta), c = cos(theta), k = T(1) - c;
    R[0] = k * ax * ax + c;      R[1] = k * ax * ay - s * az; R[2] = k * ax * az + s * ay;
    R[3] = k * ax * ay + s * az; R[4] = k * ay * ay + c;      R[5] = k * ay * az - s * ax;
    R[6] = k * ax * az - s * ay; R[7] = k * ay * az + s * ax; R[8] = k * az * az + c;
  }
  const T P0 = R[0] * X0 + R[1] * X1 + R[2] * X2 + cam(3), P1 = R[3] * X0 + R[4] * X1 + R[5] * X2 + cam(4), P2 = R[6] * X0 + R[7] * X1 + R[8] * X2 + cam(5);
  const T iz = T(1) / P2, px = -P0 * iz, py = -P1 * iz, r2 = px * px + py * py;
  const T f = cam(6), k1 = cam(7), k2 = cam(8), d = T(1) + k1 * r2 + k2 * r2 * r2, g = T(2) * (k1 + T(2) * k2 * r2);
  // A = d err / d p = f (d I + g p p^T);  d p / d P = [[-iz, 0, -px iz], [0, -iz, -py iz]]
  const T a00 = f * (d + g * px * px), a01 = f * g * px * py, a11 = f * (d + g * py * py);
  const T M[2][3] = {{-a00 * iz, -a01 * iz, -(a00 * px + a01 * py) * iz}, {-a01 * iz, -a11 * iz, -(a01 * px + a11 * py) * iz}};
  for (int r = 0; r < 2; ++r) {
    const T n0 = M[r][0] * R[0] + M[r][1] * R[3] + M[r][2] * R[6], n1 = M[r][0] * R[1] + M[r][1] * R[4] + M[r][2] * R[7], n2 = M[r][0] * R[2] + M[r][1] * R[5] + M[r][2] * R[8];
    if constexpr (I == 1) { jac[r] = (Sj)n0; jac[2 + r] = (Sj)n1; jac[4 + r] = (Sj)n2; }
    else {
      const T u0 = X1 * n2 - X2 * n1, u1 = X2 * n0 - X0 * n2, u2 = X0 * n1 - X1 * n0;
      T j0 = u0, j1 = u1, j2 = u2;
      if (theta2 > T(0)) {
        const T v0 = R[0] * u0 + R[1] * u1 + R[2] * u2 - u0, v1 = R[3] * u0 + R[4] * u1 + R[5] * u2 - u1, v2 = R[6] * u0 + R[7] * u1 + R[8] * u2 - u2;
        const T uw = u0 * wx + u1 * wy + u2 * wz, it2 = T(1) / theta2;
        j0 = (uw * wx + (v1 * wz - v2 * wy)) * it2; j1 = (uw * wy + (v2 * wx - v0 * wz)) * it2; j2 = (uw * wz + (v0 * wy - v1 * wx)) * it2;
      }
      const T pr = r == 0 ? px : py;
      jac[r] = (Sj)j0; jac[2 + r] = (Sj)j1; jac[4 + r] = (Sj)j2;
      jac[6 + r] = (Sj)M[r][0]; jac[8 + r] = (Sj)M[r][1]; jac[10 + r] = (Sj)M[r][2];
      jac[12 + r] = (Sj)(d * pr); jac[14 + r] = (Sj)(f * r2 * pr); jac[16 + r] = (Sj)(f * r2 * r2 * pr);
    }
  }
}

// ---- bal / weighted: Manual Jacobian, HuberLoss per factor --------------------------------------------------------------------
// ANALYTIC false: the Jacobian by dual numbers INSIDE the user function (arbitrary user code on the Manual path); true: the closed form above,
// as a user with a generated Jacobian supplies it (examples/bal.cu)
template <typename T, typename S, typename LossT, bool ANALYTIC> struct ReprojectionTraits {
  static constexpr size_t dimension = 2;
  using VertexDescriptors = std::tuple<VecDescriptor<T, S, 9>, VecDescriptor<T, S, 3>>;
  using Observation = Pixel<T>;
  using Data = Empty;
  using Loss = LossT;
  using Differentiation = DifferentiationMode::Manual;
  template <typename D> d_fn static void error(const D *camera, const D *point, const Observation &obs, D *error) {
    reprojection<D, T>(camera, point, obs, T(0), error);
  }
  template <typename Sj, size_t I>
  d_fn static void jacobian(const Vec<T, 9> &cam, const Vec<T, 3> &pt, const Pixel<T> &obs, Sj *jac) {
    if constexpr (ANALYTIC) { reprojection_jacobian<T, Sj, I>(cam, pt, jac); (void)obs; }
    else {
      using D = Dual<T, T>;
      constexpr int d = I == 0 ? 9 : 3;
      for (int c = 0; c < d; ++c) {
        D cp[9], pp[3], err[2];
        for (int k = 0; k < 9; ++k) cp[k] = D(cam(k));
        for (int k = 0; k < 3; ++k) pp[k] = D(pt(k));
        (I == 0 ? cp[c] : pp[c]).dual = T(1);
        reprojection<D, T>(cp, pp, obs, T(0), err);
        jac[2 * c] = (Sj)err[0].dual;
        jac[2 * c + 1] = (Sj)err[1].dual;
      }
    }
  }
};
template <typename T, typename S> using BalFactor = FactorDescriptor<T, S, ReprojectionTraits<T, S, DefaultLoss<T, 2>, false>>;
template <typename T, typename S> using WeightedFactor = FactorDescriptor<T, S, ReprojectionTraits<T, S, HuberLoss<T, 2>, true>>;

// ---- k3: constraint data + automatic differentiation ------------------------------------------------------------------------
template <typename T> struct Radial6 { T k3; };
template <typename T, typename S> struct K3Traits {
  static constexpr size_t dimension = 2;
  using VertexDescriptors = std::tuple<VecDescriptor<T, S, 9>, VecDescriptor<T, S, 3>>;
  using Observation = Pixel<T>;
  using Data = Radial6<T>;
  using Loss = DefaultLoss<T, 2>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *camera, const D *point, const Observation &obs, const Data &data, D *error) {
    reprojection<D, T>(camera, point, obs, data.k3, error);
  }
};
template <typename T, typename S> using K3Factor = FactorDescriptor<T, S, K3Traits<T, S>>;

// ---- pinhole: (6, 3) -> 2 ---------------------------------------------------------------------------------------------------
template <typename T> struct Intrinsics { T fx, fy, cx, cy; };
template <typename T, typename S> struct PinholeTraits {
  static constexpr size_t dimension = 2;
  using VertexDescriptors = std::tuple<VecDescriptor<T, S, 6>, VecDescriptor<T, S, 3>>;
  using Observation = Pixel<T>;
  using Data = Intrinsics<T>;
  using Loss = HuberLoss<T, 2>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *pose, const D *point, const Observation &obs, const Data &k, D *error) {
    D P[3];
    transform<D, T>(pose, point, P);
    error[0] = D(k.fx) * (P[0] / P[2]) + D(k.cx) - D(obs(0));
    error[1] = D(k.fy) * (P[1] / P[2]) + D(k.cy) - D(obs(1));
  }
};
template <typename T, typename S> using PinholeFactor = FactorDescriptor<T, S, PinholeTraits<T, S>>;

// ---- slam2d: (3, 2) -> 2 range-bearing and (3, 2) -> 1 range-only sightings of 2-D landmarks from SE(2) poses -------------------
// (the zero-padded layout at its smallest: pose block 3 of 9, landmark block 2 of 3, error 1 of 2)
template <typename T, typename S, int EDIM> struct SightingTraits {
  static constexpr size_t dimension = EDIM;
  using VertexDescriptors = std::tuple<VecDescriptor<T, S, 3>, VecDescriptor<T, S, 2>>;
  using Observation = Vec<T, EDIM>;
  using Data = Empty;
  using Loss = HuberLoss<T, EDIM>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *pose, const D *lm, const Observation &obs, D *error) {
    const D dx = lm[0] - pose[0], dy = lm[1] - pose[1];
    error[0] = sqrt(dx * dx + dy * dy) - D(obs(0));
    if constexpr (EDIM == 2) error[1] = atan2(dy, dx) - pose[2] - D(obs(1));
  }
};
template <typename T, typename S> using RangeBearingFactor = FactorDescriptor<T, S, SightingTraits<T, S, 2>>;
template <typename T, typename S> using RangeFactor = FactorDescriptor<T, S, SightingTraits<T, S, 1>>;

} // namespace graphite

// 2-D SLAM toy: poses on a circle, landmarks in a square, every pose sights the landmarks within reach; pose 0 is fixed (gauge)
template <typename FP, typename SP, template <typename, typename> class Factor, int EDIM>
static int run_slam2d(int argc, char **argv) {
  using namespace graphite;
  (void)hipSetDevice(0);
  const size_t NP = 60, NL = 400;
  uint64_t state = 0x2545F4914F6CDD1Dull;
  auto uni = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (double)(state >> 11) / 9007199254740992.0; };
  std::vector<std::array<double, 3>> pose_true(NP);
  std::vector<std::array<double, 2>> lm_true(NL);
  for (size_t i = 0; i < NP; ++i) { const double a = 2.0 * M_PI * i / NP; pose_true[i] = {10.0 * std::cos(a), 10.0 * std::sin(a), a + M_PI / 2 - 2.0 * M_PI * (a + M_PI / 2 > M_PI ? 1 : 0)}; }
  for (size_t l = 0; l < NL; ++l) lm_true[l] = {24.0 * uni() - 12.0, 24.0 * uni() - 12.0};
  managed_vector<Vec<FP, 3>> poses(NP);
  managed_vector<Vec<FP, 2>> lms(NL);
  for (size_t i = 0; i < NP; ++i) for (int k = 0; k < 3; ++k) poses[i](k) = (FP)(pose_true[i][k] + (i == 0 ? 0.0 : 0.05 * (uni() - 0.5)));
  for (size_t l = 0; l < NL; ++l) for (int k = 0; k < 2; ++k) lms[l](k) = (FP)(lm_true[l][k] + 0.2 * (uni() - 0.5));
  Graph<FP, SP> graph;
  VecDescriptor<FP, SP, 3> pose_desc;
  VecDescriptor<FP, SP, 2> lm_desc;
  pose_desc.reserve(NP); lm_desc.reserve(NL);
  graph.add_descriptor(&pose_desc);
  graph.add_descriptor(&lm_desc);
  for (size_t i = 0; i < NP; ++i) pose_desc.add_vertex(i, &poses[i], i == 0);
  for (size_t l = 0; l < NL; ++l) lm_desc.add_vertex(NP + l, &lms[l]);
  lm_desc.set_eliminate(true);
  Factor<FP, SP> f_desc(&pose_desc, &lm_desc);
  graph.add_descriptor(&f_desc);
  size_t nf = 0;
  for (size_t i = 0; i < NP; ++i)
    for (size_t l = 0; l < NL; ++l) {
      const double dx = lm_true[l][0] - pose_true[i][0], dy = lm_true[l][1] - pose_true[i][1], r = std::sqrt(dx * dx + dy * dy);
      if (r > 7.0 || r < 0.5) continue;
      Vec<FP, EDIM> ob;
      ob(0) = (FP)(r + 0.02 * (uni() - 0.5));
      if constexpr (EDIM == 2) {
        // (the error function does not wrap angles: only sightings whose global bearing and whose bearing relative to the pose both stay
        // well inside (-pi, pi) are kept, so that no estimate near the truth crosses the branch cut of atan2)
        const double g = std::atan2(dy, dx), b = g - pose_true[i][2];
        if (std::abs(g) > 2.6 || std::abs(b) > 2.6) continue;
        ob(1) = (FP)(b + 0.005 * (uni() - 0.5));
      }
      SP P[EDIM * EDIM];
      for (int a = 0; a < EDIM; ++a) for (int c = 0; c < EDIM; ++c) P[a * EDIM + c] = a == c ? (a == 0 ? SP(4) : SP(100)) : SP(0); // information: range 0.5, bearing 0.1
      f_desc.add_factor({i, NP + l}, ob, P, Empty(), typename Factor<FP, SP>::LossType((FP)(3.0 + (nf % 3))));
      ++nf;
    }
  const std::string kind = argv[2];
  BlockJacobiPreconditioner<FP, SP> bj;
  BlockJacobiSchurPreconditioner<FP, SP> bjs;
  std::unique_ptr<Solver<FP, SP>> solver;
  if (kind == "pcg") solver.reset(new PCGSolver<FP, SP>(10, 1e-6, 5.0, &bj));
  else if (kind == "pcg-schur") solver.reset(new PCGSchurSolver<FP, SP>(10, 1e-6, 5.0, &bjs));
  else if (kind == "eigen-schur") solver.reset(new EigenSchurLDLTSolver<FP, SP>());
  else return 2;
  StreamPool streams(2);
  optimizer::LevenbergMarquardtOptions<FP, SP> options;
  options.solver = solver.get();
  options.initial_damping = 1e-4;
  options.iterations = std::stoul(argv[3]);
  options.verbose = true;
  options.streams = &streams;
  const bool ok = optimizer::levenberg_marquardt<FP, SP>(&graph, &options);
  std::cout << std::setprecision(17) << "FINAL_CHI2 " << graph.chi2() << std::endl << "FACTORS " << nf << std::endl;
  std::cout << "CAM0 " << poses[0](0) << " " << poses[0](1) << " " << poses[0](2) << std::endl << "PT0 " << lms[0](0) << " " << lms[0](1) << " 0" << std::endl;
  std::cout << "POSE1 " << poses[1](0) << " " << poses[1](1) << " " << poses[1](2) << std::endl;
  std::cout << "ENGINE_HANDOVERS " << optimizer::engine_handover_count() << std::endl
            << "ENGINE_MODEL_HANDOVERS " << optimizer::engine_model_handover_count() << std::endl << (ok ? "OK" : "STOPPED") << std::endl;
  solver.reset();
  return 0;
}

// per-factor weights, the same closed forms tests/test_engine_model.py evaluates
static void information(size_t f, double (&P)[4]) {
  const double a = 0.5 + (double)(f % 7) / 4.0, c = 0.75 + (double)(f % 5) / 8.0;
  const double b = 0.25 * ((double)(f % 3) - 1.0) * std::sqrt(a * c);
  P[0] = a; P[1] = b; P[2] = b; P[3] = c;
}
static double huber_delta(size_t f) { return 1.0 + (double)(f % 4); }
static double k3_of(size_t f) { return 0.01 * ((double)(f % 5) - 2.0); }

template <typename FP, typename SP, template <typename, typename> class Factor, int DC>
static int run(int argc, char **argv, const std::string &mode) {
  using namespace graphite;
  (void)hipSetDevice(0);
  std::ifstream file(argv[1]);
  size_t nc = 0, np = 0, no = 0;
  file >> nc >> np >> no;
  std::vector<size_t> ci(no), pi(no);
  std::vector<Pixel<FP>> ob(no);
  for (size_t i = 0; i < no; ++i) { double u, v; file >> ci[i] >> pi[i] >> u >> v; ob[i](0) = (FP)u; ob[i](1) = (FP)v; }
  managed_vector<Vec<FP, DC>> cams(nc);
  managed_vector<Vec<FP, 3>> pts(np);
  std::vector<double> focal(nc);
  for (size_t c = 0; c < nc; ++c)
    for (int k = 0; k < 9; ++k) { double x; file >> x; if (k < DC) cams[c](k) = (FP)x; if (k == 6) focal[c] = x; }
  for (size_t p = 0; p < np; ++p) for (int k = 0; k < 3; ++k) { double x; file >> x; pts[p](k) = (FP)x; }
  if (!file) { std::cerr << "bad BAL file" << std::endl; return 2; }

  Graph<FP, SP> graph;
  VecDescriptor<FP, SP, DC> cam_desc;
  VecDescriptor<FP, SP, 3> pt_desc;
  cam_desc.reserve(nc); pt_desc.reserve(np);
  graph.add_descriptor(&cam_desc);
  graph.add_descriptor(&pt_desc);
  for (size_t c = 0; c < nc; ++c) cam_desc.add_vertex(c, &cams[c]);
  for (size_t p = 0; p < np; ++p) pt_desc.add_vertex(nc + p, &pts[p]);
  pt_desc.set_eliminate(true);
  Factor<FP, SP> r_desc(&cam_desc, &pt_desc);
  r_desc.reserve(no);
  graph.add_descriptor(&r_desc);
  const std::string jmode = argc > 5 ? argv[5] : "stored";
  if (jmode == "dynamic") r_desc.set_jacobian_storage(false);
  std::vector<size_t> handles(no);
  using Loss = typename Factor<FP, SP>::LossType;
  using Data = typename Factor<FP, SP>::ConstraintDataType;
  for (size_t i = 0; i < no; ++i) {
    SP P[4] = {SP(1), SP(0), SP(0), SP(1)};
    Loss loss{};
    Data data{};
    if (mode == "weighted") { double Pd[4]; information(i, Pd); for (int k = 0; k < 4; ++k) P[k] = (SP)Pd[k]; }
    if constexpr (std::is_same<Loss, HuberLoss<FP, 2>>::value) loss = Loss((FP)huber_delta(i));
    if constexpr (std::is_same<Data, Radial6<FP>>::value) data.k3 = (FP)k3_of(i);
    if constexpr (std::is_same<Data, Intrinsics<FP>>::value) {
      // the BAL observation is -f d p: as a pinhole pixel with the camera looking down -z it is f p' with p' = (X / Z, Y / Z) = -p
      data.fx = (FP)focal[ci[i]]; data.fy = (FP)focal[ci[i]]; data.cx = FP(0); data.cy = FP(0);
    }
    handles[i] = r_desc.add_factor({ci[i], nc + pi[i]}, ob[i], mode == "weighted" ? P : nullptr, data, loss);
  }
  // "masked": one fixed pose (VertexDescriptor::set_fixed), every 97th factor deactivated (set_active, factor.hpp:419-431) and the
  // last landmark without an active factor (an unused vertex, active.hpp:18-21): the engine problem is built from the ACTIVE factors
  // and the USED vertices only, the streams of the user-traits kernels follow that compaction
  if (argc > 7 && std::string(argv[7]) == "masked") {
    cam_desc.set_fixed(0, true);
    for (size_t i = 0; i < no; ++i) if (i % 97 == 5 || pi[i] == np - 1) r_desc.set_active(handles[i], 1);
  }
  const std::string kind = argv[2];
  BlockJacobiPreconditioner<FP, SP> bj;
  IdentityPreconditioner<FP, SP> id;
  BlockJacobiSchurPreconditioner<FP, SP> bjs;
  std::unique_ptr<Solver<FP, SP>> solver;
  if (kind == "pcg") solver.reset(new PCGSolver<FP, SP>(10, 1.0, 5.0, &bj));
  else if (kind == "pcg-identity") solver.reset(new PCGSolver<FP, SP>(10, 1.0, 5.0, &id));
  else if (kind == "pcg-schur") solver.reset(new PCGSchurSolver<FP, SP>(10, 1.0, 5.0, &bjs));
  else if (kind == "eigen-schur") solver.reset(new EigenSchurLDLTSolver<FP, SP>());
  else return 2;
  StreamPool streams(2);
  optimizer::LevenbergMarquardtOptions<FP, SP> options;
  options.solver = solver.get();
  options.initial_damping = 1e-4;
  options.iterations = std::stoul(argv[3]);
  options.verbose = true;
  options.streams = &streams;
  const auto t0 = std::chrono::steady_clock::now();
  const bool ok = optimizer::levenberg_marquardt<FP, SP>(&graph, &options);
  const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::cout << std::setprecision(17) << "FINAL_CHI2 " << graph.chi2() << std::endl;
  std::cout << "CAM0";
  for (int k = 0; k < DC; ++k) std::cout << " " << cams[0](k);
  std::cout << std::endl << "PT0 " << pts[0](0) << " " << pts[0](1) << " " << pts[0](2) << std::endl;
  std::cout << "ENGINE_HANDOVERS " << optimizer::engine_handover_count() << std::endl
            << "ENGINE_MODEL_HANDOVERS " << optimizer::engine_model_handover_count() << std::endl
            << "TOTAL_SECONDS " << total << " SETUP_SECONDS " << optimizer::engine_last_setup_seconds() << " LOOP_SECONDS " << optimizer::engine_last_loop_seconds()
            << " ITERATIONS " << optimizer::engine_last_iterations() << std::endl
            << (ok ? "OK" : "STOPPED") << std::endl;
  if (argc > 7 && std::string(argv[7]) == "twice") { // second call on the unchanged structure: cached problem, loop time only
    options.verbose = false;
    const auto t1 = std::chrono::steady_clock::now();
    (void)optimizer::levenberg_marquardt<FP, SP>(&graph, &options);
    const double second = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    std::cout << "SECOND_CALL_SECONDS " << second << " SETUP_SECONDS " << optimizer::engine_last_setup_seconds() << " CACHE_HITS " << optimizer::engine_cache_hit_count()
              << " LOOP_SECONDS " << optimizer::engine_last_loop_seconds() << " ITERATIONS " << optimizer::engine_last_iterations() << std::endl;
  }
  solver.reset();
  return 0;
}

template <typename FP, typename SP> static int dispatch(int argc, char **argv, const std::string &mode) {
  if (mode == "bal") return run<FP, SP, graphite::BalFactor, 9>(argc, argv, mode);
  if (mode == "weighted") return run<FP, SP, graphite::WeightedFactor, 9>(argc, argv, mode);
  if (mode == "k3") return run<FP, SP, graphite::K3Factor, 9>(argc, argv, mode);
  if (mode == "pinhole") return run<FP, SP, graphite::PinholeFactor, 6>(argc, argv, mode);
  return 2;
}

int main(int argc, char **argv) {
  if (argc < 5) { std::cerr << "usage: test_engine_model <file> <solver> <iterations> <bal|weighted|k3|pinhole> [stored|dynamic] [fp64|fp32|mixed] [twice]" << std::endl; return 2; }
  const std::string mode = argv[4], prec = argc > 6 ? argv[6] : "fp64";
  if (mode == "slam2d") return run_slam2d<double, double, graphite::RangeBearingFactor, 2>(argc, argv);
  if (mode == "slam2d-range") return run_slam2d<double, double, graphite::RangeFactor, 1>(argc, argv);
  // fp32 / mixed precision: the weighted BAL factor only (every instantiation is a set of kernels to compile)
  if (prec == "fp32" && mode == "weighted") return run<float, float, graphite::WeightedFactor, 9>(argc, argv, mode);
  if (prec == "mixed" && mode == "weighted") return run<double, float, graphite::WeightedFactor, 9>(argc, argv, mode);
  if (prec != "fp64") return 2;
  return dispatch<double, double>(argc, argv, mode);
}
