// Hessian<T,S> / SchurComplement<T,S> / CSCMatrix<S,I> driven the way the reference's own test drives them
// (tests/schur.cu:113-240, SchurTests.BALTwoCamerasThreePoints): build_structure, linearize, update_values, CSC
// export, then S, b_S and the landmark back-substitution against a CPU statement computed from the exported
// Hessian (the role of tests/schur_cpu_ref.cpp:8-51), all at the reference's 1e-12.
//   test_sparse_schur schur <bal file>        the flow above + a dump of H in the reference's value layout
//   test_sparse_schur handles <bal file>      factor handles stay valid across remove_factor / add_factor (factor.hpp:308-412)
//   test_sparse_schur slam <num poses>        a 2-D graph far beyond dense reach: sparse Schur PCG must converge
#include <cmath>
#include <fstream>
#include <graphite/hessian.hpp>
#include <graphite/optimizer/levenberg_marquardt.hpp>
#include <graphite/preconditioner/block_jacobi_schur.hpp>
#include <graphite/schur.hpp>
#include <graphite/solver/pcg_schur.hpp>
#include <iostream>
#include <random>
#include <string>

namespace graphite {
template <typename T, int N> struct Vec {
  T v[N];
  hd_fn T operator()(int i) const { return v[i]; }
  hd_fn T &operator()(int i) { return v[i]; }
};
template <typename T, int N> struct VecTraits {
  static constexpr size_t dimension = N;
  using Vertex = Vec<T, N>;
  template <typename P> d_fn static void parameters(const Vertex &x, P *p) { for (int i = 0; i < N; ++i) p[i] = P(x(i)); }
  d_fn static void update(Vertex &x, const T *d) { for (int i = 0; i < N; ++i) x(i) += d[i]; }
};
template <typename T, typename S> using CameraDescriptor = VertexDescriptor<T, S, VecTraits<T, 9>>;
template <typename T, typename S> using PointDescriptor = VertexDescriptor<T, S, VecTraits<T, 3>>;
template <typename T, typename S> using Pose2Descriptor = VertexDescriptor<T, S, VecTraits<T, 2>>;

template <typename D, typename T> d_fn void reprojection(const D *cam, const D *pt, const Vec<T, 2> &obs, D *err) {
  const D rx = cam[0], ry = cam[1], rz = cam[2];
  const D theta2 = rx * rx + ry * ry + rz * rz;
  D P[3];
  if (theta2 > D(T(0))) {
    const D theta = sqrt(theta2);
    const D ax = rx / theta, ay = ry / theta, az = rz / theta;
    const D s = sin(theta), c = cos(theta), k = D(T(1)) - c;
    const D R[9] = {k * ax * ax + c,      k * ax * ay - s * az, k * ax * az + s * ay,
                    k * ax * ay + s * az, k * ay * ay + c,      k * ay * az - s * ax,
                    k * ax * az - s * ay, k * ay * az + s * ax, k * az * az + c};
    for (int i = 0; i < 3; ++i) P[i] = R[3 * i] * pt[0] + R[3 * i + 1] * pt[1] + R[3 * i + 2] * pt[2] + cam[3 + i];
  } else {
    for (int i = 0; i < 3; ++i) P[i] = pt[i] + cam[3 + i];
  }
  const D px = -P[0] / P[2], py = -P[1] / P[2];
  const D r2 = px * px + py * py;
  const D d = D(T(1)) + cam[7] * r2 + cam[8] * r2 * r2;
  err[0] = cam[6] * d * px - D(obs(0));
  err[1] = cam[6] * d * py - D(obs(1));
}
template <typename T, typename S> struct ReprojectionErrorTraits {
  static constexpr size_t dimension = 2;
  using VertexDescriptors = std::tuple<CameraDescriptor<T, S>, PointDescriptor<T, S>>;
  using Observation = Vec<T, 2>;
  using Data = Empty;
  using Loss = DefaultLoss<T, dimension>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *camera, const D *point, const Observation &obs, D *error) { reprojection<D, T>(camera, point, obs, error); }
};
template <typename T, typename S> using ReprojectionError = FactorDescriptor<T, S, ReprojectionErrorTraits<T, S>>;

// 2-D relative-position factor between two 2-vectors (odometry between poses, or pose -> landmark)
template <typename T, typename S, typename DA, typename DB> struct RelativeTraits {
  static constexpr size_t dimension = 2;
  using VertexDescriptors = std::tuple<DA, DB>;
  using Observation = Vec<T, 2>;
  using Data = Empty;
  using Loss = DefaultLoss<T, dimension>;
  using Differentiation = DifferentiationMode::Auto;
  template <typename D> d_fn static void error(const D *a, const D *b, const Observation &obs, D *e) {
    e[0] = b[0] - a[0] - D(obs(0));
    e[1] = b[1] - a[1] - D(obs(1));
  }
};
} // namespace graphite

using namespace graphite;
using FP = double;

struct Bal {
  size_t nc = 0, np = 0, no = 0;
  std::vector<size_t> ci, pi;
  std::vector<Vec<FP, 2>> ob;
  managed_vector<Vec<FP, 9>> cams;
  managed_vector<Vec<FP, 3>> pts;
  bool read(const char *path) {
    std::ifstream file(path);
    file >> nc >> np >> no;
    ci.resize(no); pi.resize(no); ob.resize(no);
    for (size_t i = 0; i < no; ++i) file >> ci[i] >> pi[i] >> ob[i](0) >> ob[i](1);
    cams.resize(nc); pts.resize(np);
    for (size_t c = 0; c < nc; ++c) for (int k = 0; k < 9; ++k) file >> cams[c](k);
    for (size_t p = 0; p < np; ++p) for (int k = 0; k < 3; ++k) file >> pts[p](k);
    return (bool)file;
  }
};

static std::vector<double> dense_inverse(std::vector<double> A, int d) {
  std::vector<double> R(d * d, 0.0);
  for (int i = 0; i < d; ++i) R[i * d + i] = 1;
  for (int k = 0; k < d; ++k) {
    int piv = k;
    for (int r = k + 1; r < d; ++r) if (std::fabs(A[r * d + k]) > std::fabs(A[piv * d + k])) piv = r;
    for (int c = 0; c < d; ++c) { std::swap(A[k * d + c], A[piv * d + c]); std::swap(R[k * d + c], R[piv * d + c]); }
    const double ip = 1.0 / A[k * d + k];
    for (int c = 0; c < d; ++c) { A[k * d + c] *= ip; R[k * d + c] *= ip; }
    for (int r = 0; r < d; ++r) if (r != k) { const double f = A[r * d + k]; for (int c = 0; c < d; ++c) { A[r * d + c] -= f * A[k * d + c]; R[r * d + c] -= f * R[k * d + c]; } }
  }
  return R;
}

static int run_schur(const char *path) {
  using S = double;
  Bal bal;
  if (!bal.read(path)) { std::cerr << "bad BAL file" << std::endl; return 2; }
  Graph<FP, S> graph;
  CameraDescriptor<FP, S> camera_desc;
  PointDescriptor<FP, S> point_desc;
  ReprojectionError<FP, S> reproj_desc(&camera_desc, &point_desc);
  graph.add_vertex_descriptor(&camera_desc);
  graph.add_vertex_descriptor(&point_desc);
  graph.add_factor_descriptor(&reproj_desc);
  for (size_t c = 0; c < bal.nc; ++c) camera_desc.add_vertex(c, &bal.cams[c]);
  for (size_t p = 0; p < bal.np; ++p) point_desc.add_vertex(bal.nc + p, &bal.pts[p]);
  point_desc.set_eliminate(true);
  for (size_t i = 0; i < bal.no; ++i) reproj_desc.add_factor({bal.ci[i], bal.nc + bal.pi[i]}, bal.ob[i], nullptr, Empty(), DefaultLoss<FP, 2>());
  if (!graph.initialize_optimization(0)) return 3;

  // ---- tests/schur.cu:127-154, statement for statement --------------------------------------------------------
  StreamPool streams(2);
  Hessian<FP, S> H;
  SchurComplement<FP, S> schur(H);
  using I = int;
  CSCMatrix<S, I> d_H;
  CSCMatrix<S, I> d_Schur;
  graph.build_structure();
  H.build_structure(&graph, streams);
  schur.build_structure(&graph, streams);
  graph.linearize(streams);
  if (graph.chi2() == 0.0) { std::cerr << "chi2 is zero" << std::endl; return 4; }
  H.update_values(&graph, streams);
  schur.update_values(&graph, streams);
  H.build_csc_structure(&graph, d_H);
  schur.build_csc_structure(&graph, d_Schur);
  H.update_csc_values(&graph, d_H);
  schur.update_csc_values(&graph, d_Schur);

  // ---- CPU statement (tests/schur_cpu_ref.cpp:8-51) from the exported Hessian ----------------------------------
  const size_t n = graph.get_hessian_dimension(), pd = 9 * bal.nc, ld = n - pd;
  const auto hp = d_H.d_pointers.to_host(), hi = d_H.d_indices.to_host();
  const auto hv = d_H.d_values.to_host();
  std::vector<double> Hd(n * n, 0.0);
  for (size_t col = 0; col < n; ++col)
    for (int q = hp[col]; q < hp[col + 1]; ++q) { Hd[(size_t)hi[q] * n + col] = hv[q]; Hd[col * n + (size_t)hi[q]] = hv[q]; }
  std::vector<double> Hll(ld * ld), Hpl(pd * ld);
  for (size_t r = 0; r < ld; ++r) for (size_t c = 0; c < ld; ++c) Hll[r * ld + c] = Hd[(pd + r) * n + pd + c];
  for (size_t r = 0; r < pd; ++r) for (size_t c = 0; c < ld; ++c) Hpl[r * ld + c] = Hd[r * n + pd + c];
  std::vector<double> Hll_inv(ld * ld, 0.0); // block diagonal 3 x 3 inverses
  for (size_t l = 0; l < bal.np; ++l) {
    std::vector<double> blk(9);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) blk[r * 3 + c] = Hll[(3 * l + r) * ld + 3 * l + c];
    const auto inv = dense_inverse(blk, 3);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Hll_inv[(3 * l + r) * ld + 3 * l + c] = inv[r * 3 + c];
  }
  std::vector<double> W(pd * ld, 0.0), Sref(pd * pd, 0.0); // W = Hpl Hll^-1
  for (size_t r = 0; r < pd; ++r) for (size_t c = 0; c < ld; ++c) { double s = 0; for (size_t k = 0; k < ld; ++k) s += Hpl[r * ld + k] * Hll_inv[k * ld + c]; W[r * ld + c] = s; }
  for (size_t r = 0; r < pd; ++r) for (size_t c = 0; c < pd; ++c) { double s = 0; for (size_t k = 0; k < ld; ++k) s += W[r * ld + k] * Hpl[c * ld + k]; Sref[r * pd + c] = Hd[r * n + c] - s; }
  // S (upper) against the CPU result: Eigen isApprox(.., 1e-12) = ||a - b|| <= 1e-12 min(||a||, ||b||)
  const auto sp = d_Schur.d_pointers.to_host(), si = d_Schur.d_indices.to_host();
  const auto sv = d_Schur.d_values.to_host();
  double diff2 = 0, na2 = 0, nb2 = 0;
  size_t upper_nnz = 0;
  std::vector<char> seen(pd * pd, 0);
  for (size_t col = 0; col < pd; ++col)
    for (int q = sp[col]; q < sp[col + 1]; ++q) {
      const size_t row = (size_t)si[q];
      if (row > col) { std::cerr << "S csc holds a lower entry" << std::endl; return 5; }
      seen[row * pd + col] = 1; ++upper_nnz;
      const double a = sv[q], b = Sref[row * pd + col];
      diff2 += (a - b) * (a - b); na2 += a * a; nb2 += b * b;
    }
  for (size_t r = 0; r < pd; ++r) for (size_t c = r; c < pd; ++c) if (!seen[r * pd + c]) { diff2 += Sref[r * pd + c] * Sref[r * pd + c]; nb2 += Sref[r * pd + c] * Sref[r * pd + c]; }
  const double s_rel = std::sqrt(diff2) / std::sqrt(std::min(na2, nb2));
  std::cout << std::setprecision(17) << "S_REL " << s_rel << " NNZ " << upper_nnz << std::endl;
  // b_S (tests/schur.cu:182-208)
  std::vector<FP> b(n);
  for (size_t i = 0; i < n; ++i) b[i] = graph.get_b()[i];
  const auto bS = schur.get_b_Schur().to_host();
  double bs_err = 0;
  for (size_t r = 0; r < pd; ++r) { double s = b[r]; for (size_t k = 0; k < ld; ++k) s -= W[r * ld + k] * b[pd + k]; bs_err = std::max(bs_err, std::fabs(s - bS[r])); }
  std::cout << "BSCHUR_ABS " << bs_err << std::endl;
  // back-substitution with dx_p[i] = 0.01 (i + 1) (tests/schur.cu:210-239)
  std::vector<FP> dx_p(pd);
  for (size_t i = 0; i < pd; ++i) dx_p[i] = 0.01 * (double)(i + 1);
  device_vector<FP> d_dx_p, d_dx_l(ld);
  d_dx_p = dx_p;
  schur.compute_landmark_update(&graph, streams, d_dx_l.data().get(), d_dx_p.data().get());
  const auto dx_l = d_dx_l.to_host();
  double bk_err = 0;
  for (size_t r = 0; r < ld; ++r) {
    double s = 0;
    for (size_t c = 0; c < ld; ++c) { double rhs = b[pd + c]; for (size_t k = 0; k < pd; ++k) rhs -= Hpl[k * ld + c] * dx_p[k]; s += Hll_inv[r * ld + c] * rhs; }
    bk_err = std::max(bk_err, std::fabs(s - dx_l[r]));
  }
  std::cout << "BACKSUB_ABS " << bk_err << std::endl;
  // S x through the operator against the dense reference
  device_vector<FP> d_y(pd);
  schur.execute_schur_vector_multiply(&graph, streams, d_y.data().get(), d_dx_p.data().get());
  const auto y = d_y.to_host();
  double mv_err = 0, mv_max = 0;
  for (size_t r = 0; r < pd; ++r) { double s = 0; for (size_t c = 0; c < pd; ++c) s += Sref[r * pd + c] * dx_p[c]; mv_err = std::max(mv_err, std::fabs(s - y[r])); mv_max = std::max(mv_max, std::fabs(s)); }
  std::cout << "MATVEC_REL " << mv_err / mv_max << std::endl;
  // the Hessian in the reference's value layout (block-CSC upper, blocks column-major, diagonal block last)
  const auto hcp = H.get_block_col_pointers().to_host(), hri = H.get_block_row_indices().to_host(), hof = H.get_block_value_offsets().to_host();
  const auto hval = H.get_values().to_host();
  std::cout << "H_COLPTR"; for (auto v : hcp) std::cout << " " << v; std::cout << std::endl;
  std::cout << "H_ROWIDX"; for (auto v : hri) std::cout << " " << v; std::cout << std::endl;
  std::cout << "H_OFFSETS"; for (auto v : hof) std::cout << " " << v; std::cout << std::endl;
  std::cout << "H_VALUES"; for (auto v : hval) std::cout << " " << v; std::cout << std::endl;
  std::cout << "H_CSC_P"; for (auto v : hp) std::cout << " " << v; std::cout << std::endl;
  std::cout << "H_CSC_I"; for (auto v : hi) std::cout << " " << v; std::cout << std::endl;
  std::cout << "H_CSC_X"; for (auto v : hv) std::cout << " " << v; std::cout << std::endl;
  std::cout << "OK" << std::endl;
  return 0;
}

// factor ids are stable handles: removing factor k does not renumber any other factor; a released id is re-used
static int run_handles(const char *path) {
  using S = double;
  Bal bal;
  if (!bal.read(path)) return 2;
  Graph<FP, S> graph;
  CameraDescriptor<FP, S> cam_desc;
  PointDescriptor<FP, S> pt_desc;
  ReprojectionError<FP, S> r_desc(&cam_desc, &pt_desc);
  graph.add_descriptor(&cam_desc); graph.add_descriptor(&pt_desc); graph.add_descriptor(&r_desc);
  for (size_t c = 0; c < bal.nc; ++c) cam_desc.add_vertex(c, &bal.cams[c]);
  for (size_t p = 0; p < bal.np; ++p) pt_desc.add_vertex(bal.nc + p, &bal.pts[p]);
  pt_desc.set_eliminate(true);
  std::vector<size_t> ids(bal.no);
  for (size_t i = 0; i < bal.no; ++i) ids[i] = r_desc.add_factor({bal.ci[i], bal.nc + bal.pi[i]}, bal.ob[i], nullptr, Empty(), DefaultLoss<FP, 2>());
  bool ok = true;
  for (size_t i = 0; i < bal.no; ++i) ok &= ids[i] == i; // a fresh descriptor counts up (utils.hpp:88-96)
  // remove every third factor; the handles of all the others must still name the same factor
  std::vector<size_t> removed;
  for (size_t i = 0; i < bal.no; i += 3) { r_desc.remove_factor(ids[i]); removed.push_back(i); }
  for (size_t i = 0; i < bal.no; ++i) {
    if (i % 3 == 0) continue;
    const auto v = r_desc.get_vertex_ids(ids[i]);
    ok &= v[0] == bal.ci[i] && v[1] == bal.nc + bal.pi[i];
    ok &= r_desc.get_observation(ids[i])(0) == bal.ob[i](0) && r_desc.get_observation(ids[i])(1) == bal.ob[i](1);
  }
  ok &= r_desc.internal_count() == bal.no - removed.size();
  bool threw = false;
  try { (void)r_desc.get_vertex_ids(ids[0]); } catch (const std::out_of_range &) { threw = true; } // factor.hpp:460 (.at)
  ok &= threw;
  r_desc.remove_factor(ids[0]); // unknown id: message + no-op (factor.hpp:309-312)
  ok &= r_desc.internal_count() == bal.no - removed.size();
  // add them back: released handles are re-used (LIFO), the graph is the original one again
  std::vector<size_t> reused;
  for (size_t i : removed) { ids[i] = r_desc.add_factor({bal.ci[i], bal.nc + bal.pi[i]}, bal.ob[i], nullptr, Empty(), DefaultLoss<FP, 2>()); reused.push_back(ids[i]); }
  std::sort(reused.begin(), reused.end());
  for (size_t k = 0; k < removed.size(); ++k) ok &= reused[k] == removed[k];
  ok &= r_desc.internal_count() == bal.no;
  std::cout << "HANDLES " << (ok ? "OK" : "BAD") << std::endl;
  // remove -> add -> optimise
  BlockJacobiSchurPreconditioner<FP, S> bjs;
  PCGSchurSolver<FP, S> solver(10, 1.0, 5.0, &bjs);
  StreamPool streams(2);
  optimizer::LevenbergMarquardtOptions<FP, S> options;
  options.solver = &solver; options.initial_damping = 1e-4; options.iterations = 6; options.verbose = true; options.streams = &streams;
  const bool run = optimizer::levenberg_marquardt<FP, S>(&graph, &options);
  std::cout << std::setprecision(17) << "FINAL_CHI2 " << graph.chi2() << std::endl << (run && ok ? "OK" : "STOPPED") << std::endl;
  return ok ? 0 : 1;
}

// a long 2-D trajectory with landmarks: n poses, 4 landmarks per pose seen from 3 consecutive poses each.
// Dense H would need (2 n + 8 n)^2 scalars (n = 50 000: 2 TB); the block-sparse path needs O(n).
static int run_slam(size_t n_poses) {
  using S = double;
  using PoseD = Pose2Descriptor<FP, S>;
  using OdoTraits = RelativeTraits<FP, S, PoseD, PoseD>;
  using LmkTraits = RelativeTraits<FP, S, PoseD, PoseD>;
  const size_t lpp = 4, n_lmk = n_poses * lpp;
  managed_vector<Vec<FP, 2>> poses(n_poses), lmks(n_lmk);
  std::mt19937_64 rng(7);
  std::normal_distribution<double> noise(0.0, 0.02), big(0.0, 0.5);
  std::vector<Vec<FP, 2>> true_pose(n_poses), true_lmk(n_lmk);
  for (size_t i = 0; i < n_poses; ++i) { true_pose[i](0) = 0.5 * (double)i; true_pose[i](1) = 3.0 * std::sin(0.01 * (double)i); }
  for (size_t l = 0; l < n_lmk; ++l) { const size_t i = l / lpp; true_lmk[l](0) = true_pose[i](0) + big(rng); true_lmk[l](1) = true_pose[i](1) + 2.0 + big(rng); }
  for (size_t i = 0; i < n_poses; ++i) for (int k = 0; k < 2; ++k) poses[i](k) = true_pose[i](k) + (i ? 0.3 * big(rng) : 0.0);
  for (size_t l = 0; l < n_lmk; ++l) for (int k = 0; k < 2; ++k) lmks[l](k) = true_lmk[l](k) + 0.3 * big(rng);
  Graph<FP, S> graph;
  PoseD pose_desc, lmk_desc;
  FactorDescriptor<FP, S, OdoTraits> odo(&pose_desc, &pose_desc);
  FactorDescriptor<FP, S, LmkTraits> obs(&pose_desc, &lmk_desc);
  graph.add_descriptor(&pose_desc); graph.add_descriptor(&lmk_desc); graph.add_descriptor(&odo); graph.add_descriptor(&obs);
  pose_desc.reserve(n_poses); lmk_desc.reserve(n_lmk);
  for (size_t i = 0; i < n_poses; ++i) pose_desc.add_vertex(i, &poses[i], i == 0); // the first pose is fixed (gauge)
  for (size_t l = 0; l < n_lmk; ++l) lmk_desc.add_vertex(n_poses + l, &lmks[l]);
  lmk_desc.set_eliminate(true);
  size_t nf = 0;
  for (size_t i = 0; i + 1 < n_poses; ++i) {
    Vec<FP, 2> z; for (int k = 0; k < 2; ++k) z(k) = true_pose[i + 1](k) - true_pose[i](k) + noise(rng);
    odo.add_factor({i, i + 1}, z, nullptr, Empty(), DefaultLoss<FP, 2>()); ++nf;
  }
  for (size_t l = 0; l < n_lmk; ++l)
    for (size_t d = 0; d < 3; ++d) {
      const size_t i = l / lpp + d;
      if (i >= n_poses) continue;
      Vec<FP, 2> z; for (int k = 0; k < 2; ++k) z(k) = true_lmk[l](k) - true_pose[i](k) + noise(rng);
      obs.add_factor({i, n_poses + l}, z, nullptr, Empty(), DefaultLoss<FP, 2>()); ++nf;
    }
  BlockJacobiSchurPreconditioner<FP, S> bjs;
  PCGSchurSolver<FP, S> solver(200, 1e-10, 1e12, &bjs);
  StreamPool streams(2);
  optimizer::LevenbergMarquardtOptions<FP, S> options;
  options.solver = &solver; options.initial_damping = 1e-6; options.iterations = 6; options.verbose = false; options.streams = &streams;
  graph.initialize_optimization(0);
  graph.linearize(streams);
  const double chi0 = graph.chi2();
  const bool run = optimizer::levenberg_marquardt<FP, S>(&graph, &options);
  const double chi1 = graph.chi2();
  double err = 0;
  for (size_t i = 0; i < n_poses; i += 97) err = std::max(err, std::hypot(poses[i](0) - true_pose[i](0), poses[i](1) - true_pose[i](1)));
  std::cout << std::setprecision(9) << "SLAM vertices " << n_poses + n_lmk << " factors " << nf << " hessian_dim " << graph.get_hessian_dimension() << " chi2 " << chi0 << " -> " << chi1
            << " per_factor " << chi1 / (double)nf << " max_pose_error " << err << std::endl << (run ? "OK" : "STOPPED") << std::endl;
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 3) { std::cerr << "usage: test_sparse_schur schur|handles <bal file> | slam <poses>" << std::endl; return 2; }
  (void)hipSetDevice(0);
  const std::string mode = argv[1];
  if (mode == "schur") return run_schur(argv[2]);
  if (mode == "handles") return run_handles(argv[2]);
  if (mode == "slam") return run_slam(std::stoul(argv[2]));
  return 2;
}
