// C++ mirror tests, shaped like the reference's gtest cases for this path (tests/schur.cu:242-389):
// same fixture (2 cameras x 3 points x 6 observations), same solver call sequence
// (update_structure / update_values / set_damping_factor / solve), relational assertions.
// No gtest in this image: a tiny CHECK macro, exit code != 0 on failure.
#include "graphite_mi355x.hpp"
#include <cmath>
#include <cstdio>

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

using namespace graphite;

template <typename T> struct Fixture { std::vector<T> cameras, points, obs; std::vector<int32_t> ci, pi; };

// tests/schur.cu:35-79
template <typename T> Fixture<T> two_camera_three_point() {
  Fixture<T> f;
  f.cameras = {T(0.12), T(-0.08), T(0.03), T(0.25), T(-0.10), T(0.20), T(800.0), T(0.01), T(-0.001),
               T(-0.09), T(0.06), T(-0.04), T(-0.30), T(0.14), T(-0.22), T(820.0), T(-0.012), T(0.0009)};
  f.points = {T(0.1f), T(0.0f), T(2.0f), T(-0.1f), T(0.05f), T(2.2f), T(0.0f), T(-0.05f), T(1.8f)};
  f.obs.assign(12, T(0));
  f.ci = {0, 1, 0, 1, 0, 1};
  f.pi = {0, 0, 1, 1, 2, 2};
  return f;
}

static void pcg_vs_pcg_schur() {
  using T = double;
  auto f = two_camera_three_point<T>();
  BalGraph<T> graph(f.cameras, f.points, f.obs, f.ci, f.pi);
  CHECK(graph.initialize_optimization(0));
  CHECK(graph.build_structure());
  CHECK(graph.get_hessian_dimension() == 27);
  CHECK(graph.get_elimination_block_column() == 2);
  StreamPool streams(2);
  graph.linearize(streams);
  CHECK(graph.chi2() != 0.0);                                    // schur.cu:144
  BlockJacobiSchurPreconditioner<T> sp;
  BlockJacobiPreconditioner<T> bp;
  PCGSchurSolver<T> schur_solver(512, 1e-14, 1e6, &sp);         // schur.cu:362-363
  PCGSolver<T> full_solver(2000, 1e-30, 1e12, &bp);
  schur_solver.update_structure(&graph, streams);
  schur_solver.update_values(&graph, streams);
  const T damping = 1e-4;
  schur_solver.set_damping_factor(&graph, damping, false, streams);
  std::vector<T> dx_schur(27), dx_full(27);
  CHECK(schur_solver.solve(&graph, dx_schur.data(), streams));
  full_solver.update_structure(&graph, streams);
  full_solver.update_values(&graph, streams);
  full_solver.set_damping_factor(&graph, damping, false, streams);
  CHECK(full_solver.solve(&graph, dx_full.data(), streams));
  double worst = 0;
  for (int i = 0; i < 27; ++i) worst = std::max(worst, std::fabs(dx_full[i] - dx_schur[i]));
  std::printf("max |dx_pcg - dx_pcg_schur| = %.3e\n", worst);
  CHECK(worst < 5e-4);                                           // schur.cu:385-388 tolerance
  // b_S and S are retrievable in the reference's layouts
  CHECK(graph.get(GR_GET_B_SCHUR).size() == 18);
  CHECK(graph.get(GR_GET_S).size() == 3 * 81);
}

template <typename T> static void lm_runs(gr_solver which) {
  auto f = two_camera_three_point<T>();
  for (size_t i = 0; i < f.obs.size(); ++i) f.obs[i] = T(3.0) * ((i % 2) ? 1 : -1);   // non-zero targets
  BalGraph<T> graph(f.cameras, f.points, f.obs, f.ci, f.pi);
  StreamPool streams(2);
  BlockJacobiSchurPreconditioner<T> sp;
  BlockJacobiPreconditioner<T> bp;
  PCGSchurSolver<T> s1(20, T(1e-12), T(1e6), &sp);
  PCGSolver<T> s2(50, T(1e-12), T(1e6), &bp);
  optimizer::LevenbergMarquardtOptions<T> opt;
  CHECK(!opt.validate());                                        // solver & streams null (levenberg_marquardt.hpp:80-97)
  opt.solver = which == GR_SOLVER_PCG_SCHUR ? static_cast<Solver<T> *>(&s1) : static_cast<Solver<T> *>(&s2);
  opt.streams = &streams;
  opt.iterations = 15;
  const T chi2_0 = graph.chi2();
  gr_lm_stats st{};
  CHECK((optimizer::levenberg_marquardt<T, T>(&graph, &opt, &st)));
  const T chi2_1 = graph.chi2();
  std::printf("LM (%s, %d): chi2 %.6g -> %.6g in %d iterations\n", sizeof(T) == 8 ? "f64" : "f32", (int)which, (double)chi2_0, (double)chi2_1, st.iterations_run);
  CHECK(chi2_1 < chi2_0 * 1e-2);
  std::vector<T> c, p;
  graph.read_back(c, p);
  CHECK(c.size() == 18 && p.size() == 9);
  bool moved = false;
  for (size_t i = 0; i < c.size(); ++i) moved |= (c[i] != f.cameras[i]);
  CHECK(moved);
}

// VertexDescriptor::set_fixed through the mirror: the fixed camera and point keep their values, the others move
static void fixed_vertices_stay() {
  using T = double;
  auto f = two_camera_three_point<T>();
  for (size_t i = 0; i < f.obs.size(); ++i) f.obs[i] = T(3.0) * ((i % 2) ? 1 : -1);
  BalGraph<T> graph(f.cameras, f.points, f.obs, f.ci, f.pi);
  const unsigned char cam_fixed[2] = {1, 0}, pt_fixed[3] = {0, 0, 1};
  graph.set_fixed(cam_fixed, pt_fixed);
  StreamPool streams(2);
  BlockJacobiPreconditioner<T> bp;
  PCGSolver<T> s2(50, T(1e-12), T(1e6), &bp);
  optimizer::LevenbergMarquardtOptions<T> opt;
  opt.solver = &s2;
  opt.streams = &streams;
  opt.iterations = 10;
  const T chi2_0 = graph.chi2();
  gr_lm_stats st{};
  CHECK((optimizer::levenberg_marquardt<T, T>(&graph, &opt, &st)));
  CHECK(graph.chi2() < chi2_0);
  std::vector<T> c, p;
  graph.read_back(c, p);
  bool cam0_same = true, cam1_moved = false, pt2_same = true, pt0_moved = false;
  for (int i = 0; i < 9; ++i) { cam0_same &= c[i] == f.cameras[i]; cam1_moved |= c[9 + i] != f.cameras[9 + i]; }
  for (int i = 0; i < 3; ++i) { pt2_same &= p[6 + i] == f.points[6 + i]; pt0_moved |= p[i] != f.points[i]; }
  CHECK(cam0_same && pt2_same && cam1_moved && pt0_moved);
}

static void backup_and_revert() {
  using T = double;
  auto f = two_camera_three_point<T>();
  BalGraph<T> graph(f.cameras, f.points, f.obs, f.ci, f.pi);
  graph.linearize();
  std::vector<T> dx(27, 1e-3);
  const T chi2_0 = graph.chi2();
  graph.backup_parameters();
  graph.apply_update(dx.data());
  CHECK(graph.chi2() != chi2_0);
  graph.revert_parameters();
  CHECK(graph.chi2() == chi2_0);
}

static void bad_inputs_fail_loudly() {
  using T = double;
  auto f = two_camera_three_point<T>();
  f.ci[1] = 0; // duplicate (camera 0, point 0)
  bool threw = false;
  try { BalGraph<T> g(f.cameras, f.points, f.obs, f.ci, f.pi); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);
  threw = false;
  try { PCGSolver<T> s(10, 1.0, 5.0, nullptr); } catch (const std::invalid_argument &) { threw = true; }
  CHECK(threw);
}

int main() {
  pcg_vs_pcg_schur();
  lm_runs<double>(GR_SOLVER_PCG_SCHUR);
  lm_runs<double>(GR_SOLVER_PCG);
  lm_runs<float>(GR_SOLVER_PCG_SCHUR);
  fixed_vertices_stay();
  backup_and_revert();
  bad_inputs_fail_loudly();
  std::printf("%s (%d failures)\n", failures ? "FAILED" : "OK", failures);
  return failures ? 1 : 0;
}
