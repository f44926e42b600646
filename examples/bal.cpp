// bal.cpp — the BAL driver of the reference (examples/bal.cu:43-360) on the MI355X library.
// Same file format, same options (--lambda --iterations --verbose --pcg_iterations --pcg_tolerance
// --rejection_ratio --precision {FP64-FP64,FP64-FP32,FP32-FP32} --solver {pcg,pcg-schur,pcg-schur-implicit,eigen,eigen-schur,cudss,
// cudss-schur} --identity_damping --hybrid_memory),
// same printed summary (MSE / Half MSE).  Host-only C++17: g++ -Iinclude examples/bal.cpp
// -Lgraphite_amd -lgraphite_mi355x -Wl,-rpath,$PWD/graphite_amd -o bal
#include "graphite_mi355x.hpp"
#include <charconv>
#include <chrono>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>

struct Args {
  std::string file, precision = "FP64-FP64", solver = "pcg";
  double lambda = 1.0e-4, pcg_tolerance = 1.0, rejection_ratio = 5.0;
  size_t iterations = 50, pcg_iterations = 10;
  bool verbose = false, identity_damping = false;
};

static Args parse(int argc, char **argv) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    const std::string s = argv[i];
    auto next = [&]() -> std::string { if (i + 1 >= argc) throw std::runtime_error("missing value for " + s); return argv[++i]; };
    if (s == "--lambda") a.lambda = std::stod(next());
    else if (s == "--iterations") a.iterations = std::stoul(next());
    else if (s == "--verbose") a.verbose = true;
    else if (s == "--pcg_iterations") a.pcg_iterations = std::stoul(next());
    else if (s == "--pcg_tolerance") a.pcg_tolerance = std::stod(next());
    else if (s == "--rejection_ratio") a.rejection_ratio = std::stod(next());
    else if (s == "--precision") a.precision = next();
    else if (s == "--solver") a.solver = next();
    else if (s == "--identity_damping") a.identity_damping = true;
    else if (s == "--hybrid_memory") (void)next(); // cuDSS option of the reference: accepted, unused
    else if (s.rfind("--", 0) == 0) throw std::runtime_error("unknown option " + s);
    else a.file = s;
  }
  if (a.file.empty()) throw std::runtime_error("usage: bal <file> [--lambda 1e-4] [--iterations 50] [--verbose] [--pcg_iterations 10] "
                                               "[--pcg_tolerance 1.0] [--rejection_ratio 5.0] [--precision FP64-FP64|FP64-FP32|FP32-FP32] "
                                               "[--solver pcg|pcg-schur|pcg-schur-implicit|eigen|eigen-schur|cudss|cudss-schur] [--identity_damping] [--hybrid_memory MB]");
  return a;
}

template <typename FP, typename SP = FP> void bundle_adjustment(const Args &a) {
  using namespace graphite;
  std::cout << "Running bundle adjustment with graph precision = " << (sizeof(FP) == 8 ? "double" : "float")
            << " and solver precision = " << (sizeof(SP) == 8 ? "double" : "float") << std::endl;
  // The reference reads the file with operator>> and one managed-memory push_back per line
  // (bal.cu:96-109); at Final-13682 scale (29 M observation lines) that is minutes of host time.
  // Here: one read of the whole file, std::from_chars over the buffer.
  std::ifstream file(a.file, std::ios::binary | std::ios::ate);
  if (!file.is_open()) { std::cerr << "Error: Unable to open file " << a.file << std::endl; throw std::runtime_error("File open error"); }
  auto start = std::chrono::steady_clock::now();
  std::string buf((size_t)file.tellg(), '\0');
  file.seekg(0);
  file.read(buf.data(), (std::streamsize)buf.size());
  const char *cur = buf.data(), *end = buf.data() + buf.size();
  auto skip = [&] { while (cur < end && (*cur == ' ' || *cur == '\n' || *cur == '\r' || *cur == '\t')) ++cur; };
  auto next_u = [&]() -> size_t {
    skip();
    size_t v = 0;
    auto r = std::from_chars(cur, end, v);
    if (r.ec != std::errc()) throw std::runtime_error("truncated BAL file");
    cur = r.ptr;
    return v;
  };
  auto next_f = [&]() -> double {
    skip();
    if (cur < end && *cur == '+') ++cur;
    double v = 0;
    auto r = std::from_chars(cur, end, v);
    if (r.ec != std::errc()) throw std::runtime_error("truncated BAL file");
    cur = r.ptr;
    return v;
  };
  const size_t num_cameras = next_u(), num_points = next_u(), num_observations = next_u();
  std::cout << "Number of cameras: " << num_cameras << std::endl;
  std::cout << "Number of points: " << num_points << std::endl;
  std::cout << "Number of observations: " << num_observations << std::endl;
  std::vector<int32_t> cam_idx(num_observations), pt_idx(num_observations);
  std::vector<FP> obs(2 * num_observations), cameras(9 * num_cameras), points(3 * num_points);
  for (size_t i = 0; i < num_observations; ++i) {
    cam_idx[i] = (int32_t)next_u(); pt_idx[i] = (int32_t)next_u();
    obs[2 * i] = (FP)next_f(); obs[2 * i + 1] = (FP)next_f();
  }
  for (auto &v : cameras) v = (FP)next_f();
  for (auto &v : points) v = (FP)next_f();
  std::cout << "Reading the problem took " << std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() << " seconds." << std::endl;

  start = std::chrono::steady_clock::now();
  BalGraph<FP> graph(cameras, points, obs, cam_idx, pt_idx);
  if (sizeof(SP) < sizeof(FP)) graph.set_jacobian_precision_f32(true);
  std::cout << "Graph built with " << num_cameras << " cameras, " << num_points << " points, and " << num_observations
            << " observations (" << std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() << " s)." << std::endl;

  BlockJacobiPreconditioner<FP> preconditioner;
  BlockJacobiSchurPreconditioner<FP> schur_preconditioner;
  std::unique_ptr<Solver<FP>> solver_ptr;
  if (a.solver == "pcg") {
    std::cout << "Using PCG solver." << std::endl;
    solver_ptr = std::make_unique<PCGSolver<FP>>(a.pcg_iterations, (FP)a.pcg_tolerance, (FP)a.rejection_ratio, &preconditioner);
  } else if (a.solver == "pcg-schur") {
    std::cout << "Using PCG Schur solver." << std::endl;
    solver_ptr = std::make_unique<PCGSchurSolver<FP>>(a.pcg_iterations, (FP)a.pcg_tolerance, (FP)a.rejection_ratio, &schur_preconditioner);
  } else if (a.solver == "pcg-schur-implicit") {
    std::cout << "Using PCG Schur solver (implicit Schur complement)." << std::endl;
    solver_ptr = std::make_unique<PCGImplicitSchurSolver<FP>>(a.pcg_iterations, (FP)a.pcg_tolerance, (FP)a.rejection_ratio, &schur_preconditioner);
  } else if (a.solver == "eigen-schur") {
    std::cout << "Using Eigen Schur LDLT solver." << std::endl; // the reference's line; here: dense MFMA Cholesky of S
    solver_ptr = std::make_unique<EigenSchurLDLTSolver<FP>>();
  } else if (a.solver == "cudss-schur") {
    std::cout << "Using cuDSS Schur solver." << std::endl;
    solver_ptr = std::make_unique<cudssSchurSolver<FP>>();
  } else if (a.solver == "eigen") {
    // the reference factorises the full damped H (solver/eigen.hpp:49-98); eliminating the points first and factorising S
    // gives the same step (block Gaussian elimination of the same system), which is what the engine's direct solver does
    std::cout << "Using Eigen LDLT solver." << std::endl;
    solver_ptr = std::make_unique<EigenLDLTSolver<FP>>();
  } else if (a.solver == "cudss") {
    std::cout << "Using cuDSS solver." << std::endl;
    solver_ptr = std::make_unique<cudssSolver<FP>>();
  } else throw std::runtime_error("Unsupported solver option (pcg | pcg-schur | pcg-schur-implicit | eigen | eigen-schur | cudss | cudss-schur)");

  std::cout << "Optimizing!" << std::endl;
  StreamPool streams(8);
  optimizer::LevenbergMarquardtOptions<FP> options;
  options.solver = solver_ptr.get();
  options.initial_damping = a.lambda;
  options.iterations = a.iterations;
  options.optimization_level = 0;
  options.verbose = a.verbose;
  options.streams = &streams;
  options.use_identity = a.identity_damping;
  start = std::chrono::steady_clock::now();
  gr_lm_stats st{};
  optimizer::levenberg_marquardt<FP, FP>(&graph, &options, &st);
  std::cout << "Optimization took " << std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() << " seconds ("
            << st.iterations_run << " LM iterations, " << st.pcg_iterations << " PCG iterations)." << std::endl;
  const auto mse = graph.chi2() / num_observations;
  std::cout << "MSE: " << mse << std::endl;
  std::cout << "Half MSE: " << mse / 2 << std::endl;
  graph.read_back(cameras, points);
  solver_ptr.reset();
}

int main(int argc, char *argv[]) {
  try {
    const Args a = parse(argc, argv);
    if (a.precision == "FP64-FP64") bundle_adjustment<double>(a);
    else if (a.precision == "FP32-FP32") bundle_adjustment<float>(a);
    else if (a.precision == "FP64-FP32") bundle_adjustment<double, float>(a);
    else throw std::runtime_error("Unsupported precision option (FP64-FP64 | FP64-FP32 | FP32-FP32; the BF16 storage modes are not provided)");
  } catch (const std::exception &e) {
    std::cerr << "Error during bundle adjustment: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
