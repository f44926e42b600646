// bal.cpp — the BAL driver of the reference (examples/bal.cu:43-360) on the MI355X library.
// Same file format, same options (--lambda --iterations --verbose --pcg_iterations --pcg_tolerance
// --rejection_ratio --precision {FP64-FP64,FP32-FP32} --solver {pcg,pcg-schur} --identity_damping),
// same printed summary (MSE / Half MSE).  Host-only C++17: g++ -Iinclude examples/bal.cpp
// -Lgraphite_amd -lgraphite_mi355x -Wl,-rpath,$PWD/graphite_amd -o bal
#include "graphite_mi355x.hpp"
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
    else if (s.rfind("--", 0) == 0) throw std::runtime_error("unknown option " + s);
    else a.file = s;
  }
  if (a.file.empty()) throw std::runtime_error("usage: bal <file> [--lambda 1e-4] [--iterations 50] [--verbose] [--pcg_iterations 10] "
                                               "[--pcg_tolerance 1.0] [--rejection_ratio 5.0] [--precision FP64-FP64|FP32-FP32] "
                                               "[--solver pcg|pcg-schur] [--identity_damping]");
  return a;
}

template <typename FP> void bundle_adjustment(const Args &a) {
  using namespace graphite;
  std::cout << "Running bundle adjustment with graph precision = " << (sizeof(FP) == 8 ? "double" : "float")
            << " and solver precision = " << (sizeof(FP) == 8 ? "double" : "float") << std::endl;
  std::ifstream file(a.file);
  if (!file.is_open()) { std::cerr << "Error: Unable to open file " << a.file << std::endl; throw std::runtime_error("File open error"); }
  size_t num_cameras = 0, num_points = 0, num_observations = 0;
  file >> num_cameras >> num_points >> num_observations;
  std::cout << "Number of cameras: " << num_cameras << std::endl;
  std::cout << "Number of points: " << num_points << std::endl;
  std::cout << "Number of observations: " << num_observations << std::endl;
  auto start = std::chrono::steady_clock::now();
  std::vector<int32_t> cam_idx(num_observations), pt_idx(num_observations);
  std::vector<FP> obs(2 * num_observations), cameras(9 * num_cameras), points(3 * num_points);
  for (size_t i = 0; i < num_observations; ++i) file >> cam_idx[i] >> pt_idx[i] >> obs[2 * i] >> obs[2 * i + 1];
  for (auto &v : cameras) file >> v;
  for (auto &v : points) file >> v;
  if (!file) throw std::runtime_error("truncated BAL file");
  std::cout << "Reading the problem took " << std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() << " seconds." << std::endl;

  start = std::chrono::steady_clock::now();
  BalGraph<FP> graph(cameras, points, obs, cam_idx, pt_idx);
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
  } else throw std::runtime_error("Unsupported solver option (pcg | pcg-schur; eigen/cudss variants are not provided)");

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
    else throw std::runtime_error("Unsupported precision option (FP64-FP64 | FP32-FP32)");
  } catch (const std::exception &e) {
    std::cerr << "Error during bundle adjustment: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
