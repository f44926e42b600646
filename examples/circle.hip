// circle.hip — the reference's examples/circle.cu (fit noisy points to a circle of known radius) against the
// MI355X generic layer.  Same includes, same traits, same driver; the only change is the 2-vector type
// (the reference uses Eigen::Matrix<T,2,1>, which this image does not have).
//   hipcc --offload-arch=gfx950 -std=c++17 -Iinclude examples/circle.hip -Lgraphite_amd -lgraphite_mi355x
#include <array>
#include <chrono>
#include <graphite/optimizer/levenberg_marquardt.hpp>
#include <string>
#include <graphite/preconditioner/identity.hpp>
#include <graphite/solver/pcg.hpp>
#include <iostream>
#include <numeric>
#include <random>
#include <vector>

namespace graphite {

// Point definition
template <typename T> struct Point {
  T v[2];
  hd_fn Point() : v{0, 0} {}
  hd_fn Point(T x, T y) : v{x, y} {}
  hd_fn T operator()(int i) const { return v[i]; }
  hd_fn T &operator()(int i) { return v[i]; }
};

// Traits for Point
template <typename T, typename S> struct PointTraits {
  static constexpr size_t dimension = 2;
  using Vertex = Point<T>;

  template <typename P> d_fn static void parameters(const Vertex &vertex, P *parameters) {
    parameters[0] = P(vertex(0));
    parameters[1] = P(vertex(1));
  }

  d_fn static void update(Vertex &vertex, const T *delta) {
    vertex(0) += delta[0];
    vertex(1) += delta[1];
  }
};

template <typename T, typename S> using PointDescriptor = VertexDescriptor<T, S, PointTraits<T, S>>;

// Factor traits for the circle constraint
template <typename T, typename S> struct CircleFactorTraits {
  static constexpr size_t dimension = 1;
  using VertexDescriptors = std::tuple<PointDescriptor<T, S>>;
  using Observation = T;
  using Data = Empty;
  using Loss = DefaultLoss<T, dimension>;
#ifdef CIRCLE_AUTODIFF
  using Differentiation = DifferentiationMode::Auto;
#else
  using Differentiation = DifferentiationMode::Manual;
#endif

  template <typename D> d_fn static void error(const D *point, const T &obs, D *error) {
    const auto x = point[0];
    const auto y = point[1];
    const auto r = obs;
    error[0] = x * x + y * y - D(r * r);
  }

  template <typename J, size_t I> d_fn static void jacobian(const Point<T> &point, const T &obs, J *jacobian) {
    if constexpr (I == 0) {
      const auto x = point(0);
      const auto y = point(1);
      jacobian[0] = 2 * x;
      jacobian[1] = 2 * y;
    }
  }
};

template <typename T, typename S> using CircleFactor = FactorDescriptor<T, S, CircleFactorTraits<T, S>>;

} // namespace graphite

int main(int argc, char **argv) {
  using namespace graphite;
  (void)hipSetDevice(0);
  using FP = double;
  using SP = double;
  Graph<FP, SP> graph;

  const size_t num_vertices = argc > 1 ? std::stoul(argv[1]) : 5;

  auto point_desc = PointDescriptor<FP, SP>();
  point_desc.reserve(num_vertices);
  graph.add_descriptor(&point_desc);

  FP center[2] = {0.0, 0.0};
  std::mt19937 gen(5); // fixed seed (the reference draws from std::random_device)
  std::uniform_real_distribution<FP> dist(0.0, 2 * M_PI);
  const FP radius = 4.0;
  const FP sigma = 0.3;
  std::normal_distribution<FP> n1(0.0, sigma);
  std::normal_distribution<FP> n2(0.0, sigma);

  managed_vector<Point<FP>> pts(num_vertices); // addresses must not change
  constexpr auto id_offset = 10;               // user provides arbitrary ids
  std::vector<Point<FP>> initial(num_vertices);

  for (size_t vertex_id = 0; vertex_id < num_vertices; ++vertex_id) {
    FP angle = dist(gen);
    FP point[2] = {center[0] + radius * cos(angle), center[1] + radius * sin(angle)};
    point[0] += n1(gen);
    point[1] += n2(gen);
    pts[vertex_id] = Point<FP>(point[0], point[1]);
    initial[vertex_id] = pts[vertex_id];
    point_desc.add_vertex(vertex_id + id_offset, &pts[vertex_id]);
  }
  auto factor_desc = CircleFactor<FP, SP>(&point_desc);
  factor_desc.reserve(num_vertices);
  graph.add_descriptor(&factor_desc);

  const auto loss = DefaultLoss<FP, 1>();
  for (size_t vertex_id = 0; vertex_id < num_vertices; ++vertex_id)
    factor_desc.add_factor({vertex_id + id_offset}, radius, nullptr, Empty(), loss);

  // Set the last vertex as fixed
  point_desc.set_fixed(num_vertices - 1 + id_offset, true);
  // Disable third constraint for point 2
  factor_desc.set_active(2, 0x1);

  graphite::IdentityPreconditioner<FP, SP> preconditioner;
  graphite::PCGSolver<FP, SP> solver(50, 1e-20, 10.0, &preconditioner);

  constexpr size_t iterations = 100;
  std::cout << "Graph built with " << num_vertices << " vertices and " << factor_desc.internal_count() << " factors." << std::endl;
  std::cout << "Optimizing!" << std::endl;

  StreamPool streams(1);
  optimizer::LevenbergMarquardtOptions<FP, SP> options;
  options.solver = &solver;
  options.initial_damping = 1e-6;
  options.iterations = iterations;
  options.optimization_level = 0;
  options.verbose = argc > 2;
  options.streams = &streams;

  auto start = std::chrono::steady_clock::now();
  if (argc > 3 && std::string(argv[3]) == "lm2") optimizer::levenberg_marquardt2<FP, SP>(&graph, &options); // early termination variant
  else optimizer::levenberg_marquardt<FP, SP>(&graph, &options);
  std::chrono::duration<double> elapsed = std::chrono::steady_clock::now() - start;
  std::cout << "Optimization took " << elapsed.count() << " seconds." << std::endl;

  int failures = 0;
  for (size_t vertex_id = 0; vertex_id < num_vertices; ++vertex_id) {
    const auto &p = *point_desc.get_vertex(vertex_id + id_offset);
    const FP r = sqrt(p(0) * p(0) + p(1) * p(1));
    std::cout << "Optimized point " << vertex_id << "=(" << p(0) << ", " << p(1) << ") with radius=" << r << std::endl;
    const bool frozen = vertex_id == 2 || vertex_id == num_vertices - 1;
    if (frozen) failures += !(p(0) == initial[vertex_id](0) && p(1) == initial[vertex_id](1));
    else failures += !(std::abs(r - radius) < 1e-6);
  }
  std::cout << "points 2 and " << num_vertices - 1 << " should remain unchanged." << std::endl;
  std::cout << (failures ? "FAILED" : "OK") << " (" << failures << " failures)" << std::endl;
  return failures != 0;
}
