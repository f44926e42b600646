// The literal known answers of the reference's own unit tests (every bold row of SURVEY section 4), replayed on the HIP generic
// layer — same fixtures, same expected numbers, float with EXPECT_FLOAT_EQ's 4-ULP bar:
//   tests/factor.cu:126-137 use_autodiff | :139-157 ComputeError | :159-294 AddFactor / RemoveFactorFrom{Beginning,Middle,End} /
//     RemoveAllFactors (swap-remove bookkeeping, each as its own fixture) | :296-322 ComputeErrorAutodiff | :324-358
//     FlagActiveVerticesAsync (activity by optimisation level, the member the reference calls) | :360-423 Jacobians + scaling |
//     :425-509 b, Huber b | :511-595 block / scalar diagonal | :597-756 Jv / J^T v | :758-784 chi2 | :786-801 default precision
//     matrix | :803-852 ClearResetsStorage | :854-967 ComputeHessian (5 block coordinates, offsets [0 8 0 4 8], the 12 values —
//     through get_hessian_block_coordinates / setup_ / execute_hessian_computation AND through Hessian::build_structure /
//     update_values)
//   tests/vertex.cu:23-74 add / replace | :76-119 update | :121-166 AugmentBlockDiagonal | :168-226 ApplyBlockJacobi |
//     :228-297 RemoveVertexFrom{Start,Middle,End} | :299-341 backup / restore
//   tests/vector.cu:6-79 managed_vector (default construction, push / pop, reserve, resize / clear, data / begin / end)
// One Vec2 vertex at (7, 0), observation 2.5, J = [1 0] or [2 3], Huber delta 1 unless the row says otherwise.
// The CPU oracle replays the numeric vectors too (tests/test_oracle_known_answers.py).
#include <cmath>
#include <cstring>
#include <graphite/factor.hpp>
#include <graphite/graph.hpp>
#include <graphite/hessian.hpp>
#include <iostream>
#include <unordered_map>

static int failures = 0, checks = 0;
static bool float_eq(float a, float b) { // gtest AlmostEquals: 4 ULPs
  if (std::isnan(a) || std::isnan(b)) return false;
  int32_t ia, ib;
  std::memcpy(&ia, &a, 4); std::memcpy(&ib, &b, 4);
  if (ia < 0) ia = (int32_t)0x80000000 - ia;
  if (ib < 0) ib = (int32_t)0x80000000 - ib;
  return std::abs((int64_t)ia - (int64_t)ib) <= 4;
}
#define EXPECT_FLOAT_EQ(a, b) do { ++checks; if (!float_eq((a), (b))) { ++failures; std::cout << "FAIL " << __LINE__ << ": " #a " = " << (a) << " expected " << (b) << std::endl; } } while (0)
#define EXPECT_EQ(a, b) do { ++checks; if (!((a) == (b))) { ++failures; std::cout << "FAIL " << __LINE__ << ": " #a << std::endl; } } while (0)

struct Vec2 { float x, y; };
struct Vec2Traits {
  static constexpr size_t dimension = 2;
  using Vertex = Vec2;
  template <typename P> d_fn static void parameters(const Vertex &v, P *p) { p[0] = P(v.x); p[1] = P(v.y); }
  d_fn static void update(Vertex &v, const float *d) { v.x += d[0]; v.y += d[1]; }
};
using Vec2Descriptor = graphite::VertexDescriptor<float, float, Vec2Traits>;

template <typename DiffMode, template <typename, int> class LossT = graphite::DefaultLoss> struct UnaryFactorTraits {
  static constexpr size_t dimension = 1;
  using VertexDescriptors = std::tuple<Vec2Descriptor>;
  using Observation = float;
  using Data = graphite::Empty;
  using Loss = LossT<float, dimension>;
  using Differentiation = DiffMode;
  template <typename D> d_fn static void error(const D *vertex, const Observation &obs, D *residual) { residual[0] = vertex[0] - D(obs); }
  template <typename D, size_t I> d_fn static void jacobian(const Vec2 &, const Observation &, D *jacobian) { jacobian[0] = D(1); jacobian[1] = D(0); }
};
template <typename DiffMode, template <typename, int> class LossT = graphite::DefaultLoss> struct CoupledUnaryFactorTraits {
  static constexpr size_t dimension = 1;
  using VertexDescriptors = std::tuple<Vec2Descriptor>;
  using Observation = float;
  using Data = graphite::Empty;
  using Loss = LossT<float, dimension>;
  using Differentiation = DiffMode;
  template <typename D> d_fn static void error(const D *vertex, const Observation &obs, D *residual) { residual[0] = D(2.0f) * vertex[0] + D(3.0f) * vertex[1] - D(obs); }
  template <typename D, size_t I> d_fn static void jacobian(const Vec2 &, const Observation &, D *jacobian) { jacobian[0] = D(2); jacobian[1] = D(3); }
};
template <typename DiffMode> struct BinaryFactorTraits {
  static constexpr size_t dimension = 1;
  using VertexDescriptors = std::tuple<Vec2Descriptor, Vec2Descriptor>;
  using Observation = float;
  using Data = graphite::Empty;
  using Loss = graphite::DefaultLoss<float, dimension>;
  using Differentiation = DiffMode;
  template <typename D> d_fn static void error(const D *v0, const D *v1, const Observation &obs, D *residual) {
    residual[0] = v0[0] + D(2.0f) * v0[1] + D(3.0f) * v1[0] + D(4.0f) * v1[1] - D(obs);
  }
  template <typename D, size_t I> d_fn static void jacobian(const Vec2 &, const Vec2 &, const Observation &, D *jacobian) {
    if constexpr (I == 0) { jacobian[0] = D(1); jacobian[1] = D(2); } else { jacobian[0] = D(3); jacobian[1] = D(4); }
  }
};
struct Residual2FactorTraits { // tests/factor.cu:96-115: a two-component residual (its precision matrix is 2 x 2)
  static constexpr size_t dimension = 2;
  using VertexDescriptors = std::tuple<Vec2Descriptor>;
  using Observation = Vec2;
  using Data = graphite::Empty;
  using Loss = graphite::DefaultLoss<float, dimension>;
  using Differentiation = graphite::DifferentiationMode::Manual;
  template <typename D> d_fn static void error(const D *vertex, const Observation &obs, D *residual) { residual[0] = vertex[0] - D(obs.x); residual[1] = vertex[1] - D(obs.y); }
  template <typename D, size_t I> d_fn static void jacobian(const Vec2 &, const Observation &, D *jacobian) { jacobian[0] = D(1); jacobian[1] = D(0); jacobian[2] = D(0); jacobian[3] = D(1); }
};
using Residual2ManualFactor = graphite::FactorDescriptor<float, float, Residual2FactorTraits>;
using Auto = graphite::DifferentiationMode::Auto;
using Manual = graphite::DifferentiationMode::Manual;
using AutoFactor = graphite::FactorDescriptor<float, float, UnaryFactorTraits<Auto>>;
using ManualFactor = graphite::FactorDescriptor<float, float, UnaryFactorTraits<Manual>>;
using CoupledManualFactor = graphite::FactorDescriptor<float, float, CoupledUnaryFactorTraits<Manual>>;
using CoupledAutoFactor = graphite::FactorDescriptor<float, float, CoupledUnaryFactorTraits<Auto>>;
using BinaryManualFactor = graphite::FactorDescriptor<float, float, BinaryFactorTraits<Manual>>;
using BinaryAutoFactor = graphite::FactorDescriptor<float, float, BinaryFactorTraits<Auto>>;
using ManualHuberFactor = graphite::FactorDescriptor<float, float, UnaryFactorTraits<Manual, graphite::HuberLoss>>;

struct Fixture {
  Vec2Descriptor vertex_desc;
  graphite::managed_vector<Vec2> vertices;
  Fixture() {
    vertices.push_back(Vec2{7.0f, 0.0f});
    vertex_desc.add_vertex(10, vertices.data().get(), false);
    vertex_desc.set_hessian_column(10, 0, 0);
  }
};
static void dsync() { (void)hipDeviceSynchronize(); }

int main() {
  using namespace graphite;
  (void)hipSetDevice(0);
  { // UseAutodiffReflectsDifferentiationMode (factor.cu:126-137)
    EXPECT_EQ(AutoFactor::use_autodiff(), true); EXPECT_EQ(ManualFactor::use_autodiff(), false);
    EXPECT_EQ(AutoFactor::supports_dynamic_jacobians(), false); EXPECT_EQ(ManualFactor::supports_dynamic_jacobians(), true);
  }
  { // ComputeError (factor.cu:139-157)
    Fixture fx; ManualFactor factor(&fx.vertex_desc);
    factor.add_factor({10}, 2.5f);
    factor.initialize_device_ids(0);
    factor.compute_error(); dsync();
    EXPECT_EQ(factor.residuals.size(), 1u);
    EXPECT_FLOAT_EQ(factor.residuals[0], 7.0f - 2.5f);
  }
  { // AddFactor / RemoveFactor* (factor.cu:159-294)
    Fixture fx; ManualFactor factor(&fx.vertex_desc);
    graphite::managed_vector<Vec2> more; more.reserve(2); more.push_back(Vec2{1, 2}); more.push_back(Vec2{3, 4});
    fx.vertex_desc.add_vertex(20, &more[0]); fx.vertex_desc.add_vertex(30, &more[1]);
    const auto f0 = factor.add_factor({10}, 1.5f), f1 = factor.add_factor({20}, 2.5f), f2 = factor.add_factor({30}, 3.5f);
    EXPECT_EQ(factor.internal_count(), 3u);
    EXPECT_FLOAT_EQ(factor.device_obs[0], 1.5f); EXPECT_FLOAT_EQ(factor.device_obs[1], 2.5f); EXPECT_FLOAT_EQ(factor.device_obs[2], 3.5f);
    EXPECT_EQ(factor.get_vertex_ids(f0)[0], 10u); EXPECT_EQ(factor.get_vertex_ids(f1)[0], 20u); EXPECT_EQ(factor.get_vertex_ids(f2)[0], 30u);
    factor.remove_factor(f1); // RemoveFactorFromMiddle (factor.cu:212-236): the other handles keep naming their factors
    EXPECT_EQ(factor.internal_count(), 2u);
    EXPECT_EQ(factor.get_vertex_ids(f0)[0], 10u); EXPECT_EQ(factor.get_vertex_ids(f2)[0], 30u);
    EXPECT_FLOAT_EQ(factor.device_obs[1], 3.5f); // the last factor took the freed slot (factor.hpp:318-355)
    const auto f3 = factor.add_factor({20}, 4.5f); // the released handle is re-used (utils.hpp:88-96)
    EXPECT_EQ(f3, f1); EXPECT_EQ(factor.get_vertex_ids(f3)[0], 20u);
    factor.remove_factor(f0); factor.remove_factor(f3); factor.remove_factor(f2); // RemoveAllFactors (factor.cu:264-294)
    EXPECT_EQ(factor.internal_count(), 0u);
    factor.initialize_device_ids(0);
    EXPECT_EQ(factor.active_count(), 0u);
  }
  { // ComputeErrorAutodiff (factor.cu:296-322)
    Fixture fx; AutoFactor factor(&fx.vertex_desc);
    factor.add_factor({10}, 2.5f);
    factor.initialize_device_ids(0);
    EXPECT_EQ(factor.active_count(), 1u);
    factor.compute_error(); factor.compute_jacobians(); dsync();
    EXPECT_FLOAT_EQ(factor.residuals[0], 7.0f - 2.5f);
    EXPECT_FLOAT_EQ(factor.jacobians[0].data[0], 1.0f); EXPECT_FLOAT_EQ(factor.jacobians[0].data[1], 0.0f);
  }
  { // FlagActiveVerticesAsync (factor.cu:324-358)
    Fixture fx; graphite::managed_vector<Vec2> more; more.push_back(Vec2{1, 2});
    fx.vertex_desc.add_vertex(20, &more[0]);
    ManualFactor factor(&fx.vertex_desc);
    const auto f0 = factor.add_factor({10}, 2.5f); const auto f1 = factor.add_factor({20}, 2.5f);
    factor.set_active(f1, 1);
    factor.initialize_device_ids(0);
    factor.flag_active_vertices(); dsync();
    const auto *st = fx.vertex_desc.get_active_state();
    EXPECT_EQ(st[fx.vertex_desc.get_local_id(10)] & 0x80, 0x80); EXPECT_EQ(st[fx.vertex_desc.get_local_id(20)] & 0x80, 0x00);
    factor.set_active(f1, 0); factor.initialize_device_ids(0); factor.flag_active_vertices(); dsync();
    EXPECT_EQ(st[fx.vertex_desc.get_local_id(20)] & 0x80, 0x80);
    (void)f0;
  }
  { // ComputeJacobians + ScaleJacobiansAsync (factor.cu:360-423)
    Fixture fx; CoupledManualFactor factor(&fx.vertex_desc);
    factor.add_factor({10}, 2.5f); factor.add_factor({10}, 2.5f);
    factor.initialize_device_ids(0);
    StreamPool streams(1);
    factor.compute_jacobians(streams); dsync();
    EXPECT_EQ(factor.jacobians[0].data.size(), 4u);
    EXPECT_FLOAT_EQ(factor.jacobians[0].data[0], 2.0f); EXPECT_FLOAT_EQ(factor.jacobians[0].data[1], 3.0f);
    EXPECT_FLOAT_EQ(factor.jacobians[0].data[2], 2.0f); EXPECT_FLOAT_EQ(factor.jacobians[0].data[3], 3.0f);
    graphite::managed_vector<float> scales(2); scales[0] = 2.0f; scales[1] = 3.0f;
    factor.scale_jacobians(scales.data().get()); dsync();
    EXPECT_FLOAT_EQ(factor.jacobians[0].data[0], 4.0f); EXPECT_FLOAT_EQ(factor.jacobians[0].data[1], 9.0f);
    EXPECT_FLOAT_EQ(factor.jacobians[0].data[2], 4.0f); EXPECT_FLOAT_EQ(factor.jacobians[0].data[3], 9.0f);
  }
  { // the same Jacobian by dual numbers
    Fixture fx; CoupledAutoFactor factor(&fx.vertex_desc);
    factor.add_factor({10}, 2.5f);
    factor.initialize_device_ids(0);
    factor.compute_jacobians(); dsync();
    EXPECT_FLOAT_EQ(factor.jacobians[0].data[0], 2.0f); EXPECT_FLOAT_EQ(factor.jacobians[0].data[1], 3.0f);
  }
  { // ComputeB (factor.cu:425-466)
    Fixture fx; ManualFactor factor(&fx.vertex_desc);
    factor.add_factor({10}, 2.5f); factor.add_factor({10}, 2.5f);
    factor.initialize_device_ids(0);
    factor.compute_error(); factor.compute_jacobians(); factor.chi2();
    graphite::managed_vector<float> b(2); b[0] = 3.0f; b[1] = -7.0f;
    factor.compute_b(b.data().get()); factor.compute_b(b.data().get()); dsync();
    EXPECT_FLOAT_EQ(b[0], 3.0f - 4.0f * (7.0f - 2.5f)); EXPECT_FLOAT_EQ(b[1], -7.0f);
  }
  { // ComputeBHuberLoss (factor.cu:468-509)
    Fixture fx; ManualHuberFactor factor(&fx.vertex_desc);
    const HuberLoss<float, 1> huber(1.0f);
    factor.add_factor({10}, 2.5f, nullptr, Empty{}, huber); factor.add_factor({10}, 2.5f, nullptr, Empty{}, huber);
    factor.initialize_device_ids(0);
    factor.compute_error(); factor.compute_jacobians(); factor.chi2();
    graphite::managed_vector<float> b(2); b[0] = 5.0f; b[1] = -11.0f;
    factor.compute_b(b.data().get()); factor.compute_b(b.data().get()); dsync();
    EXPECT_FLOAT_EQ(b[0], 5.0f - 4.0f); EXPECT_FLOAT_EQ(b[1], -11.0f);
  }
  { // ComputeHessianBlockDiagonal + ScalarDiagonal (factor.cu:511-595)
    Fixture fx; CoupledManualFactor factor(&fx.vertex_desc);
    factor.add_factor({10}, 2.5f); factor.add_factor({10}, 2.5f);
    factor.initialize_device_ids(0);
    factor.compute_jacobians(); factor.compute_error(); factor.chi2();
    graphite::managed_vector<float> blk(4, 0.0f), diag(2, 0.0f);
    factor.block_diagonal(0, blk.data().get()); factor.scalar_diagonal(diag.data().get()); dsync();
    EXPECT_FLOAT_EQ(blk[0], 8.0f); EXPECT_FLOAT_EQ(blk[1], 12.0f); EXPECT_FLOAT_EQ(blk[2], 12.0f); EXPECT_FLOAT_EQ(blk[3], 18.0f);
    EXPECT_FLOAT_EQ(diag[0], 8.0f); EXPECT_FLOAT_EQ(diag[1], 18.0f);
  }
  { // the same expectations with set_jacobian_storage(false) (factor.hpp:626-640): nothing stored, blocks recomputed
    Fixture fx; CoupledManualFactor factor(&fx.vertex_desc);
    factor.set_jacobian_storage(false);
    EXPECT_EQ(factor.dynamic_jacobians(), true);
    factor.add_factor({10}, 2.5f); factor.add_factor({10}, 2.5f);
    factor.initialize_device_ids(0);
    factor.compute_jacobians(); factor.compute_error(); factor.chi2();
    EXPECT_EQ(factor.jacobians[0].data.size(), 0u);
    graphite::managed_vector<float> blk(4, 0.0f), diag(2, 0.0f), b(2, 0.0f), sc(2);
    factor.block_diagonal(0, blk.data().get()); factor.scalar_diagonal(diag.data().get()); factor.compute_b(b.data().get()); dsync();
    EXPECT_FLOAT_EQ(blk[0], 8.0f); EXPECT_FLOAT_EQ(blk[1], 12.0f); EXPECT_FLOAT_EQ(blk[2], 12.0f); EXPECT_FLOAT_EQ(blk[3], 18.0f);
    EXPECT_FLOAT_EQ(diag[0], 8.0f); EXPECT_FLOAT_EQ(diag[1], 18.0f);
    const float r = 2.0f * 7.0f + 3.0f * 0.0f - 2.5f; // CoupledUnaryFactorTraits: J = [2 3]
    EXPECT_FLOAT_EQ(b[0], -2.0f * 2.0f * r); EXPECT_FLOAT_EQ(b[1], -2.0f * 3.0f * r);
    // column scales act on the recomputed blocks exactly as they would on stored ones
    sc[0] = 0.5f; sc[1] = 2.0f;
    factor.scale_jacobians(sc.data().get());
    blk[0] = blk[1] = blk[2] = blk[3] = 0.0f;
    factor.block_diagonal(0, blk.data().get()); dsync();
    EXPECT_FLOAT_EQ(blk[0], 8.0f * 0.25f); EXPECT_FLOAT_EQ(blk[1], 12.0f); EXPECT_FLOAT_EQ(blk[3], 18.0f * 4.0f);
    // an Auto factor keeps storing (ops/linearize.hpp:109)
    CoupledAutoFactor af(&fx.vertex_desc); af.set_jacobian_storage(false);
    EXPECT_EQ(af.dynamic_jacobians(), false);
  }
  { // ComputeJvHuberLoss / ComputeJtvHuberLoss (factor.cu:597-756)
    Fixture fx; ManualHuberFactor factor(&fx.vertex_desc);
    const HuberLoss<float, 1> huber(1.0f);
    const auto f0 = factor.add_factor({10}, 2.5f, nullptr, Empty{}, huber), f1 = factor.add_factor({10}, 2.5f, nullptr, Empty{}, huber);
    factor.initialize_device_ids(0);
    EXPECT_EQ(factor.active_count(), 2u);
    factor.compute_error(); factor.compute_jacobians(); factor.chi2();
    graphite::managed_vector<float> out(2, 0.0f), in(2);
    in[0] = 3.0f; in[1] = 5.0f;
    factor.compute_Jv(out.data().get(), in.data().get()); dsync();
    EXPECT_FLOAT_EQ(out[0], 3.0f); EXPECT_FLOAT_EQ(out[1], 3.0f); // J = [1 0]: each factor's row is in[0]
    fx.vertex_desc.set_fixed(10, true); out[0] = 17.0f; out[1] = 23.0f;
    factor.compute_Jv(out.data().get(), in.data().get()); dsync();
    EXPECT_FLOAT_EQ(out[0], 17.0f); EXPECT_FLOAT_EQ(out[1], 23.0f);
    fx.vertex_desc.set_fixed(10, false); fx.vertex_desc.get_active_state()[0] = 0x80; out[0] = 29.0f; out[1] = 31.0f;
    factor.compute_Jv(out.data().get(), in.data().get()); dsync();
    EXPECT_FLOAT_EQ(out[0], 29.0f); EXPECT_FLOAT_EQ(out[1], 31.0f);
    fx.vertex_desc.get_active_state()[0] = 0;
    // J^T: rho' = 1 / 4.5 per factor, in = [9 9] -> [4 0]
    out[0] = 0.0f; out[1] = 0.0f; in[0] = 9.0f; in[1] = 9.0f;
    factor.compute_Jtv(out.data().get(), in.data().get()); dsync();
    EXPECT_FLOAT_EQ(out[0], 4.0f); EXPECT_FLOAT_EQ(out[1], 0.0f);
    fx.vertex_desc.set_fixed(10, true); out[0] = 43.0f; out[1] = 47.0f;
    factor.compute_Jtv(out.data().get(), in.data().get()); dsync();
    EXPECT_FLOAT_EQ(out[0], 43.0f); EXPECT_FLOAT_EQ(out[1], 47.0f);
    fx.vertex_desc.set_fixed(10, false);
    factor.set_active(f0, 1); factor.set_active(f1, 1);
    factor.initialize_device_ids(0);
    EXPECT_EQ(factor.active_count(), 0u);
    out[0] = 61.0f; out[1] = 67.0f;
    factor.compute_Jtv(out.data().get(), in.data().get()); factor.compute_Jv(out.data().get(), in.data().get()); dsync();
    EXPECT_FLOAT_EQ(out[0], 61.0f); EXPECT_FLOAT_EQ(out[1], 67.0f);
  }
  { // Chi2HuberLoss (factor.cu:758-784): r = 4.5, delta = 1 -> 2 sqrt(20.25) - 1 = 8 per factor
    Fixture fx; ManualHuberFactor factor(&fx.vertex_desc);
    const HuberLoss<float, 1> huber(1.0f);
    factor.add_factor({10}, 2.5f, nullptr, Empty{}, huber); factor.add_factor({10}, 2.5f, nullptr, Empty{}, huber);
    factor.initialize_device_ids(0);
    factor.compute_error();
    EXPECT_FLOAT_EQ(factor.chi2(), 16.0f);
    EXPECT_FLOAT_EQ(factor.chi2(0), 8.0f);
    EXPECT_FLOAT_EQ(factor.chi2_derivative[0], 1.0f / 4.5f);
  }
  { // binary factor, both differentiation modes: J0 = [1 2], J1 = [3 4]
    Fixture fx; graphite::managed_vector<Vec2> more; more.push_back(Vec2{1.0f, -2.0f});
    fx.vertex_desc.add_vertex(20, &more[0]);
    BinaryManualFactor fm(&fx.vertex_desc, &fx.vertex_desc); BinaryAutoFactor fa(&fx.vertex_desc, &fx.vertex_desc);
    fm.add_factor({10, 20}, 0.5f); fa.add_factor({10, 20}, 0.5f);
    fm.initialize_device_ids(0); fa.initialize_device_ids(0);
    fm.compute_error(); fm.compute_jacobians(); fa.compute_error(); fa.compute_jacobians(); dsync();
    EXPECT_FLOAT_EQ(fm.residuals[0], 7.0f + 3.0f * 1.0f + 4.0f * -2.0f - 0.5f); EXPECT_FLOAT_EQ(fa.residuals[0], fm.residuals[0]);
    for (int k = 0; k < 2; ++k) { EXPECT_FLOAT_EQ(fa.jacobians[0].data[k], fm.jacobians[0].data[k]); EXPECT_FLOAT_EQ(fa.jacobians[1].data[k], fm.jacobians[1].data[k]); }
    EXPECT_FLOAT_EQ(fm.jacobians[0].data[1], 2.0f); EXPECT_FLOAT_EQ(fm.jacobians[1].data[0], 3.0f);
  }
  { // vertex update / backup / restore (vertex.cu:76-119, 299-341): v + delta * scale, fixed vertices untouched
    graphite::managed_vector<Vec2> vs; vs.reserve(2); vs.push_back(Vec2{1.0f, 2.0f}); vs.push_back(Vec2{3.0f, 4.0f});
    Vec2Descriptor vd; vd.add_vertex(5, &vs[0], false); vd.add_vertex(6, &vs[1], true);
    vd.set_hessian_column(5, 0, 0);
    graphite::managed_vector<float> dx(2), sc(2); dx[0] = 0.5f; dx[1] = -1.0f; sc[0] = 2.0f; sc[1] = 3.0f;
    vd.backup_parameters(); vd.apply_update(dx.data().get(), sc.data().get()); dsync();
    EXPECT_FLOAT_EQ(vs[0].x, 2.0f); EXPECT_FLOAT_EQ(vs[0].y, -1.0f); EXPECT_FLOAT_EQ(vs[1].x, 3.0f); EXPECT_FLOAT_EQ(vs[1].y, 4.0f);
    vd.restore_parameters(); dsync();
    EXPECT_FLOAT_EQ(vs[0].x, 1.0f); EXPECT_FLOAT_EQ(vs[0].y, 2.0f);
    EXPECT_EQ(vd.is_fixed(6), true); EXPECT_EQ(vd.is_active(5), true); EXPECT_EQ(vd.exists(7), false);
    vd.remove_vertex(5);
    EXPECT_EQ(vd.count(), 1u); EXPECT_EQ(vd.get_vertex(6), &vs[1]);
  }
  // ---- round 6: the reference vectors that were not yet replayed here ----------------------------------------------------------
  for (int which = 0; which < 3; ++which) { // RemoveFactorFromBeginning / Middle / End (factor.cu:187-262), each from a fresh descriptor
    graphite::managed_vector<Vec2> vs; vs.reserve(3); vs.push_back(Vec2{1, 0}); vs.push_back(Vec2{2, 0}); vs.push_back(Vec2{3, 0});
    Vec2Descriptor vd; vd.add_vertex(10, &vs[0], false); vd.add_vertex(20, &vs[1], false); vd.add_vertex(30, &vs[2], false);
    ManualFactor factor(&vd);
    const size_t h[3] = {factor.add_factor({10}, 1.5f), factor.add_factor({20}, 2.5f), factor.add_factor({30}, 3.5f)};
    factor.to_device();
    factor.remove_factor(h[which]);
    EXPECT_EQ(factor.internal_count(), 2u);
    for (int k = 0; k < 3; ++k) if (k != which) EXPECT_EQ(factor.get_vertex_ids(h[k])[0], (size_t)(10 * (k + 1)));
  }
  { // RemoveAllFactors (factor.cu:264-294): the id tables are empty afterwards
    graphite::managed_vector<Vec2> vs; vs.reserve(3); vs.push_back(Vec2{1, 0}); vs.push_back(Vec2{2, 0}); vs.push_back(Vec2{3, 0});
    Vec2Descriptor vd; vd.add_vertex(10, &vs[0], false); vd.add_vertex(20, &vs[1], false); vd.add_vertex(30, &vs[2], false);
    ManualFactor factor(&vd);
    const auto f0 = factor.add_factor({10}, 1.5f), f1 = factor.add_factor({20}, 2.5f), f2 = factor.add_factor({30}, 3.5f);
    factor.to_device();
    factor.remove_factor(f0); factor.remove_factor(f1); factor.remove_factor(f2);
    EXPECT_EQ(factor.internal_count(), 0u);
    factor.initialize_device_ids(0);
    EXPECT_EQ(factor.active_count(), 0u);
    EXPECT_EQ(factor.host_ids.empty(), true); EXPECT_EQ(factor.device_ids.empty(), true);
  }
  { // FlagActiveVerticesAsync (factor.cu:324-358) through the member the reference calls, level 0 then level 1
    graphite::managed_vector<Vec2> vs; vs.reserve(2); vs.push_back(Vec2{7, 0}); vs.push_back(Vec2{9, 1});
    Vec2Descriptor vd; vd.add_vertex(10, &vs[0], false); vd.add_vertex(20, &vs[1], false); vd.to_device();
    ManualFactor factor(&vd);
    const auto f0 = factor.add_factor({10}, 2.5f); const auto f1 = factor.add_factor({20}, 3.5f);
    factor.set_active(f1, 1); // inactive at optimisation level 0
    factor.initialize_device_ids(0);
    EXPECT_EQ(factor.active_count(), 1u);
    factor.flag_active_vertices_async(0); dsync();
    const auto local0 = vd.get_global_map().at(10), local1 = vd.get_global_map().at(20);
    const uint8_t *st = vd.get_active_state();
    EXPECT_EQ(st[local0] & 0x80, 0x80); EXPECT_EQ(st[local1] & 0x80, 0x00);
    factor.flag_active_vertices_async(1); dsync(); // level-gated activity when a higher level is selected
    EXPECT_EQ(st[local1] & 0x80, 0x80);
    (void)f0;
  }
  { // GetDefaultPrecisionMatrix (factor.cu:786-801): identity, E x E
    graphite::managed_vector<Vec2> vs; vs.push_back(Vec2{7, 5});
    Vec2Descriptor vd; vd.add_vertex(10, &vs[0], false); vd.to_device();
    Residual2ManualFactor factor(&vd);
    factor.add_factor({10}, Vec2{1.0f, 2.0f});
    EXPECT_EQ(factor.precision_matrices.size(), 4u);
    EXPECT_FLOAT_EQ(factor.precision_matrices[0], 1.0f); EXPECT_FLOAT_EQ(factor.precision_matrices[1], 0.0f);
    EXPECT_FLOAT_EQ(factor.precision_matrices[2], 0.0f); EXPECT_FLOAT_EQ(factor.precision_matrices[3], 1.0f);
  }
  { // ClearResetsStorage (factor.cu:803-852)
    graphite::managed_vector<Vec2> vs; vs.reserve(2); vs.push_back(Vec2{7, 5}); vs.push_back(Vec2{11, 13});
    Vec2Descriptor vd; vd.add_vertex(10, &vs[0], false); vd.add_vertex(20, &vs[1], false); vd.to_device();
    BinaryManualFactor factor(&vd, &vd);
    factor.add_factor({10, 20}, 2.5f); factor.add_factor({20, 10}, 3.5f);
    factor.initialize_device_ids(0); factor.to_device(); factor.initialize_jacobian_storage();
    StreamPool streams(2);
    factor.compute_error(); factor.compute_jacobians(streams); factor.chi2();
    EXPECT_EQ(factor.internal_count() > 0u, true); EXPECT_EQ(factor.active_count() > 0u, true);
    EXPECT_EQ(factor.host_ids.empty(), false); EXPECT_EQ(factor.device_ids.empty(), false);
    EXPECT_EQ(factor.device_obs.size() > 0u, true); EXPECT_EQ(factor.residuals.empty(), false);
    factor.clear();
    EXPECT_EQ(factor.internal_count(), 0u); EXPECT_EQ(factor.active_count(), 0u);
    EXPECT_EQ(factor.host_ids.empty(), true); EXPECT_EQ(factor.device_ids.empty(), true);
    EXPECT_EQ(factor.device_obs.size(), 0u); EXPECT_EQ(factor.residuals.empty(), true);
    EXPECT_EQ(factor.precision_matrices.size(), 0u); EXPECT_EQ(factor.data.size(), 0u);
    EXPECT_EQ(factor.chi2_vec.size(), 0u); EXPECT_EQ(factor.chi2_derivative.empty(), true);
    EXPECT_EQ(factor.loss.size(), 0u); EXPECT_EQ(factor.active.empty(), true); EXPECT_EQ(factor.active_indices.empty(), true);
  }
  { // ComputeHessian (factor.cu:854-967): two vertices (block columns 0 and 1, scalar columns 0 and 2), two unary factors with
    // J = [2 3] and one binary factor with J0 = [1 2], J1 = [3 4]; the reference's own sequence of calls
    graphite::managed_vector<Vec2> vs; vs.reserve(2); vs.push_back(Vec2{7, 5}); vs.push_back(Vec2{11, 13});
    Vec2Descriptor vd; vd.add_vertex(10, &vs[0], false); vd.add_vertex(20, &vs[1], false);
    vd.set_hessian_column(10, 0, 0); vd.set_hessian_column(20, 2, 1); vd.to_device();
    CoupledManualFactor unary(&vd); BinaryManualFactor binary(&vd, &vd);
    unary.add_factor({10}, 2.5f); unary.add_factor({20}, 3.5f); binary.add_factor({10, 20}, 4.5f);
    unary.initialize_device_ids(0); binary.initialize_device_ids(0);
    EXPECT_EQ(unary.active_count(), 2u); EXPECT_EQ(binary.active_count(), 1u);
    unary.to_device(); binary.to_device(); unary.initialize_jacobian_storage(); binary.initialize_jacobian_storage();
    StreamPool streams(2);
    unary.compute_jacobians(streams); binary.compute_jacobians(streams);
    unary.compute_error(); binary.compute_error(); unary.chi2(); binary.chi2();
    device_vector<BlockCoordinates> block_coords;
    unary.get_hessian_block_coordinates(block_coords); binary.get_hessian_block_coordinates(block_coords);
    const std::vector<BlockCoordinates> hc = block_coords.to_host();
    EXPECT_EQ(hc.size(), 5u);
    size_t n00 = 0, n01 = 0, n11 = 0;
    for (const auto &c : hc) { n00 += c.row == 0u && c.col == 0u; n01 += c.row == 0u && c.col == 1u; n11 += c.row == 1u && c.col == 1u; }
    EXPECT_EQ(n00, 2u); EXPECT_EQ(n01, 1u); EXPECT_EQ(n11, 2u);
    std::unordered_map<BlockCoordinates, size_t> block_indices;
    block_indices[BlockCoordinates{0, 0}] = 0; block_indices[BlockCoordinates{0, 1}] = 4; block_indices[BlockCoordinates{1, 1}] = 8;
    device_vector<float> d_hessian(12);
    d_hessian.zero();
    std::vector<size_t> h_block_offsets(5, static_cast<size_t>(-1));
    size_t mul_count = 0;
    mul_count += unary.setup_hessian_computation(block_indices, d_hessian, h_block_offsets.data() + mul_count, streams);
    mul_count += binary.setup_hessian_computation(block_indices, d_hessian, h_block_offsets.data() + mul_count, streams);
    EXPECT_EQ(mul_count, 5u);
    EXPECT_EQ(h_block_offsets[0], 0u); EXPECT_EQ(h_block_offsets[1], 8u); EXPECT_EQ(h_block_offsets[2], 0u);
    EXPECT_EQ(h_block_offsets[3], 4u); EXPECT_EQ(h_block_offsets[4], 8u);
    device_vector<size_t> d_block_offsets;
    d_block_offsets = h_block_offsets;
    d_hessian.zero();
    size_t exec = 0;
    exec += unary.execute_hessian_computation(block_indices, d_hessian, d_block_offsets.data().get() + exec, streams);
    exec += binary.execute_hessian_computation(block_indices, d_hessian, d_block_offsets.data().get() + exec, streams);
    EXPECT_EQ(exec, 5u);
    const std::vector<float> hessian = d_hessian.to_host();
    const float want[12] = {5, 8, 8, 13, /* block (0,1): binary */ 3, 6, 4, 8, /* block (1,1) */ 13, 18, 18, 25};
    for (int k = 0; k < 12; ++k) EXPECT_FLOAT_EQ(hessian[k], want[k]);
    // ... and the same graph through Graph + Hessian::build_structure / update_values (hessian.hpp:257-307): the same three blocks at
    // offsets 0 (0,0), 4 (0,1), 8 (1,1) — column-major order of the block keys, the diagonal block last in its column — same 12 values
    Graph<float, float> graph;
    graph.scale_system(false);
    graph.add_descriptor(&vd); graph.add_descriptor(&unary); graph.add_descriptor(&binary);
    EXPECT_EQ(graph.initialize_optimization(0), true);
    graph.linearize(streams);
    Hessian<float, float> H;
    H.build_structure(&graph, streams);
    H.update_values(&graph, streams);
    EXPECT_EQ(H.host_col_pointers().size(), 3u);
    EXPECT_EQ(H.host_col_pointers()[1], 1u); EXPECT_EQ(H.host_col_pointers()[2], 3u);
    EXPECT_EQ(H.block_offset(0, 0), 0u); EXPECT_EQ(H.block_offset(0, 1), 4u); EXPECT_EQ(H.block_offset(1, 1), 8u);
    const std::vector<float> hv = H.get_values().to_host();
    EXPECT_EQ(hv.size(), 12u);
    for (int k = 0; k < 12 && k < (int)hv.size(); ++k) EXPECT_FLOAT_EQ(hv[k], want[k]);
  }
  { // AddAndReplaceVertex (vertex.cu:23-74)
    graphite::managed_vector<Vec2> vs; vs.reserve(3); vs.push_back(Vec2{1, 2}); vs.push_back(Vec2{3, 4}); vs.push_back(Vec2{9, 9});
    Vec2Descriptor vd; vd.add_vertex(10, &vs[0], false); vd.add_vertex(20, &vs[1], true);
    EXPECT_EQ(vd.count(), 2u); EXPECT_EQ(vd.exists(10), true); EXPECT_EQ(vd.exists(20), true);
    EXPECT_EQ(vd.is_fixed(10), false); EXPECT_EQ(vd.is_fixed(20), true); EXPECT_EQ(vd.is_active(10), true); EXPECT_EQ(vd.is_active(20), false);
    vd.replace_vertex(20, &vs[2]);
    EXPECT_EQ(vd.get_vertex(20), &vs[2]);
    EXPECT_FLOAT_EQ(vd.get_vertex(20)->x, 9.0f); EXPECT_FLOAT_EQ(vd.get_vertex(20)->y, 9.0f);
  }
  { // AugmentBlockDiagonal (vertex.cu:121-166): d + mu d on the active vertex's diagonal only
    graphite::managed_vector<Vec2> vs; vs.reserve(2); vs.push_back(Vec2{0, 0}); vs.push_back(Vec2{0, 0});
    Vec2Descriptor vd; vd.add_vertex(10, &vs[0], false); vd.add_vertex(20, &vs[1], true);
    graphite::managed_vector<float> blk, sd; blk.resize(8); sd.resize(4);
    for (size_t i = 0; i < blk.size(); ++i) blk[i] = -1.0f;
    sd[0] = 2.0f; sd[1] = 4.0f; sd[2] = 8.0f; sd[3] = 16.0f;
    const float mu = 0.5f;
    vd.augment_block_diagonal_async(blk.data().get(), sd.data().get(), mu, false, 0); dsync();
    EXPECT_FLOAT_EQ(blk[0], sd[0] + mu * sd[0]); EXPECT_FLOAT_EQ(blk[1], -1.0f); EXPECT_FLOAT_EQ(blk[2], -1.0f); EXPECT_FLOAT_EQ(blk[3], sd[1] + mu * sd[1]);
    for (int k = 4; k < 8; ++k) EXPECT_FLOAT_EQ(blk[k], -1.0f); // the fixed vertex's block: untouched
  }
  { // ApplyBlockJacobi (vertex.cu:168-226): z = block * r on the active vertex
    graphite::managed_vector<Vec2> vs; vs.reserve(2); vs.push_back(Vec2{0, 0}); vs.push_back(Vec2{0, 0});
    Vec2Descriptor vd; vd.add_vertex(10, &vs[0], false); vd.add_vertex(20, &vs[1], true);
    vd.set_hessian_column(10, 0, 0); vd.set_hessian_column(20, 2, 1); vd.to_device();
    graphite::managed_vector<float> z, r, blk; z.resize(4); r.resize(4); blk.resize(8);
    for (size_t i = 0; i < z.size(); ++i) z[i] = -5.0f;
    r[0] = 11.0f; r[1] = 13.0f; r[2] = 17.0f; r[3] = 19.0f;
    blk[0] = 2.0f; blk[1] = 3.0f; blk[2] = 5.0f; blk[3] = 7.0f; // [2 5; 3 7] column-major
    blk[4] = 101.0f; blk[5] = 103.0f; blk[6] = 107.0f; blk[7] = 109.0f;
    vd.apply_block_jacobi(z.data().get(), r.data().get(), blk.data().get(), 0); dsync();
    EXPECT_FLOAT_EQ(z[0], blk[0] * r[0] + blk[2] * r[1]); EXPECT_FLOAT_EQ(z[1], blk[1] * r[0] + blk[3] * r[1]);
    EXPECT_FLOAT_EQ(z[2], -5.0f); EXPECT_FLOAT_EQ(z[3], -5.0f);
  }
  for (size_t gone : {10u, 20u, 30u}) { // RemoveVertexFromStart / Middle / End (vertex.cu:228-297)
    graphite::managed_vector<Vec2> vs; vs.reserve(3); vs.push_back(Vec2{0, 0}); vs.push_back(Vec2{1, 1}); vs.push_back(Vec2{2, 2});
    Vec2Descriptor vd; vd.add_vertex(10, &vs[0], false); vd.add_vertex(20, &vs[1], false); vd.add_vertex(30, &vs[2], true);
    EXPECT_EQ(vd.count(), 3u);
    vd.remove_vertex(gone);
    EXPECT_EQ(vd.count(), 2u);
    EXPECT_EQ(vd.exists(10), gone != 10); EXPECT_EQ(vd.exists(20), gone != 20); EXPECT_EQ(vd.exists(30), gone != 30);
    if (gone != 10) { EXPECT_EQ(vd.get_vertex(10), &vs[0]); EXPECT_EQ(vd.is_fixed(10), false); EXPECT_EQ(vd.is_active(10), true); }
    if (gone != 20) { EXPECT_EQ(vd.get_vertex(20), &vs[1]); EXPECT_EQ(vd.is_fixed(20), false); EXPECT_EQ(vd.is_active(20), true); }
    if (gone != 30) { EXPECT_EQ(vd.get_vertex(30), &vs[2]); EXPECT_EQ(vd.is_fixed(30), true); EXPECT_EQ(vd.is_active(30), false); }
    const auto &map = vd.get_global_map();
    EXPECT_EQ(map.size(), 2u);
    for (size_t id : {10u, 20u, 30u}) if (id != gone) EXPECT_EQ(map.at(id) < vd.count(), true);
  }
  { // managed_vector (vector.cu:6-79)
    graphite::managed_vector<int> v;
    EXPECT_EQ(v.size(), 0u); EXPECT_EQ(v.capacity(), 0u); EXPECT_EQ(v.data().get(), (int *)nullptr);
    v.push_back(10); v.push_back(20); v.push_back(30);
    EXPECT_EQ(v.size(), 3u); EXPECT_EQ(v.capacity() >= 3u, true);
    EXPECT_EQ(v[0], 10); EXPECT_EQ(v[1], 20); EXPECT_EQ(v[2], 30); EXPECT_EQ(v.back(), 30);
    v.pop_back();
    EXPECT_EQ(v.size(), 2u); EXPECT_EQ(v.back(), 20);
    v.pop_back(); v.pop_back(); v.pop_back(); // the last one: a no-op on an empty vector
    EXPECT_EQ(v.size(), 0u);
    graphite::managed_vector<int> w; w.push_back(3); w.push_back(7);
    w.reserve(16);
    EXPECT_EQ(w.size(), 2u); EXPECT_EQ(w.capacity(), 16u); EXPECT_EQ(w[0], 3); EXPECT_EQ(w[1], 7);
    graphite::managed_vector<int> u; u.push_back(1); u.push_back(2);
    u.resize(8);
    EXPECT_EQ(u.size(), 8u); EXPECT_EQ(u.capacity() >= 8u, true);
    u.resize(1);
    EXPECT_EQ(u.size(), 1u); EXPECT_EQ(u[0], 1);
    const size_t cap = u.capacity();
    u.clear();
    EXPECT_EQ(u.size(), 0u); EXPECT_EQ(u.capacity(), cap);
    graphite::managed_vector<float> fv; fv.push_back(1.5f); fv.push_back(2.5f);
    EXPECT_EQ(fv.data().get() != nullptr, true); EXPECT_EQ(fv.begin(), fv.data().get());
    EXPECT_EQ(fv.end(), fv.begin() + static_cast<ptrdiff_t>(fv.size())); EXPECT_FLOAT_EQ(*(fv.begin() + 1), 2.5f);
  }
  std::cout << (failures ? "FAILED" : "OK") << " (" << failures << " failures, " << checks << " checks)" << std::endl;
  return failures != 0;
}
