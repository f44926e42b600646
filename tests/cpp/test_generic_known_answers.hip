// The literal known answers of the reference's own unit tests for the generic per-factor kernels
// (tests/factor.cu:139-157, 296-322, 360-423, 425-509, 511-595, 597-756, 758-784 and
// tests/vertex.cu:76-119, 299-341), replayed on the HIP generic layer: same fixtures (one Vec2 vertex
// at (7, 0), observation 2.5, J = [1 0] or [2 3], Huber delta 1), same expected numbers, float with
// EXPECT_FLOAT_EQ's 4-ULP bar.  The CPU oracle replays the same vectors (tests/test_oracle_known_answers.py).
#include <cmath>
#include <cstring>
#include <graphite/factor.hpp>
#include <graphite/graph.hpp>
#include <iostream>

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
  std::cout << (failures ? "FAILED" : "OK") << " (" << failures << " failures, " << checks << " checks)" << std::endl;
  return failures != 0;
}
