// graphite/engine_model.hpp — the engine of libgraphite_mi355x.so instantiated on the USER's factor / vertex traits.
//
// The reference compiles every hot kernel against the user's traits (ops/linearize.hpp:10-138 compute_jacobians,
// ops/error.hpp:253-323 compute_error, ops/chi2.hpp:34-44, ops/product.hpp:103,292 the recompute forms, ops/update.hpp:11-31)
// and carries a precision matrix and a loss object per factor through all of them (factor.hpp:158-174).  This header does
// the same for the hand-written engine: the kernels below are the engine's per-observation kernels (graphite_amd/csrc/
// kernels_mf.hpp: camera-major order, one observation per lane, wave transpose-reductions into (wave, camera) segments,
// landmark records at point-major slots, XCD-aware persistent grids) with the built-in camera model replaced by calls of
// Traits::error / Traits::jacobian (or dual numbers) / Loss / the factor's precision matrix / Traits::update.  They are
// compiled by hipcc in the user's translation unit and handed to the library as a table of launchers
// (include/graphite_mi355x_model.h); everything that does not see the traits stays in the library.
//
// Fits: ONE binary factor type, slot 0 a non-eliminated "pose" vertex type of tangent dimension <= 9, slot 1 a "landmark"
// vertex type of dimension <= 3, error dimension <= 2, plain-data vertex / state / observation / data / loss types, symmetric
// positive semi-definite precision matrices.  Anything else stays on the generic kernels (core.hpp).
#pragma once
#include "core.hpp"
#include "../graphite_mi355x_model.h"
#include "../graphite_mi355x_device.hpp"

namespace graphite {
namespace detail {

// type-erased handle the optimiser (solve.hpp) keeps next to the library problem
class EngineModelBase {
public:
  virtual ~EngineModelBase() = default;
  virtual const gr_model_ops *ops() const = 0;
  // engine vertex -> descriptor-local vertex id (empty = identity), engine landmark order and observation order of the library
  virtual void bind(const std::vector<uint32_t> &cam_used, const std::vector<uint32_t> &pt_used, const std::vector<int32_t> &obs_order,
                    const std::vector<int32_t> &landmark_order) = 0;
  virtual void upload_vertices() = 0;   // user's vertex objects -> engine-order copies in HBM
  virtual void download_vertices() = 0; // and back
};

// K directional derivatives carried at once as the dual part of Dual<T, Tangent<T, K>>: the user's error function is evaluated
// ceil((pose_dim + landmark_dim) / K) times instead of once per column (ops/linearize.hpp:43-79 seeds one column per evaluation;
// every component below goes through the same operations in the same order as that column's own evaluation would).
// GRAPHITE_ENGINE_TANGENT_WIDTH: 0 (default) = ALL columns in one evaluation (K = pose_dim + landmark_dim <= 12; measured on the
// Ladybug-1723 shape, fp64, no spills at 180-190 VGPRs: k3 factor 99 / 89 / 80 us per linearisation at K = 4 / 6 / 12, pinhole 89 / 81 / 72);
// a smaller K for an error function whose temporaries do not fit; 1 keeps the one-column form (for error functions that name
// Dual<T, T> explicitly).
#ifndef GRAPHITE_ENGINE_TANGENT_WIDTH
#define GRAPHITE_ENGINE_TANGENT_WIDTH 0
#endif
template <typename T, int K> struct Tangent {
  T v[K];
  hd_fn Tangent() {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = T(0);
  }
  hd_fn Tangent(T s) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = s;
  }
#define GRAPHITE_TANGENT_OP(expr) Tangent r; _Pragma("unroll") for (int k = 0; k < K; ++k) r.v[k] = expr; return r;
  hd_fn friend Tangent operator+(const Tangent &a, const Tangent &b) { GRAPHITE_TANGENT_OP(a.v[k] + b.v[k]) }
  hd_fn friend Tangent operator-(const Tangent &a, const Tangent &b) { GRAPHITE_TANGENT_OP(a.v[k] - b.v[k]) }
  hd_fn friend Tangent operator-(const Tangent &a) { GRAPHITE_TANGENT_OP(-a.v[k]) }
  hd_fn friend Tangent operator*(T s, const Tangent &a) { GRAPHITE_TANGENT_OP(s * a.v[k]) }
  hd_fn friend Tangent operator*(const Tangent &a, T s) { GRAPHITE_TANGENT_OP(a.v[k] * s) }
  hd_fn friend Tangent operator/(const Tangent &a, T s) { GRAPHITE_TANGENT_OP(a.v[k] / s) }
#undef GRAPHITE_TANGENT_OP
};
} // namespace detail
} // namespace graphite
namespace std { // Dual::operator/ returns numeric_limits<D>::infinity() as the dual part of a division by zero (core.hpp)
template <typename T, int K> struct numeric_limits<graphite::detail::Tangent<T, K>> {
  static constexpr bool is_specialized = true;
  hd_fn static graphite::detail::Tangent<T, K> infinity() { return graphite::detail::Tangent<T, K>(numeric_limits<T>::infinity()); }
};
} // namespace std
namespace graphite {
namespace detail {

template <typename F> struct ModelView {
  using V0 = slot_vertex<F, 0>;
  using V1 = slot_vertex<F, 1>;
  const V0 *camv;
  const V1 *ptv;
  const typename F::ObservationType *obs;    // the factor streams, in the engine's observation order
  const typename F::ConstraintDataType *data;
  const typename F::LossType *loss;
  const typename F::Storage *pmat;           // [No][E * E] read row-major (ops/linearize.hpp:283); nullptr = identity
};

// value of S-typed storage as the kernels use it (types.hpp:8-43: a stored Jacobian entry is rounded to S once)
template <typename T, typename S> __device__ __forceinline__ T em_round(T v) {
  if constexpr (std::is_same<T, S>::value) return v; else return (T)(S)v;
}

// One observation: the user's residual e (E, zero-padded to 2), chi2 = e^T P e, rho / rho' of the factor's loss and, with
// WITH_J, the WEIGHTED blocks Jt = L J (pose 2 x 9, landmark 2 x 3, E x d column-major, zero-padded) and et = L e with
// rho' P = L^T L.  Returns rho(chi2).
template <typename F, bool WITH_J>
__device__ __forceinline__ typename F::Scalar em_evaluate(const ModelView<F> &mv, int j, int c, int l, typename F::Scalar (&et)[2],
                                                            typename F::Scalar (&Jc)[18], typename F::Scalar (&Jp)[6]) {
  using T = typename F::Scalar;
  using S = typename F::Storage;
  constexpr int E = (int)F::E, DC = (int)slot_dim<F, 0>(), DL = (int)slot_dim<F, 1>();
  constexpr auto seq = std::make_index_sequence<2>{};
  const auto v = std::make_tuple(const_cast<slot_vertex<F, 0> *>(mv.camv + c), const_cast<slot_vertex<F, 1> *>(mv.ptv + l));
  std::tuple<T[DC], T[DL]> p;
  slot_traits<F, 0>::parameters(*std::get<0>(v), (T *)std::get<0>(p));
  slot_traits<F, 1>::parameters(*std::get<1>(v), (T *)std::get<1>(p));
  T e[E];
  call_error<F, T>(v, p, mv.obs[j], mv.data[j], e, seq);
  // chi2 = e^T P e as ops/chi2.hpp:10-44 sums it
  T P[2][2] = {{T(1), T(0)}, {T(0), T(1)}};
  if (mv.pmat) {
#pragma unroll
    for (int i = 0; i < E; ++i)
#pragma unroll
      for (int k = 0; k < E; ++k) P[i][k] = (T)mv.pmat[(size_t)j * E * E + i * E + k];
  }
  if (E == 1) { P[0][1] = P[1][0] = P[1][1] = T(0); }
  T value = 0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    T r2 = 0;
#pragma unroll
    for (int k = 0; k < E; ++k) r2 += P[i][k] * e[k];
    value += r2 * e[i];
  }
  const T rho = mv.loss[j].loss(value);
  if constexpr (WITH_J) {
    const T w = mv.loss[j].loss_derivative(value);
    // W = w P = L^T L, L upper (positive semi-definite safe)
    const T w00 = w * P[0][0], w01 = w * P[0][1], w11 = w * P[1][1];
    const T l00 = w00 > T(0) ? (T)sqrt((double)w00) : T(0);
    const T l01 = l00 > T(0) ? w01 / l00 : T(0);
    const T d11 = w11 - l01 * l01;
    const T l11 = d11 > T(0) ? (T)sqrt((double)d11) : T(0);
    T jc[E * DC], jp[E * DL];
#pragma unroll
    for (int i = 0; i < E * DC; ++i) jc[i] = T(0); // the reference zero-fills the storage before the user function writes (ops/linearize.hpp:127)
#pragma unroll
    for (int i = 0; i < E * DL; ++i) jp[i] = T(0);
    if constexpr (std::is_same<typename F::Traits::Differentiation, DifferentiationMode::Manual>::value) {
      call_jacobian_t<F, 0, T>(v, mv.obs[j], mv.data[j], jc, seq);
      call_jacobian_t<F, 1, T>(v, mv.obs[j], mv.data[j], jp, seq);
    } else if constexpr (GRAPHITE_ENGINE_TANGENT_WIDTH != 1) { // dual numbers, K columns per evaluation (Tangent above)
      constexpr int K = GRAPHITE_ENGINE_TANGENT_WIDTH > 1 ? GRAPHITE_ENGINE_TANGENT_WIDTH : DC + DL, NPASS = (DC + DL + K - 1) / K;
      using D = Dual<T, Tangent<T, K>>;
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        std::tuple<D[DC], D[DL]> pd;
        slot_traits<F, 0>::parameters(*std::get<0>(v), (D *)std::get<0>(pd));
        slot_traits<F, 1>::parameters(*std::get<1>(v), (D *)std::get<1>(pd));
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int col = pass * K + k; // a constant after unrolling
          if (col < DC) std::get<0>(pd)[col < DC ? col : 0].dual.v[k] = T(1);
          else if (col < DC + DL) std::get<1>(pd)[col < DC + DL ? col - DC : 0].dual.v[k] = T(1);
        }
        D ed[E];
        call_error<F, D>(v, pd, mv.obs[j], mv.data[j], ed, seq);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int col = pass * K + k;
#pragma unroll
          for (int i = 0; i < E; ++i) {
            if (col < DC) jc[(col < DC ? col : 0) * E + i] = ed[i].dual.v[k];
            else if (col < DC + DL) jp[(col < DC + DL ? col - DC : 0) * E + i] = ed[i].dual.v[k];
          }
        }
      }
    } else { // dual numbers, one column at a time (ops/linearize.hpp:43-79 with Dual<T>)
      using D = Dual<T, T>;
#pragma unroll 1
      for (int col = 0; col < DC + DL; ++col) {
        std::tuple<D[DC], D[DL]> pd;
        slot_traits<F, 0>::parameters(*std::get<0>(v), (D *)std::get<0>(pd));
        slot_traits<F, 1>::parameters(*std::get<1>(v), (D *)std::get<1>(pd));
        if (col < DC) std::get<0>(pd)[col].dual = T(1); else std::get<1>(pd)[col - DC].dual = T(1);
        D ed[E];
        call_error<F, D>(v, pd, mv.obs[j], mv.data[j], ed, seq);
#pragma unroll
        for (int i = 0; i < E; ++i) { if (col < DC) jc[col * E + i] = ed[i].dual; else jp[(col - DC) * E + i] = ed[i].dual; }
      }
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) Jc[i] = T(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) Jp[i] = T(0);
#pragma unroll
    for (int col = 0; col < DC; ++col) {
      const T a0 = em_round<T, S>(jc[col * E]), a1 = E > 1 ? em_round<T, S>(jc[col * E + (E > 1 ? 1 : 0)]) : T(0);
      Jc[2 * col] = l00 * a0 + l01 * a1; Jc[2 * col + 1] = l11 * a1;
    }
#pragma unroll
    for (int col = 0; col < DL; ++col) {
      const T a0 = em_round<T, S>(jp[col * E]), a1 = E > 1 ? em_round<T, S>(jp[col * E + (E > 1 ? 1 : 0)]) : T(0);
      Jp[2 * col] = l00 * a0 + l01 * a1; Jp[2 * col + 1] = l11 * a1;
    }
    const T e0 = e[0], e1 = E > 1 ? e[E > 1 ? 1 : 0] : T(0);
    et[0] = l00 * e0 + l01 * e1; et[1] = l11 * e1;
  } else {
    et[0] = e[0]; et[1] = E > 1 ? e[E > 1 ? 1 : 0] : T(0);
  }
  return rho;
}

// Graph::linearize + Hessian::update_values for the model's factors: the counterpart of k_linearize (kernels_mf.hpp)
//   camera side : 45 + 9 sums per (wave, camera) segment -> cam_partial[seg][54]
//   point side  : [Jt_l (6), et (2)] per observation       -> g9[pm slot][8]
//   Hcp^u       : 9 x 3 block per observation               -> Hcp[pm slot][27]   (Schur solvers)
//   stored Jt   : 24 streams in observation order           -> jst                 (stored-Jacobian operator)
//   chi2        : one partial per workgroup
template <typename F, bool WRITE_HCP, bool STORE>
__global__ void __launch_bounds__(256) k_em_linearize(gr_model_lin_args a, ModelView<F> mv) {
  using T = typename F::Scalar;
  using S = typename F::Storage;
  using V2 = typename gr::Vec2T<T>::type;
  constexpr int DC = (int)slot_dim<F, 0>(), DL = (int)slot_dim<F, 1>();
  if (a.lm && static_cast<const gr::LmDev *>(a.lm)->stop) return;
  if (a.gate && !*a.gate) return;
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  int j0, jstride, niter, jlim;
  gr::xcd_obs_range(a.ntiles, a.No, j0, jstride, niter, jlim);
  T *const g9 = static_cast<T *>(a.g9), *const cam_partial = static_cast<T *>(a.cam_partial);
  T *const Hcp = static_cast<T *>((WRITE_HCP && a.hsel && (static_cast<const gr::LmDev *>(a.hsel)->hsel & 1) == 0) ? a.Hcp_alt : a.Hcp);
  double chi2 = 0.0;
  int j = j0 + threadIdx.x;
  bool valid = niter > 0 && j < jlim;
  int c_n = -1, l_n = 0, a_n = 0;
  if (valid) { c_n = a.cam[j]; l_n = a.pt[j]; a_n = a.pos[j]; }
  for (int it = 0; it < niter; ++it) {
    const int c = c_n, l = l_n;
    const size_t slot = (size_t)a_n;
    const int jn = j + jstride;
    const bool validn = (it + 1 < niter) && jn < jlim;
    if (validn) { c_n = a.cam[jn]; l_n = a.pt[jn]; a_n = a.pos[jn]; }
    T Jc[18], Jp[6], et[2] = {T(0), T(0)};
#pragma unroll
    for (int i = 0; i < 18; ++i) Jc[i] = T(0);
    if (valid) {
      chi2 += (double)em_evaluate<F, true>(mv, j, c, l, et, Jc, Jp);
      V2 *g = reinterpret_cast<V2 *>(g9 + 8 * slot);
      V2 q0, q1, q2, q3;
      q0.x = Jp[0]; q0.y = Jp[1]; q1.x = Jp[2]; q1.y = Jp[3]; q2.x = Jp[4]; q2.y = Jp[5]; q3.x = et[0]; q3.y = et[1];
      g[0] = q0; g[1] = q1; g[2] = q2; g[3] = q3;
      if constexpr (WRITE_HCP) {
        T *h = Hcp + 27 * slot;
        const T keep = ((a.cam_fixed && a.cam_fixed[c]) || (a.pt_fixed && a.pt_fixed[l])) ? T(0) : T(1);
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          h[r] = keep * (Jc[2 * r] * Jp[0] + Jc[2 * r + 1] * Jp[1]);
          h[r + 9] = keep * (Jc[2 * r] * Jp[2] + Jc[2 * r + 1] * Jp[3]);
          h[r + 18] = keep * (Jc[2 * r] * Jp[4] + Jc[2 * r + 1] * Jp[5]);
        }
      }
      if constexpr (STORE) { // only the streams of real rows / columns: the padded ones were zeroed once by the library
        S *q = static_cast<S *>(a.jst) + j;
#pragma unroll
        for (int i = 0; i < 2 * DC; ++i) q[(long long)i * a.jst_stride] = (S)Jc[i];
#pragma unroll
        for (int i = 0; i < 2 * DL; ++i) q[(long long)(18 + i) * a.jst_stride] = (S)Jp[i];
      }
    }
    // camera-side reduction, once per distinct camera in the wave (usually one)
    unsigned long long remaining = __ballot(valid);
    int segf = a.blk_seg[__builtin_amdgcn_readfirstlane(j >> 6)];
    while (remaining) {
      const int leader = __builtin_ctzll(remaining);
      const int cl = __builtin_amdgcn_readlane(c, leader);
      const int segl = a.seg_slot[segf++];
      const bool mine = valid && c == cl;
      const T wm = mine ? T(1) : T(0);
      T acc[64];
      int kk = 0;
#pragma unroll
      for (int col = 0; col < 9; ++col) {
        const T wx = wm * Jc[2 * col], wy = wm * Jc[2 * col + 1];
#pragma unroll
        for (int row = 0; row <= col; ++row) acc[kk++] = Jc[2 * row] * wx + Jc[2 * row + 1] * wy;
        acc[45 + col] = -(wx * et[0] + wy * et[1]);
      }
#pragma unroll
      for (int i = 54; i < 64; ++i) acc[i] = T(0);
      const T tot = gr::wave_transpose_sum<T, 64>(acc, lane);
      if (lane < 54) cam_partial[54 * (size_t)segl + lane] = tot;
      remaining &= ~__ballot(mine);
    }
    valid = validn;
    j = jn;
  }
  chi2 = gr::block_sum_256(chi2, red);
  if (threadIdx.x == 0) a.chi2_partial[blockIdx.x] = chi2;
}

// Graph::compute_error + Graph::chi2 (graph.hpp:212-225): one partial per workgroup, residuals on request
template <typename F> __global__ void __launch_bounds__(256) k_em_chi2(gr_model_chi2_args a, ModelView<F> mv) {
  using T = typename F::Scalar;
  __shared__ double red[4];
  double chi2 = 0;
  T *const res = static_cast<T *>(a.res_out);
  for (int j = blockIdx.x * 256 + threadIdx.x; j < a.No; j += gridDim.x * 256) {
    T e[2], Jc[18], Jp[6];
    chi2 += (double)em_evaluate<F, false>(mv, j, a.cam[j], a.pt[j], e, Jc, Jp);
    if (res) { const size_t s = (size_t)a.pos[j]; res[2 * s] = e[0]; res[2 * s + 1] = e[1]; }
  }
  chi2 = gr::block_sum_256(chi2, red);
  if (threadIdx.x == 0) a.chi2_partial[blockIdx.x] = chi2;
}

// Matrix-free operator with the blocks RECOMPUTED through the user's jacobian<> (set_jacobian_storage(false);
// ops/product.hpp:103,292 compute_Jv_dynamic / compute_Jtv_dynamic): the counterpart of k_pcg_operator
template <typename F> __global__ void __launch_bounds__(256) k_em_operator(gr_model_op_args a, ModelView<F> mv) {
  using T = typename F::Scalar;
  if (a.lm && static_cast<const gr::LmDev *>(a.lm)->stop) return;
  if (a.done && *a.done) return;
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  const T *const ps = static_cast<const T *>(a.ps);
  T *const g3 = static_cast<T *>(a.g3), *const op_partial = static_cast<T *>(a.op_partial);
  const size_t pose_dim = 9 * (size_t)a.Nc;
  int j0, jstride, niter, jlim;
  gr::xcd_obs_range(a.ntiles, a.No, j0, jstride, niter, jlim);
  double den = 0;
  int j = j0 + threadIdx.x;
  bool valid = niter > 0 && j < jlim;
  int c_n = -1, l_n = 0, a_n = 0;
  if (valid) { c_n = a.cam[j]; l_n = a.pt[j]; a_n = a.pos ? a.pos[j] : j; }
  for (int it = 0; it < niter; ++it) {
    const int c = c_n, l = l_n;
    const size_t slot = (size_t)a_n;
    const int jn = j + jstride;
    const bool validn = (it + 1 < niter) && jn < jlim;
    if (validn) { c_n = a.cam[jn]; l_n = a.pt[jn]; a_n = a.pos ? a.pos[jn] : jn; }
    T Jc[18], Jp[6], et[2], pl0 = 0, pl1 = 0, pl2 = 0;
#pragma unroll
    for (int i = 0; i < 18; ++i) Jc[i] = T(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) Jp[i] = T(0);
    if (valid) {
      (void)em_evaluate<F, true>(mv, j, c, l, et, Jc, Jp);
      const T *pl = ps + pose_dim + 3 * (size_t)l;
      pl0 = pl[0]; pl1 = pl[1]; pl2 = pl[2];
    }
    const T up0 = Jp[0] * pl0 + Jp[2] * pl1 + Jp[4] * pl2, up1 = Jp[1] * pl0 + Jp[3] * pl1 + Jp[5] * pl2;
    unsigned long long remaining = __ballot(valid);
    int segf = a.blk_seg[__builtin_amdgcn_readfirstlane(j >> 6)];
    while (remaining) {
      const int leader = __builtin_ctzll(remaining);
      const int cl = __builtin_amdgcn_readlane(c, leader);
      const bool mine = valid && c == cl;
      const int segl = a.seg_slot[segf++];
      T u0 = up0, u1 = up1;
#pragma unroll
      for (int i = 0; i < 9; ++i) { const T pc = ps[9 * (size_t)cl + i]; u0 += Jc[2 * i] * pc; u1 += Jc[2 * i + 1] * pc; }
      if (mine) den += (double)(u0 * u0 + u1 * u1);
      T m[16];
#pragma unroll
      for (int i = 0; i < 9; ++i) m[i] = mine ? Jc[2 * i] * u0 + Jc[2 * i + 1] * u1 : T(0);
#pragma unroll
      for (int i = 9; i < 16; ++i) m[i] = T(0);
      if (mine) {
        T *g = g3 + 3 * slot;
        g[0] = Jp[0] * u0 + Jp[1] * u1; g[1] = Jp[2] * u0 + Jp[3] * u1; g[2] = Jp[4] * u0 + Jp[5] * u1;
      }
      const T tot = gr::wave_transpose_sum<T, 16>(m, lane);
      if ((lane & 3) == 0 && (lane >> 2) < 9) op_partial[9 * (size_t)segl + (lane >> 2)] = tot;
      remaining &= ~__ballot(mine);
    }
    valid = validn;
    j = jn;
  }
  den = gr::block_sum_256(den, red);
  if (threadIdx.x == 0) atomicAdd(&a.den_slots[blockIdx.x & 63], den);
}

// Graph::backup_parameters + Graph::apply_update (graph.hpp:292-309, ops/update.hpp:11-31) through Traits::update, plus
// the workgroup's share of compute_rho's denominator sum dx (mu dx + s b) (levenberg_marquardt.hpp:34-41).
// Workgroups [0, nbc): one thread per pose; the others: one thread per landmark.  PAD = the engine's block width (9 / 3).
template <typename Tr, typename T, int PAD>
__device__ __forceinline__ double em_step_one(typename Tr::Vertex &vtx, typename state_of<Tr>::type *bak, bool with_backup,
                                              const T *dx, const T *scales, const T *bu, double mu, bool count, bool restore_first = false) {
  constexpr int D = (int)Tr::dimension;
  if (restore_first) { if constexpr (state_of<Tr>::custom) Tr::set_state(vtx, *bak); else vtx = *bak; } // (a rejected trial point: back to the kept one; its backup stands)
  else if (with_backup) { if constexpr (state_of<Tr>::custom) *bak = Tr::get_state(vtx); else *bak = vtx; }
  T d[D];
  double rho = 0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const T x = dx[k], s = scales[k];
    d[k] = x * s; // ops/update.hpp:26
    if (count) rho += (double)(x * ((T)mu * x + s * bu[k]));
  }
  Tr::update(vtx, d);
  return rho;
}
template <typename F>
__global__ void __launch_bounds__(256) k_em_step(gr_model_step_args a, slot_vertex<F, 0> *camv, typename state_of<slot_traits<F, 0>>::type *cam_bak, int Nc,
                                                 slot_vertex<F, 1> *ptv, typename state_of<slot_traits<F, 1>>::type *pt_bak, int Np, int nbc) {
  using T = typename F::Scalar;
  if (a.lm && static_cast<const gr::LmDev *>(a.lm)->stop) return;
  if (a.gate && !*a.gate) return;
  if (a.lm) a.mu = static_cast<const gr::LmDev *>(a.lm)->mu; // device-decided loop: the damping the decision left
  const bool restore_first = a.restore_on_hsel && a.lm && (static_cast<const gr::LmDev *>(a.lm)->hsel & 2) != 0;
  __shared__ double red[4];
  const T *dx = static_cast<const T *>(a.dx), *sc = static_cast<const T *>(a.scales), *bu = static_cast<const T *>(a.bu);
#ifndef GR_NO_CLEAR
  { // the library's loop-state reset (gr_model_step_args::clear_ptr): a few thousand words, spread over the launch
    const uint32_t *const inf_hi = a.inf_word ? reinterpret_cast<const uint32_t *>(a.inf_word) + 1 : nullptr;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      uint32_t *const base = static_cast<uint32_t *>(a.clear_ptr[q]);
      const long long nw = a.clear_bytes[q] / 4;
      for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < nw; w += (long long)gridDim.x * 256) base[w] = (base + w == inf_hi) ? 0x7FF00000u : 0u;
    }
  }
#endif
  double rho = 0;
  // (a fixed vertex keeps its value: no update — but its state IS saved, the revert restores every vertex)
  if ((int)blockIdx.x < nbc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < Nc) {
      if (a.cam_fixed && a.cam_fixed[i]) { if (a.with_backup && !restore_first) { if constexpr (state_of<slot_traits<F, 0>>::custom) cam_bak[i] = slot_traits<F, 0>::get_state(camv[i]); else cam_bak[i] = camv[i]; } }
      else rho = em_step_one<slot_traits<F, 0>, T, 9>(camv[i], cam_bak + i, a.with_backup != 0, dx + 9 * (size_t)i, sc + 9 * (size_t)i, bu + 9 * (size_t)i, a.mu, a.cam_weight != 0, restore_first);
    }
  } else {
    const int i = (blockIdx.x - nbc) * 256 + threadIdx.x;
    const size_t o = 9 * (size_t)Nc + 3 * (size_t)i;
    if (i < Np) {
      if (a.pt_fixed && a.pt_fixed[i]) { if (a.with_backup && !restore_first) { if constexpr (state_of<slot_traits<F, 1>>::custom) pt_bak[i] = slot_traits<F, 1>::get_state(ptv[i]); else pt_bak[i] = ptv[i]; } }
      else rho = em_step_one<slot_traits<F, 1>, T, 3>(ptv[i], pt_bak + i, a.with_backup != 0, dx + o, sc + o, bu + o, a.mu, true, restore_first);
    }
  }
  rho = gr::block_sum_256(rho, red);
  if (threadIdx.x == 0 && a.rho_partial) a.rho_partial[blockIdx.x] = rho;
}
template <typename Tr, bool RESTORE>
__global__ void k_em_state(typename Tr::Vertex *v, typename state_of<Tr>::type *bak, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if constexpr (RESTORE) { if constexpr (state_of<Tr>::custom) Tr::set_state(v[i], bak[i]); else v[i] = bak[i]; }
  else { if constexpr (state_of<Tr>::custom) bak[i] = Tr::get_state(v[i]); else bak[i] = v[i]; }
}
// engine-order copies of the user's vertex objects: dst[k] = *src[sel[k]] (sel == nullptr: k), and back
template <typename V, bool OUT> __global__ void k_em_vertices(V *const *user, const uint32_t *sel, int n, V *copy) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  V *u = user[sel ? sel[k] : (uint32_t)k];
  if constexpr (OUT) *u = copy[k]; else copy[k] = *u;
}

// what the user-traits engine can represent at all (compile time)
template <typename F> constexpr bool em_representable() {
  if constexpr (F::N != 2) return false;
  else {
    using T = typename F::Scalar;
    using S = typename F::Storage;
    return slot_dim<F, 0>() >= 1 && slot_dim<F, 0>() <= 9 && slot_dim<F, 1>() >= 1 && slot_dim<F, 1>() <= 3 && F::E >= 1 && F::E <= 2 &&
           std::is_trivially_copyable<slot_vertex<F, 0>>::value && std::is_trivially_copyable<slot_vertex<F, 1>>::value &&
           std::is_trivially_copyable<typename state_of<slot_traits<F, 0>>::type>::value && std::is_trivially_copyable<typename state_of<slot_traits<F, 1>>::type>::value &&
           std::is_trivially_copyable<typename F::ObservationType>::value && std::is_trivially_copyable<typename F::ConstraintDataType>::value &&
           std::is_trivially_copyable<typename F::LossType>::value && !std::is_polymorphic<typename F::LossType>::value &&
           (std::is_same<S, T>::value || (std::is_same<T, double>::value && std::is_same<S, float>::value));
  }
}

template <typename F> class EngineModel final : public EngineModelBase {
public:
  using T = typename F::Scalar;
  using S = typename F::Storage;
  using Tr0 = slot_traits<F, 0>;
  using Tr1 = slot_traits<F, 1>;
  using V0 = slot_vertex<F, 0>;
  using V1 = slot_vertex<F, 1>;
  using St0 = typename state_of<Tr0>::type;
  using St1 = typename state_of<Tr1>::type;
  using VD0 = typename std::tuple_element<0, typename F::VDTuple>::type;
  using VD1 = typename std::tuple_element<1, typename F::VDTuple>::type;

  F *fd;
  VD0 *vd0;
  VD1 *vd1;
  hbm_vector<V0> camv;
  hbm_vector<St0> cam_bak;
  hbm_vector<V1> ptv;
  hbm_vector<St1> pt_bak;
  hbm_vector<uint32_t> cam_sel, pt_sel;
  bool cam_identity = true;
  hbm_vector<typename F::ObservationType> obs;
  // empty types still need one addressable element per factor for the kernels' references: a single shared dummy array of No bytes
  hbm_vector<typename F::ConstraintDataType> data;
  hbm_vector<typename F::LossType> loss;
  hbm_vector<S> pmat;
  bool identity_precision = true;
  int Nc = 0, Np = 0, No = 0, nbc = 0;
  gr_model_ops table{};

  EngineModel(F *fd_, bool any_pmat) : fd(fd_), identity_precision(!any_pmat) {
    vd0 = static_cast<VD0 *>(fd->typed_descriptors[0]);
    vd1 = static_cast<VD1 *>(fd->typed_descriptors[1]);
    table.ctx = this;
    table.pose_dim = (int32_t)slot_dim<F, 0>(); table.landmark_dim = (int32_t)slot_dim<F, 1>(); table.error_dim = (int32_t)F::E;
    table.storage_dtype = sizeof(S) == 8 ? GR_F64 : GR_F32;
    table.store_jacobians = fd->dynamic_jacobians() ? 0 : 1;
    table.linearize = &s_linearize; table.chi2 = &s_chi2; table.op = F::supports_dynamic_jacobians() ? &s_op : nullptr;
    table.step = &s_step; table.backup = &s_backup; table.revert = &s_revert;
    auto per_cu = [](const void *fn) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 2; }
      return std::min(nb, 8);
    };
    table.lin_wg_per_cu = table.store_jacobians ? per_cu(reinterpret_cast<const void *>(&k_em_linearize<F, false, true>)) : per_cu(reinterpret_cast<const void *>(&k_em_linearize<F, false, false>));
    table.op_wg_per_cu = per_cu(reinterpret_cast<const void *>(&k_em_operator<F>));
  }
  const gr_model_ops *ops() const override { return &table; }
  void set_counts(int nc, int np) { Nc = nc; Np = np; nbc = (nc + 255) / 256; table.step_blocks = nbc + (np + 255) / 256; }

  void bind(const std::vector<uint32_t> &cam_used, const std::vector<uint32_t> &pt_used, const std::vector<int32_t> &obs_order,
            const std::vector<int32_t> &landmark_order) override {
    No = (int)obs_order.size();
    cam_identity = cam_used.empty();
    if (!cam_identity) cam_sel.assign(cam_used.data(), cam_used.size());
    std::vector<uint32_t> ps(landmark_order.size());
    for (size_t q = 0; q < ps.size(); ++q) ps[q] = pt_used.empty() ? (uint32_t)landmark_order[q] : pt_used[landmark_order[q]];
    pt_sel.assign(ps.data(), ps.size());
    camv.resize_uninit(Nc); cam_bak.resize_uninit(Nc); ptv.resize_uninit(Np); pt_bak.resize_uninit(Np);
    // the factor streams in the engine's observation order: observation j is ACTIVE factor obs_order[j] (export order = active_indices order)
    const size_t na = (size_t)No;
    std::vector<typename F::ObservationType> ho(na);
    std::vector<typename F::ConstraintDataType> hd(na);
    std::vector<typename F::LossType> hl(na);
    std::vector<S> hp(identity_precision ? 0 : na * F::E * F::E);
    parallel_chunks(na, 1 << 15, [&](size_t b, size_t e, size_t) {
      for (size_t j = b; j < e; ++j) {
        const size_t f = fd->active_indices[(size_t)obs_order[j]];
        ho[j] = fd->device_obs[f]; hd[j] = fd->data[f]; hl[j] = fd->loss[f];
        if (!identity_precision) for (size_t q = 0; q < F::E * F::E; ++q) hp[j * F::E * F::E + q] = fd->precision_matrices[f * F::E * F::E + q];
      }
    });
    obs.assign(ho.data(), na); data.assign(hd.data(), na); loss.assign(hl.data(), na);
    if (!identity_precision) pmat.assign(hp.data(), hp.size());
  }
  void upload_vertices() override {
    k_em_vertices<V0, false><<<blocks(Nc), TPB>>>(vd0->vertices(), cam_identity ? nullptr : cam_sel.raw(), Nc, camv.raw());
    k_em_vertices<V1, false><<<blocks(Np), TPB>>>(vd1->vertices(), pt_sel.raw(), Np, ptv.raw());
    sync();
  }
  void download_vertices() override {
    k_em_vertices<V0, true><<<blocks(Nc), TPB>>>(vd0->vertices(), cam_identity ? nullptr : cam_sel.raw(), Nc, camv.raw());
    k_em_vertices<V1, true><<<blocks(Np), TPB>>>(vd1->vertices(), pt_sel.raw(), Np, ptv.raw());
    sync();
  }
  ModelView<F> view() const {
    ModelView<F> mv;
    mv.camv = camv.raw(); mv.ptv = ptv.raw(); mv.obs = obs.raw(); mv.data = data.raw(); mv.loss = loss.raw();
    mv.pmat = identity_precision ? nullptr : pmat.raw();
    return mv;
  }

private:
  static int s_linearize(void *ctx, const gr_model_lin_args *a) {
    auto *m = static_cast<EngineModel *>(ctx);
    const ModelView<F> mv = m->view();
    const dim3 g((unsigned)a->grid), b(256);
    hipStream_t s = static_cast<hipStream_t>(a->stream);
    if (a->Hcp) { if (a->jst) k_em_linearize<F, true, true><<<g, b, 0, s>>>(*a, mv); else k_em_linearize<F, true, false><<<g, b, 0, s>>>(*a, mv); }
    else { if (a->jst) k_em_linearize<F, false, true><<<g, b, 0, s>>>(*a, mv); else k_em_linearize<F, false, false><<<g, b, 0, s>>>(*a, mv); }
    return (int)hipGetLastError();
  }
  static int s_chi2(void *ctx, const gr_model_chi2_args *a) {
    auto *m = static_cast<EngineModel *>(ctx);
    k_em_chi2<F><<<dim3((unsigned)a->grid), dim3(256), 0, static_cast<hipStream_t>(a->stream)>>>(*a, m->view());
    return (int)hipGetLastError();
  }
  static int s_op(void *ctx, const gr_model_op_args *a) {
    if constexpr (F::supports_dynamic_jacobians()) {
      auto *m = static_cast<EngineModel *>(ctx);
      k_em_operator<F><<<dim3((unsigned)a->grid), dim3(256), 0, static_cast<hipStream_t>(a->stream)>>>(*a, m->view());
      return (int)hipGetLastError();
    } else { (void)ctx; (void)a; return (int)hipErrorNotSupported; }
  }
  static int s_step(void *ctx, const gr_model_step_args *a) {
    auto *m = static_cast<EngineModel *>(ctx);
    k_em_step<F><<<dim3((unsigned)m->table.step_blocks), dim3(256), 0, static_cast<hipStream_t>(a->stream)>>>(*a, m->camv.raw(), m->cam_bak.raw(), m->Nc, m->ptv.raw(), m->pt_bak.raw(), m->Np, m->nbc);
    return (int)hipGetLastError();
  }
  static int s_backup(void *ctx, void *stream) {
    auto *m = static_cast<EngineModel *>(ctx);
    k_em_state<Tr0, false><<<blocks(m->Nc), TPB, 0, static_cast<hipStream_t>(stream)>>>(m->camv.raw(), m->cam_bak.raw(), m->Nc);
    k_em_state<Tr1, false><<<blocks(m->Np), TPB, 0, static_cast<hipStream_t>(stream)>>>(m->ptv.raw(), m->pt_bak.raw(), m->Np);
    return (int)hipGetLastError();
  }
  static int s_revert(void *ctx, void *stream) {
    auto *m = static_cast<EngineModel *>(ctx);
    k_em_state<Tr0, true><<<blocks(m->Nc), TPB, 0, static_cast<hipStream_t>(stream)>>>(m->camv.raw(), m->cam_bak.raw(), m->Nc);
    k_em_state<Tr1, true><<<blocks(m->Np), TPB, 0, static_cast<hipStream_t>(stream)>>>(m->ptv.raw(), m->pt_bak.raw(), m->Np);
    return (int)hipGetLastError();
  }
};

} // namespace detail

template <typename T, typename S, typename FTraits>
std::shared_ptr<detail::EngineModelBase> FactorDescriptor<T, S, FTraits>::make_engine_model(std::vector<int32_t> &cam, std::vector<int32_t> &pt, size_t num_poses, size_t num_landmarks) {
  if constexpr (detail::em_representable<FactorDescriptor>()) {
    const size_t na = active_count();
    if (!na || na >= (size_t)std::numeric_limits<int32_t>::max() / 27) return nullptr;
    detail::sync();
    cam.resize(na); pt.resize(na);
    std::atomic<bool> ok{true}, any{false};
    detail::parallel_chunks(na, 1 << 15, [&](size_t b, size_t e, size_t) {
      for (size_t a = b; a < e; ++a) {
        const size_t f = active_indices[a];
        cam[a] = (int32_t)device_ids[2 * f]; pt[a] = (int32_t)device_ids[2 * f + 1];
        const S *P = precision_matrices.raw() + f * E * E;
        if constexpr (E == 1) { if (!((double)P[0] >= 0.0)) ok = false; if (P[0] != S(1)) any = true; }
        else {
          const double p00 = (double)P[0], p01 = (double)P[1], p10 = (double)P[2], p11 = (double)P[3];
          if (!(p01 == p10 && p00 >= 0.0 && p11 >= 0.0 && p00 * p11 - p01 * p01 >= -1e-12 * std::abs(p00 * p11))) ok = false; // W = L^T L needs it
          if (!(p00 == 1.0 && p01 == 0.0 && p11 == 1.0)) any = true;
        }
      }
    });
    if (!ok.load()) return nullptr;
    auto m = std::make_shared<detail::EngineModel<FactorDescriptor>>(this, any.load());
    m->set_counts((int)num_poses, (int)num_landmarks);
    return m;
  } else { (void)cam; (void)pt; (void)num_poses; (void)num_landmarks; return nullptr; }
}

} // namespace graphite
