// ORACLE — test infrastructure only (see bal_model.hpp header).
//
// CPU restatement of the reference's hot path on a BAL graph
// (cameras d=9 not eliminated, points d=3 eliminated, reprojection factors E=2,
// identity precision, DefaultLoss or HuberLoss), stage by stage:
//   Graph::initialize_optimization ordering   graph.hpp:100-149
//   Graph::linearize                          graph.hpp:236-290
//   Hessian::update_values / apply_damping    hessian.hpp:290-307, 136-176
//   SchurComplement::update_values            schur.hpp:227-235 (+ :587, :1067, :649, :901)
//   SchurComplement::compute_landmark_update  schur.hpp:279-302
//   execute_schur_vector_multiply             schur.hpp:347-393
//   BlockJacobiSchurPreconditioner            preconditioner/block_jacobi_schur.hpp:114-178
//   PCGSchurSolver::solve                     solver/pcg_schur.hpp:79-168
//   BlockJacobiPreconditioner                 preconditioner/block_jacobi.hpp:79-186
//   PCGSolver::solve                          solver/pcg.hpp:61-232
//   EigenLDLTSolver / EigenSchurLDLTSolver    solver/eigen.hpp:49-98, eigen_schur.hpp:52-108
//   levenberg_marquardt / compute_rho         optimizer/levenberg_marquardt.hpp:20-47,110-242
// Everything lives in the column-SCALED space exactly as the reference does it
// (scale stored J in place, then multiply), per-vertex sums sequential in factor order.
// The GLOBAL reductions (chi2, the PCG dot products, compute_rho's denominator) are
// thrust::reduce / thrust::inner_product in the reference (ops/chi2.hpp:61-64,
// ops/vector.hpp), i.e. tree reductions in T whose order is not pinned; they are restated
// as pairwise (tree) sums in T.  A sequential fp32 accumulation of the 5 M chi2 terms of
// Venice-1778 is off by 0.4 %, which no tree reduction is.
#pragma once
#include "bal_model.hpp"
#include "user_models.hpp"
#include "generic_ops.hpp"
#include "sparse_ldlt.hpp"
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstring>
#include <limits>
#include <numeric>

namespace gro {

enum SolverKind : int {
  SOLVER_PCG_SCHUR = 0,   // PCGSchurSolver + BlockJacobiSchurPreconditioner
  SOLVER_PCG = 1,         // PCGSolver + BlockJacobiPreconditioner
  SOLVER_PCG_IDENTITY = 2,// PCGSolver + IdentityPreconditioner
  SOLVER_LDLT = 3,        // EigenLDLTSolver (full H)
  SOLVER_LDLT_SCHUR = 4,  // EigenSchurLDLTSolver
};

struct LMOptions {
  int solver = SOLVER_PCG_SCHUR;
  int iterations = 10;
  double initial_damping = 1e-4;
  int use_identity = 0;
  int pcg_max_iter = 10;
  double pcg_tol = 1.0;
  double pcg_rejection_ratio = 5.0;
  int early_stop = 0; // levenberg_marquardt2 (:255-418): leave after 3 accepted steps that each gain < 0.1 %
};

struct LMStats {
  int iterations_run = 0;
  int accepted = 0;
  int pcg_iterations = 0;   // total inner iterations
  double solve_seconds = 0; // time inside solver->solve
  double loop_seconds = 0;  // time of the LM for-loop
  double setup_seconds = 0; // structure + first linearize
};

// pairwise (tree) sum in T of f(0..m-1): the summation shape of thrust::reduce
template <typename T, typename F> static T tree_sum(size_t lo, size_t hi, F &&f) {
  if (hi - lo <= 32) { T s = 0; for (size_t i = lo; i < hi; ++i) s += f(i); return s; }
  const size_t mid = lo + (hi - lo) / 2;
  return tree_sum<T>(lo, mid, f) + tree_sum<T>(mid, hi, f);
}

template <typename T> struct BalOracle {
  size_t Nc = 0, Np = 0, No = 0, n = 0, pose_dim = 0;
  std::vector<T> cams, pts, obs;
  std::vector<int32_t> cam_idx, pt_idx;
  int loss_kind = LOSS_DEFAULT;
  T loss_delta = 0;
  bool scale_system = true;

  // linearisation state (scaled space)
  std::vector<T> res, Jc, Jp, chi2_vec, dchi2, scales, b;
  // H, block layout: Hcc[c] 9x9, Hcp[o] 9x3 (rows camera), Hll[l] 3x3, all column-major
  std::vector<T> Hcc, Hcp, Hll, prev_diag;
  // point CSR over observations (obs ids sorted by camera inside each point)
  std::vector<int64_t> pt_ptr, pt_obs;
  // Schur: upper blocks (i<=j) in column-major block order
  std::vector<int64_t> S_colptr, S_row; // block CSC
  std::vector<T> S, b_schur, Hll_inv;
  // backups
  std::vector<T> cams_backup, pts_backup;
  // preconditioner state
  std::vector<T> bj_blocks, bj_scalar, bj_inv; // PCGSolver block-Jacobi (cams then points)
  T damping = 0;
  bool damping_identity = false;
  // direct solvers
  SparseLDLT ldlt;
  std::vector<int64_t> csc_p, csc_i;
  std::vector<T> csc_x;
  bool ldlt_ready = false;
  int ldlt_kind = -1;
  int last_pcg_iters = 0;

  void init(size_t nc, size_t np, size_t no, const T *c, const T *p, const T *o, const int32_t *ci,
            const int32_t *pi) {
    Nc = nc; Np = np; No = no;
    cams.assign(c, c + 9 * nc); pts.assign(p, p + 3 * np); obs.assign(o, o + 2 * no);
    cam_idx.assign(ci, ci + no); pt_idx.assign(pi, pi + no);
    n = 9 * Nc + 3 * Np; pose_dim = 9 * Nc;
    res.assign(2 * No, 0); Jc.assign(18 * No, 0); Jp.assign(6 * No, 0);
    chi2_vec.assign(No, 0); dchi2.assign(No, 1);
    scales.assign(n, 1); b.assign(n, 0);
    build_structure();
  }

  // ---- structure ---------------------------------------------------------
  // graph.hpp:100-149: cameras (not eliminated) take block columns 0..Nc-1 by
  // global id, points (eliminated) Nc..Nc+Np-1; hessian.hpp:257-288 +
  // csc_utils.hpp:16-50: upper blocks sorted column-major.
  void build_structure() {
    pt_ptr.assign(Np + 1, 0);
    for (size_t o = 0; o < No; ++o) pt_ptr[pt_idx[o] + 1]++;
    for (size_t l = 0; l < Np; ++l) pt_ptr[l + 1] += pt_ptr[l];
    pt_obs.assign(No, 0);
    std::vector<int64_t> w(pt_ptr.begin(), pt_ptr.end() - 1);
    for (size_t o = 0; o < No; ++o) pt_obs[w[pt_idx[o]]++] = o;
    for (size_t l = 0; l < Np; ++l)
      std::stable_sort(pt_obs.begin() + pt_ptr[l], pt_obs.begin() + pt_ptr[l + 1],
                       [&](int64_t a, int64_t c) { return cam_idx[a] < cam_idx[c]; });
    Hcc.assign(81 * Nc, 0); Hcp.assign(27 * No, 0); Hll.assign(9 * Np, 0);
    prev_diag.assign(n, 0);
    // Schur pattern: Hpp pattern U {(i,j): i<=j co-observe a point} (schur.hpp:397-476)
    std::vector<int64_t> keys;
    for (size_t c = 0; c < Nc; ++c) keys.push_back((int64_t)c * (int64_t)Nc + (int64_t)c);
    for (size_t l = 0; l < Np; ++l)
      for (int64_t a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a)
        for (int64_t bb = a; bb < pt_ptr[l + 1]; ++bb) {
          int64_t i = cam_idx[pt_obs[a]], j = cam_idx[pt_obs[bb]];
          if (i > j) std::swap(i, j);
          keys.push_back(j * (int64_t)Nc + i); // column-major key
        }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    S_colptr.assign(Nc + 1, 0);
    S_row.resize(keys.size());
    for (size_t k = 0; k < keys.size(); ++k) {
      S_colptr[keys[k] / (int64_t)Nc + 1]++;
      S_row[k] = keys[k] % (int64_t)Nc;
    }
    for (size_t c = 0; c < Nc; ++c) S_colptr[c + 1] += S_colptr[c];
    S.assign(81 * keys.size(), 0);
    b_schur.assign(pose_dim, 0);
    Hll_inv.assign(9 * Np, 0);
    ldlt_ready = false;
  }
  int64_t s_block(int64_t i, int64_t j) const { // i<=j
    const auto beg = S_row.begin() + S_colptr[j], end = S_row.begin() + S_colptr[j + 1];
    return std::lower_bound(beg, end, i) - S_row.begin();
  }
  size_t nnzb_S() const { return S_row.size(); }

  // ---- linearisation -----------------------------------------------------
  // Per-factor precision matrices (factor.hpp:158-174 precision_matrices, read row-major, ops/linearize.hpp:283), per-factor
  // losses (factor.hpp:373-412: every factor carries its own loss object) and per-factor constraint data; empty = identity /
  // the one loss of set_loss / none.  model_kind selects the factor function (user_models.hpp): 0 the BAL camera of
  // examples/reprojection_error.cuh with its analytic Jacobian, > 0 a test model differentiated by dual numbers.
  std::vector<T> pmat, fdata;          // [No][4] each
  std::vector<int> loss_kinds;         // [No]
  std::vector<T> loss_deltas;          // [No]
  int model_kind = MODEL_BAL;
  // Jacobian STORAGE type S of Graph<T, S> (types.hpp:8-43, examples/bal.cu:338-345): 0 = T; 1 = bfloat16, 2 = float — every stored
  // entry is rounded to S when the Jacobian kernel writes it (ops/linearize.hpp:43-79: the block is evaluated in T, stored as S) and
  // again when scale_jacobians rewrites it (ops/linearize.hpp:142-180: J <- S(T(J) * scale)); everything downstream reads T(S)
  int jac_storage = 0;
  static T round_storage(T v, int mode) {
    if (mode == 0) return v;
    const float f = (float)v;
    if (mode == 2) return (T)f;
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) u = (u | 0x400000u) & 0xffff0000u; // NaN stays NaN
    else u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;                 // round to nearest even (cuda_bf16.h __float2bfloat16_rn)
    float g;
    std::memcpy(&g, &u, 4);
    return (T)g;
  }
  void round_jacobians() {
    if (!jac_storage) return;
    for (auto &v : Jc) v = round_storage(v, jac_storage);
    for (auto &v : Jp) v = round_storage(v, jac_storage);
  }
  // a^T (rho' P) b as ops/hessian.hpp's jtpj + the dchi2 factor evaluate it; P = I keeps the closed form used before
  T wdot(size_t o, T a0, T a1, T b0, T b1) const {
    if (pmat.empty()) return (a0 * b0 + a1 * b1) * dchi2[o];
    const T *P = &pmat[4 * o];
    return ((P[0] * b0 + P[1] * b1) * a0 + (P[2] * b0 + P[3] * b1) * a1) * dchi2[o];
  }
  void wvec(size_t o, T r0, T r1, T &x0, T &x1) const { // rho' P r  (ops/linearize.hpp:283-290)
    if (pmat.empty()) { x0 = dchi2[o] * r0; x1 = dchi2[o] * r1; return; }
    const T *P = &pmat[4 * o];
    x0 = dchi2[o] * (P[0] * r0 + P[1] * r1); x1 = dchi2[o] * (P[2] * r0 + P[3] * r1);
  }
  void compute_error() { // graph.hpp:212-217
    for (size_t o = 0; o < No; ++o) {
      if (model_kind == MODEL_BAL) bal_residual(&cams[9 * cam_idx[o]], &pts[3 * pt_idx[o]], &obs[2 * o], &res[2 * o]);
      else user_model_residual<T>(model_kind, &cams[9 * cam_idx[o]], &pts[3 * pt_idx[o]], &obs[2 * o], fdata.empty() ? nullptr : &fdata[4 * o], &res[2 * o]);
    }
  }
  T chi2() { // graph.hpp:219-225, factor.hpp:551-557, ops/chi2.hpp:34-44
    for (size_t o = 0; o < No; ++o) {
      T raw;
      if (pmat.empty()) raw = res[2 * o] * res[2 * o] + res[2 * o + 1] * res[2 * o + 1];
      else { // value += (sum_j P_ij r_j) r_i, ops/chi2.hpp:20-30
        const T *P = &pmat[4 * o];
        raw = (P[0] * res[2 * o] + P[1] * res[2 * o + 1]) * res[2 * o] + (P[2] * res[2 * o] + P[3] * res[2 * o + 1]) * res[2 * o + 1];
      }
      const int lk = loss_kinds.empty() ? loss_kind : loss_kinds[o];
      const T ld = loss_deltas.empty() ? loss_delta : loss_deltas[o];
      chi2_vec[o] = loss_value(lk, ld, raw);
      dchi2[o] = loss_derivative(lk, ld, raw);
    }
    return tree_sum<T>(0, No, [&](size_t o) { return chi2_vec[o]; }); // thrust::reduce, ops/chi2.hpp:61-64
  }
  // VertexDescriptor::set_fixed (vertex.hpp:262-264): the Jacobian kernels return before writing the block of a fixed vertex
  // (ops/linearize.hpp:24) and every consumer skips it (ops/hessian.hpp:95,135,190; the vertex has no Hessian column).
  // Restated with the column kept and the block zero: the vertex's gradient, Hessian block and step are exactly 0.
  std::vector<uint8_t> cam_fixed, pt_fixed;
  void set_fixed(const uint8_t *cf, const uint8_t *pf) {
    cam_fixed.assign(Nc, 0); pt_fixed.assign(Np, 0);
    if (cf) for (size_t c = 0; c < Nc; ++c) cam_fixed[c] = cf[c] ? 1 : 0;
    if (pf) for (size_t l = 0; l < Np; ++l) pt_fixed[l] = pf[l] ? 1 : 0;
  }
  void linearize() { // graph.hpp:236-290
    for (size_t o = 0; o < No; ++o) {
      if (model_kind == MODEL_BAL) bal_residual_jacobian(&cams[9 * cam_idx[o]], &pts[3 * pt_idx[o]], &obs[2 * o], &res[2 * o], &Jc[18 * o], &Jp[6 * o]);
      else user_model_residual_jacobian<T>(model_kind, &cams[9 * cam_idx[o]], &pts[3 * pt_idx[o]], &obs[2 * o], fdata.empty() ? nullptr : &fdata[4 * o], &res[2 * o], &Jc[18 * o], &Jp[6 * o]);
    }
    if (!cam_fixed.empty())
      for (size_t o = 0; o < No; ++o) {
        if (cam_fixed[cam_idx[o]]) std::fill(&Jc[18 * o], &Jc[18 * o] + 18, T(0));
        if (pt_fixed[pt_idx[o]]) std::fill(&Jp[6 * o], &Jp[6 * o] + 6, T(0));
      }
    round_jacobians(); // stored as S
    chi2();
    if (scale_system) {
      std::fill(scales.begin(), scales.end(), T(0));
      for (size_t o = 0; o < No; ++o) { // ops/hessian.hpp:419-474 (P = I)
        T *dc = &scales[9 * cam_idx[o]], *dp = &scales[pose_dim + 3 * pt_idx[o]];
        for (int c = 0; c < 9; ++c)
          dc[c] += wdot(o, Jc[18 * o + 2 * c], Jc[18 * o + 2 * c + 1], Jc[18 * o + 2 * c], Jc[18 * o + 2 * c + 1]);
        for (int c = 0; c < 3; ++c)
          dp[c] += wdot(o, Jp[6 * o + 2 * c], Jp[6 * o + 2 * c + 1], Jp[6 * o + 2 * c], Jp[6 * o + 2 * c + 1]);
      }
      for (size_t i = 0; i < n; ++i) { // graph.hpp:262-270
        const double denom = std::numeric_limits<double>::epsilon() + std::sqrt(static_cast<double>(scales[i]));
        scales[i] = static_cast<T>(1.0 / denom);
      }
      for (size_t o = 0; o < No; ++o) { // ops/linearize.hpp:142-180
        const T *sc = &scales[9 * cam_idx[o]], *sp = &scales[pose_dim + 3 * pt_idx[o]];
        for (int c = 0; c < 9; ++c) { Jc[18 * o + 2 * c] *= sc[c]; Jc[18 * o + 2 * c + 1] *= sc[c]; }
        for (int c = 0; c < 3; ++c) { Jp[6 * o + 2 * c] *= sp[c]; Jp[6 * o + 2 * c + 1] *= sp[c]; }
      }
      round_jacobians(); // rewritten as S
    } else {
      std::fill(scales.begin(), scales.end(), T(1));
    }
    std::fill(b.begin(), b.end(), T(0));
    for (size_t o = 0; o < No; ++o) { // ops/linearize.hpp:240-303
      T x0, x1;
      wvec(o, res[2 * o], res[2 * o + 1], x0, x1); // rho' P r
      T *bc = &b[9 * cam_idx[o]], *bp = &b[pose_dim + 3 * pt_idx[o]];
      for (int c = 0; c < 9; ++c) bc[c] -= Jc[18 * o + 2 * c] * x0 + Jc[18 * o + 2 * c + 1] * x1;
      for (int c = 0; c < 3; ++c) bp[c] -= Jp[6 * o + 2 * c] * x0 + Jp[6 * o + 2 * c + 1] * x1;
    }
  }

  // ---- Hessian -----------------------------------------------------------
  void hessian_update_values() { // hessian.hpp:290-307, ops/hessian.hpp:10-78
    std::fill(Hcc.begin(), Hcc.end(), T(0));
    std::fill(Hcp.begin(), Hcp.end(), T(0));
    std::fill(Hll.begin(), Hll.end(), T(0));
    for (size_t o = 0; o < No; ++o) {
      const T *jc = &Jc[18 * o], *jp = &Jp[6 * o];
      T *hcc = &Hcc[81 * cam_idx[o]], *hcp = &Hcp[27 * o], *hll = &Hll[9 * pt_idx[o]];
      for (int col = 0; col < 9; ++col)
        for (int row = 0; row < 9; ++row)
          hcc[row + 9 * col] += wdot(o, jc[2 * row], jc[2 * row + 1], jc[2 * col], jc[2 * col + 1]);
      for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 9; ++row)
          hcp[row + 9 * col] += wdot(o, jc[2 * row], jc[2 * row + 1], jp[2 * col], jp[2 * col + 1]);
      for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row)
          hll[row + 3 * col] += wdot(o, jp[2 * row], jp[2 * row + 1], jp[2 * col], jp[2 * col + 1]);
    }
    for (size_t c = 0; c < Nc; ++c) // backup_diagonal, hessian.hpp:102-134
      for (int i = 0; i < 9; ++i) prev_diag[9 * c + i] = Hcc[81 * c + 10 * i];
    for (size_t l = 0; l < Np; ++l)
      for (int i = 0; i < 3; ++i) prev_diag[pose_dim + 3 * l + i] = Hll[9 * l + 4 * i];
  }
  void apply_damping(T mu, bool use_identity) { // hessian.hpp:136-176
    damping = mu; damping_identity = use_identity;
    auto damp = [&](T d) {
      if (use_identity) return (T)((double)d + (double)mu);
      return (T)((double)d + mu * std::clamp((double)d, 1.0e-6, 1.0e32));
    };
    for (size_t c = 0; c < Nc; ++c)
      for (int i = 0; i < 9; ++i) Hcc[81 * c + 10 * i] = damp(prev_diag[9 * c + i]);
    for (size_t l = 0; l < Np; ++l)
      for (int i = 0; i < 3; ++i) Hll[9 * l + 4 * i] = damp(prev_diag[pose_dim + 3 * l + i]);
  }

  // Export H exactly in the reference's value layout (block-CSC upper,
  // column-major sorted, hessian.hpp:257-288).  Returns total value count.
  // Duplicate (camera,point) edges are summed into one block.
  size_t export_hessian(std::vector<T> &values, std::vector<int64_t> &colptr,
                        std::vector<int64_t> &rowidx, std::vector<int64_t> &offsets) const {
    colptr.assign(Nc + Np + 1, 0); rowidx.clear(); offsets.clear(); values.clear();
    for (size_t c = 0; c < Nc; ++c) {
      rowidx.push_back(c); offsets.push_back(values.size());
      values.insert(values.end(), &Hcc[81 * c], &Hcc[81 * c] + 81);
      colptr[c + 1] = rowidx.size();
    }
    for (size_t l = 0; l < Np; ++l) {
      int64_t last_cam = -1;
      for (int64_t a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) {
        const int64_t o = pt_obs[a];
        if (cam_idx[o] == last_cam) {
          for (int k = 0; k < 27; ++k) values[offsets.back() + k] += Hcp[27 * o + k];
          continue;
        }
        last_cam = cam_idx[o];
        rowidx.push_back(cam_idx[o]); offsets.push_back(values.size());
        values.insert(values.end(), &Hcp[27 * o], &Hcp[27 * o] + 27);
      }
      rowidx.push_back(Nc + l); offsets.push_back(values.size());
      values.insert(values.end(), &Hll[9 * l], &Hll[9 * l] + 9);
      colptr[Nc + l + 1] = rowidx.size();
    }
    return values.size();
  }

  // ---- Schur -------------------------------------------------------------
  void schur_update_values() { // schur.hpp:227-235
    std::fill(S.begin(), S.end(), T(0));
    for (size_t c = 0; c < Nc; ++c) // execute_Hpp_copy :587
      std::copy(&Hcc[81 * c], &Hcc[81 * c] + 81, &S[81 * s_block(c, c)]);
    for (size_t l = 0; l < Np; ++l) // execute_block_diagonal_inversion :1067
      small_inverse(3, &Hll[9 * l], &Hll_inv[9 * l]);
    // execute_schur_multiplication :649, ops/schur.hpp:155-188: S_ij -= L M R^T
    for (size_t l = 0; l < Np; ++l) {
      const T *M = &Hll_inv[9 * l];
      for (int64_t a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a)
        for (int64_t bb = a; bb < pt_ptr[l + 1]; ++bb) {
          int64_t oa = pt_obs[a], ob = pt_obs[bb];
          // destination is block (cam(oa), cam(ob)) with cam(oa) <= cam(ob) (sorted)
          const T *L = &Hcp[27 * oa], *R = &Hcp[27 * ob];
          T *dst = &S[81 * s_block(cam_idx[oa], cam_idx[ob])];
          const bool same = cam_idx[oa] == cam_idx[ob] && oa != ob;
          for (int col = 0; col < 9; ++col)
            for (int row = 0; row < 9; ++row) {
              T value = 0;
              for (int k = 0; k < 3; ++k) {
                T m_rt = 0;
                for (int j = 0; j < 3; ++j) m_rt += M[k + 3 * j] * R[col + 9 * j];
                value += L[row + 9 * k] * m_rt;
              }
              dst[row + 9 * col] -= value;
              if (same) dst[col + 9 * row] -= value; // duplicate edge: both orders land on the diagonal block
            }
        }
    }
    // execute_b_Schur_computation :901  b_S = b_p - Hpl Hll^-1 b_l
    std::copy(b.begin(), b.begin() + pose_dim, b_schur.begin());
    for (size_t l = 0; l < Np; ++l) {
      T v[3];
      for (int r = 0; r < 3; ++r) {
        v[r] = 0;
        for (int c = 0; c < 3; ++c) v[r] += Hll_inv[9 * l + r + 3 * c] * b[pose_dim + 3 * l + c];
      }
      for (int64_t a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) {
        const int64_t o = pt_obs[a];
        for (int r = 0; r < 9; ++r) {
          T sum = 0;
          for (int c = 0; c < 3; ++c) sum += Hcp[27 * o + r + 9 * c] * v[c];
          b_schur[9 * cam_idx[o] + r] -= sum;
        }
      }
    }
  }
  void landmark_update(const T *xp, T *xl) const { // schur.hpp:279-302
    for (size_t l = 0; l < Np; ++l) {
      T rhs[3] = {b[pose_dim + 3 * l], b[pose_dim + 3 * l + 1], b[pose_dim + 3 * l + 2]};
      for (int64_t a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) {
        const int64_t o = pt_obs[a];
        for (int c = 0; c < 3; ++c) {
          T sum = 0;
          for (int r = 0; r < 9; ++r) sum += Hcp[27 * o + r + 9 * c] * xp[9 * cam_idx[o] + r];
          rhs[c] -= sum;
        }
      }
      for (int r = 0; r < 3; ++r) {
        T sum = 0;
        for (int c = 0; c < 3; ++c) sum += Hll_inv[9 * l + r + 3 * c] * rhs[c];
        xl[3 * l + r] = sum;
      }
    }
  }
  void schur_matvec(const T *x, T *y) const { // schur.hpp:347-393
    std::fill(y, y + pose_dim, T(0));
    for (size_t j = 0; j < Nc; ++j)
      for (int64_t k = S_colptr[j]; k < S_colptr[j + 1]; ++k) {
        const int64_t i = S_row[k];
        const T *A = &S[81 * k];
        for (int r = 0; r < 9; ++r) {
          T sum = 0;
          for (int c = 0; c < 9; ++c) sum += A[r + 9 * c] * x[9 * j + c];
          y[9 * i + r] += sum;
        }
        if (i != (int64_t)j)
          for (int c = 0; c < 9; ++c) {
            T sum = 0;
            for (int r = 0; r < 9; ++r) sum += A[r + 9 * c] * x[9 * i + r];
            y[9 * j + c] += sum;
          }
      }
  }
  // scalar CSC (upper) of S, csc_utils.hpp:74-193
  void export_schur_csc(std::vector<int64_t> &p, std::vector<int64_t> &i, std::vector<T> &x) const {
    p.assign(pose_dim + 1, 0); i.clear(); x.clear();
    for (size_t j = 0; j < Nc; ++j)
      for (int cc = 0; cc < 9; ++cc) {
        const int64_t col = 9 * j + cc;
        for (int64_t k = S_colptr[j]; k < S_colptr[j + 1]; ++k)
          for (int r = 0; r < 9; ++r) {
            const int64_t row = 9 * S_row[k] + r;
            if (row <= col) { i.push_back(row); x.push_back(S[81 * k + r + 9 * cc]); }
          }
        p[col + 1] = i.size();
      }
  }
  void export_hessian_csc(std::vector<int64_t> &p, std::vector<int64_t> &i, std::vector<T> &x) const {
    std::vector<T> values; std::vector<int64_t> colptr, rowidx, offsets;
    export_hessian(values, colptr, rowidx, offsets);
    auto dim = [&](int64_t blk) { return blk < (int64_t)Nc ? 9 : 3; };
    auto off = [&](int64_t blk) { return blk < (int64_t)Nc ? 9 * blk : (int64_t)pose_dim + 3 * (blk - (int64_t)Nc); };
    p.assign(n + 1, 0); i.clear(); x.clear();
    for (int64_t bc = 0; bc < (int64_t)(Nc + Np); ++bc)
      for (int cc = 0; cc < dim(bc); ++cc) {
        const int64_t col = off(bc) + cc;
        for (int64_t k = colptr[bc]; k < colptr[bc + 1]; ++k) {
          const int64_t nr = dim(rowidx[k]);
          for (int r = 0; r < nr; ++r) {
            const int64_t row = off(rowidx[k]) + r;
            if (row <= col) { i.push_back(row); x.push_back(values[offsets[k] + r + nr * cc]); }
          }
        }
        p[col + 1] = i.size();
      }
  }

  // ---- solvers -----------------------------------------------------------
  static T dot(size_t m, const T *a, const T *c) { return tree_sum<T>(0, m, [&](size_t i) { return a[i] * c[i]; }); } // thrust::inner_product

  // PCGSchurSolver::solve, solver/pcg_schur.hpp:79-168
  bool solve_pcg_schur(T *x, int max_iter, T tol, T rejection_ratio) {
    schur_update_values();
    std::vector<T> Minv(81 * Nc); // block_jacobi_schur.hpp:114-150
    for (size_t c = 0; c < Nc; ++c) small_inverse(9, &S[81 * s_block(c, c)], &Minv[81 * c]);
    auto precond = [&](T *z, const T *r) { // :153-178
      for (size_t c = 0; c < Nc; ++c)
        for (int row = 0; row < 9; ++row) {
          T sum = 0;
          for (int k = 0; k < 9; ++k) sum += Minv[81 * c + row + 9 * k] * r[9 * c + k];
          z[9 * c + row] = sum;
        }
    };
    std::fill(x, x + n, T(0));
    std::vector<T> r(b_schur), p(pose_dim), z(pose_dim), Ap(pose_dim), xb(pose_dim);
    precond(z.data(), r.data());
    p = z;
    T rz = dot(pose_dim, r.data(), z.data());
    T rz_0 = std::numeric_limits<T>::infinity();
    last_pcg_iters = 0;
    for (int k = 0; k < max_iter; ++k) {
      if (rz == T(0)) break;
      schur_matvec(p.data(), Ap.data());
      const T denom = dot(pose_dim, p.data(), Ap.data());
      if (denom == T(0) || std::isnan(denom)) break;
      last_pcg_iters++;
      const T alpha = rz / denom;
      std::copy(x, x + pose_dim, xb.begin());
      for (size_t i = 0; i < pose_dim; ++i) x[i] = alpha * p[i] + x[i];
      for (size_t i = 0; i < pose_dim; ++i) r[i] = -alpha * Ap[i] + r[i];
      precond(z.data(), r.data());
      const T rz_new = dot(pose_dim, r.data(), z.data());
      if (std::abs(rz_new) > rejection_ratio * rz_0 || std::isnan(rz_new)) {
        std::copy(xb.begin(), xb.end(), x);
        break;
      }
      rz_0 = std::min(rz_0, std::abs(rz_new));
      const T beta = rz_new / rz;
      rz = rz_new;
      for (size_t i = 0; i < pose_dim; ++i) p[i] = beta * p[i] + z[i];
      if (std::abs(rz_new) < tol) break;
    }
    landmark_update(x, x + pose_dim);
    return true;
  }

  // BlockJacobiPreconditioner::update_values, block_jacobi.hpp:79-118
  void block_jacobi_update_values() {
    bj_blocks.assign(81 * Nc + 9 * Np, 0);
    for (size_t o = 0; o < No; ++o) {
      const T *jc = &Jc[18 * o], *jp = &Jp[6 * o];
      T *hc = &bj_blocks[81 * cam_idx[o]], *hl = &bj_blocks[81 * Nc + 9 * pt_idx[o]];
      for (int col = 0; col < 9; ++col)
        for (int row = 0; row < 9; ++row)
          hc[row + 9 * col] += wdot(o, jc[2 * row], jc[2 * row + 1], jc[2 * col], jc[2 * col + 1]);
      for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row)
          hl[row + 3 * col] += wdot(o, jp[2 * row], jp[2 * row + 1], jp[2 * col], jp[2 * col + 1]);
    }
    bj_scalar.assign(n, 0);
    for (size_t c = 0; c < Nc; ++c) for (int i = 0; i < 9; ++i) bj_scalar[9 * c + i] = bj_blocks[81 * c + 10 * i];
    for (size_t l = 0; l < Np; ++l) for (int i = 0; i < 3; ++i) bj_scalar[pose_dim + 3 * l + i] = bj_blocks[81 * Nc + 9 * l + 4 * i];
  }
  // BlockJacobiPreconditioner::set_damping_factor, block_jacobi.hpp:120-172
  void block_jacobi_set_damping(T mu, bool use_identity) {
    bj_inv.assign(81 * Nc + 9 * Np, 0);
    auto damp = [&](T d) {
      if (use_identity) return (T)((double)d + (double)mu);
      return (T)((double)d + (double)mu * std::clamp((double)d, 1.0e-6, 1.0e32));
    };
    for (size_t c = 0; c < Nc; ++c) {
      for (int i = 0; i < 9; ++i) bj_blocks[81 * c + 10 * i] = damp(bj_scalar[9 * c + i]);
      small_inverse(9, &bj_blocks[81 * c], &bj_inv[81 * c]);
    }
    for (size_t l = 0; l < Np; ++l) {
      for (int i = 0; i < 3; ++i) bj_blocks[81 * Nc + 9 * l + 4 * i] = damp(bj_scalar[pose_dim + 3 * l + i]);
      small_inverse(3, &bj_blocks[81 * Nc + 9 * l], &bj_inv[81 * Nc + 9 * l]);
    }
  }
  void block_jacobi_apply(T *z, const T *r) const { // block_jacobi.hpp:174-186
    for (size_t c = 0; c < Nc; ++c)
      for (int row = 0; row < 9; ++row) {
        T s = 0;
        for (int k = 0; k < 9; ++k) s += bj_inv[81 * c + row + 9 * k] * r[9 * c + k];
        z[9 * c + row] = s;
      }
    for (size_t l = 0; l < Np; ++l)
      for (int row = 0; row < 3; ++row) {
        T s = 0;
        for (int k = 0; k < 3; ++k) s += bj_inv[81 * Nc + 9 * l + row + 3 * k] * r[pose_dim + 3 * l + k];
        z[pose_dim + 3 * l + row] = s;
      }
  }
  // (J^T rho' J) p  via compute_Jv then compute_Jtv (solver/pcg.hpp:141-163)
  void JtJ_matvec(const T *p, T *v2, std::vector<T> &v1) const {
    v1.assign(2 * No, 0);
    for (size_t o = 0; o < No; ++o) {
      const T *pc = &p[9 * cam_idx[o]], *pp = &p[pose_dim + 3 * pt_idx[o]];
      for (int e = 0; e < 2; ++e) {
        T s = 0;
        for (int i = 0; i < 9; ++i) s += Jc[18 * o + e + 2 * i] * pc[i];
        v1[2 * o + e] += s;
      }
      for (int e = 0; e < 2; ++e) {
        T s = 0;
        for (int i = 0; i < 3; ++i) s += Jp[6 * o + e + 2 * i] * pp[i];
        v1[2 * o + e] += s;
      }
    }
    std::fill(v2, v2 + n, T(0));
    for (size_t o = 0; o < No; ++o) {
      T *yc = &v2[9 * cam_idx[o]], *yp = &v2[pose_dim + 3 * pt_idx[o]];
      for (int c = 0; c < 9; ++c)
        yc[c] += wdot(o, Jc[18 * o + 2 * c], Jc[18 * o + 2 * c + 1], v1[2 * o], v1[2 * o + 1]);
      for (int c = 0; c < 3; ++c)
        yp[c] += wdot(o, Jp[6 * o + 2 * c], Jp[6 * o + 2 * c + 1], v1[2 * o], v1[2 * o + 1]);
    }
  }
  // DOCUMENTED VARIANT, not in the reference: the same preconditioned CG with the Chronopoulos-Gear single-reduction
  // recurrence (one global reduction point per iteration instead of two), which the GPU engine runs on landmark shards
  // (one all-reduce per inner iteration).  Same iterates as solve_pcg in exact arithmetic:
  //   u_k = M^-1 (r_k / |r_k|), gamma_k = r_k.u_k, w_k = A u_k, delta_k = u_k.w_k,
  //   beta_k = gamma_k / gamma_{k-1} (0 for k = 0), alpha_k = gamma_k / (delta_k - beta_k gamma_k / alpha_{k-1}),
  //   p_k = u_k + beta_k p_{k-1}, s_k = w_k + beta_k s_{k-1} (= A p_k), x += alpha_k p_k, r -= alpha_k s_k,
  // with the exits of solver/pcg.hpp:166-229 (rejection ratio, tolerance, rz == 0) on gamma_{k+1} unchanged.
  bool pcg_single_reduction = false;
  bool solve_pcg_cg(T *x, int max_iter, T tol, T rejection_ratio, bool identity_precond) {
    std::vector<T> v1, w(n), r(b), p(n, 0), sv(n, 0), u(n), diag(n, 0), xb(n), y(n);
    std::fill(x, x + n, T(0));
    for (size_t o = 0; o < No; ++o) {
      T *dc = &diag[9 * cam_idx[o]], *dp = &diag[pose_dim + 3 * pt_idx[o]];
      for (int c = 0; c < 9; ++c)
        dc[c] += wdot(o, Jc[18 * o + 2 * c], Jc[18 * o + 2 * c + 1], Jc[18 * o + 2 * c], Jc[18 * o + 2 * c + 1]);
      for (int c = 0; c < 3; ++c)
        dp[c] += wdot(o, Jp[6 * o + 2 * c], Jp[6 * o + 2 * c + 1], Jp[6 * o + 2 * c], Jp[6 * o + 2 * c + 1]);
    }
    for (size_t i = 0; i < n; ++i) diag[i] = std::clamp(diag[i], T(1.0e-6), T(1.0e32));
    auto precond = [&](T *zz, const T *yy) {
      if (identity_precond) std::copy(yy, yy + n, zz);
      else block_jacobi_apply(zz, yy);
    };
    auto normalised_precond = [&]() {
      const T scale = T(1.0 / std::sqrt(dot(n, r.data(), r.data())));
      for (size_t i = 0; i < n; ++i) y[i] = scale * r[i];
      precond(u.data(), y.data());
    };
    normalised_precond();
    T gamma = dot(n, r.data(), u.data()), gamma_prev = 0, alpha_prev = 0;
    T rz_0 = std::numeric_limits<T>::infinity();
    last_pcg_iters = 0;
    for (int k = 0; k < max_iter; ++k) {
      if (gamma == 0) break;
      JtJ_matvec(u.data(), w.data(), v1);
      for (size_t i = 0; i < n; ++i) w[i] += damping_identity ? damping * u[i] : damping * diag[i] * u[i];
      last_pcg_iters++;
      const T delta = dot(n, u.data(), w.data());
      const T beta = k == 0 ? T(0) : gamma / gamma_prev;
      const T alpha = k == 0 ? gamma / delta : gamma / (delta - beta * gamma / alpha_prev);
      for (size_t i = 0; i < n; ++i) { p[i] = u[i] + beta * p[i]; sv[i] = w[i] + beta * sv[i]; }
      xb.assign(x, x + n);
      for (size_t i = 0; i < n; ++i) x[i] = alpha * p[i] + x[i];
      for (size_t i = 0; i < n; ++i) r[i] = -alpha * sv[i] + r[i];
      normalised_precond();
      const T gamma_new = dot(n, r.data(), u.data());
      if (std::abs(gamma_new) > rejection_ratio * rz_0 || std::isnan(gamma_new)) {
        std::copy(xb.begin(), xb.end(), x);
        break;
      }
      rz_0 = std::min(rz_0, std::abs(gamma_new));
      gamma_prev = gamma; alpha_prev = alpha; gamma = gamma_new;
      if (std::abs(gamma_new) < tol) break;
    }
    return true;
  }
  // PCGSolver::solve, solver/pcg.hpp:61-232.  identity_precond: IdentityPreconditioner
  bool solve_pcg(T *x, int max_iter, T tol, T rejection_ratio, bool identity_precond) {
    if (pcg_single_reduction) return solve_pcg_cg(x, max_iter, tol, rejection_ratio, identity_precond);
    std::vector<T> v1, v2(n), r(b), p(n), z(n), diag(n, 0), xb(n), y(n);
    std::fill(x, x + n, T(0));
    for (size_t o = 0; o < No; ++o) { // pcg.hpp:93-98 diag(J^T rho' J) from the (scaled) stored J
      T *dc = &diag[9 * cam_idx[o]], *dp = &diag[pose_dim + 3 * pt_idx[o]];
      for (int c = 0; c < 9; ++c)
        dc[c] += wdot(o, Jc[18 * o + 2 * c], Jc[18 * o + 2 * c + 1], Jc[18 * o + 2 * c], Jc[18 * o + 2 * c + 1]);
      for (int c = 0; c < 3; ++c)
        dp[c] += wdot(o, Jp[6 * o + 2 * c], Jp[6 * o + 2 * c + 1], Jp[6 * o + 2 * c], Jp[6 * o + 2 * c + 1]);
    }
    for (size_t i = 0; i < n; ++i) diag[i] = std::clamp(diag[i], T(1.0e-6), T(1.0e32));
    auto precond = [&](T *zz, const T *yy) {
      if (identity_precond) std::copy(yy, yy + n, zz);
      else block_jacobi_apply(zz, yy);
    };
    T rnorm = std::sqrt(dot(n, r.data(), r.data()));
    T scale = T(1.0 / rnorm);
    for (size_t i = 0; i < n; ++i) y[i] = scale * r[i];
    precond(z.data(), y.data());
    p = z;
    T rz = dot(n, r.data(), z.data());
    T rz_0 = std::numeric_limits<T>::infinity();
    last_pcg_iters = 0;
    for (int k = 0; k < max_iter; ++k) {
      if (rz == 0) break;
      JtJ_matvec(p.data(), v2.data(), v1);
      for (size_t i = 0; i < n; ++i) // ops/vector.hpp:25-41
        v2[i] += damping_identity ? damping * p[i] : damping * diag[i] * p[i];
      last_pcg_iters++;
      const T alpha = rz / dot(n, p.data(), v2.data());
      xb.assign(x, x + n);
      for (size_t i = 0; i < n; ++i) x[i] = alpha * p[i] + x[i];
      for (size_t i = 0; i < n; ++i) r[i] = -alpha * v2[i] + r[i];
      rnorm = std::sqrt(dot(n, r.data(), r.data()));
      scale = T(1.0 / rnorm);
      for (size_t i = 0; i < n; ++i) y[i] = scale * r[i];
      precond(z.data(), y.data());
      const T rz_new = dot(n, r.data(), z.data());
      if (std::abs(rz_new) > rejection_ratio * rz_0 || std::isnan(rz_new)) {
        std::copy(xb.begin(), xb.end(), x);
        break;
      }
      rz_0 = std::min(rz_0, std::abs(rz_new));
      const T beta = rz_new / rz;
      rz = rz_new;
      for (size_t i = 0; i < n; ++i) p[i] = beta * p[i] + z[i];
      if (std::abs(rz_new) < tol) break;
    }
    return true;
  }

  // reverse Cuthill-McKee on the camera co-observation graph (S pattern)
  std::vector<int64_t> camera_rcm() const {
    std::vector<std::vector<int64_t>> adj(Nc);
    for (size_t j = 0; j < Nc; ++j)
      for (int64_t k = S_colptr[j]; k < S_colptr[j + 1]; ++k)
        if (S_row[k] != (int64_t)j) { adj[j].push_back(S_row[k]); adj[S_row[k]].push_back(j); }
    std::vector<int64_t> order; order.reserve(Nc);
    std::vector<char> seen(Nc, 0);
    std::vector<int64_t> byDeg(Nc);
    std::iota(byDeg.begin(), byDeg.end(), 0);
    std::stable_sort(byDeg.begin(), byDeg.end(), [&](int64_t a, int64_t c) { return adj[a].size() < adj[c].size(); });
    for (int64_t start : byDeg) {
      if (seen[start]) continue;
      size_t head = order.size();
      order.push_back(start); seen[start] = 1;
      while (head < order.size()) {
        const int64_t v = order[head++];
        std::vector<int64_t> nb;
        for (int64_t u : adj[v]) if (!seen[u]) { seen[u] = 1; nb.push_back(u); }
        std::stable_sort(nb.begin(), nb.end(), [&](int64_t a, int64_t c) { return adj[a].size() < adj[c].size(); });
        order.insert(order.end(), nb.begin(), nb.end());
      }
    }
    std::reverse(order.begin(), order.end());
    return order;
  }
  // EigenSchurLDLTSolver::solve, solver/eigen_schur.hpp:71-108
  bool solve_ldlt_schur(T *x) {
    schur_update_values();
    export_schur_csc(csc_p, csc_i, csc_x);
    if (!ldlt_ready || ldlt_kind != SOLVER_LDLT_SCHUR) {
      std::vector<int64_t> perm;
      for (int64_t c : camera_rcm()) for (int k = 0; k < 9; ++k) perm.push_back(9 * c + k);
      ldlt.analyze(pose_dim, csc_p.data(), csc_i.data(), perm);
      ldlt_ready = true; ldlt_kind = SOLVER_LDLT_SCHUR;
    }
    if (!ldlt.factorize(csc_x.data())) return false;
    std::fill(x, x + n, T(0));
    if (!ldlt.solve(b_schur.data(), x)) return false;
    landmark_update(x, x + pose_dim);
    return true;
  }
  // EigenLDLTSolver::solve, solver/eigen.hpp:71-98
  bool solve_ldlt_full(T *x) {
    export_hessian_csc(csc_p, csc_i, csc_x);
    if (!ldlt_ready || ldlt_kind != SOLVER_LDLT) {
      std::vector<int64_t> perm;
      for (size_t l = 0; l < Np; ++l) for (int k = 0; k < 3; ++k) perm.push_back(pose_dim + 3 * l + k);
      for (int64_t c : camera_rcm()) for (int k = 0; k < 9; ++k) perm.push_back(9 * c + k);
      ldlt.analyze(n, csc_p.data(), csc_i.data(), perm);
      ldlt_ready = true; ldlt_kind = SOLVER_LDLT;
    }
    if (!ldlt.factorize(csc_x.data())) return false;
    return ldlt.solve(b.data(), x);
  }

  // ---- update / backup ---------------------------------------------------
  void backup() { cams_backup = cams; pts_backup = pts; }        // graph.hpp:302-309
  void revert() { cams = cams_backup; pts = pts_backup; }        // graph.hpp:311-318
  void apply_update(const T *dx) {                                // ops/update.hpp:11-31
    for (size_t i = 0; i < pose_dim; ++i) cams[i] += dx[i] * scales[i];
    for (size_t i = 0; i < 3 * Np; ++i) pts[i] += dx[pose_dim + i] * scales[pose_dim + i];
  }

  // ---- Solver interface glue (solver/solver.hpp:12-25) ---------------------
  void solver_update_values(int kind) {
    if (kind == SOLVER_PCG) block_jacobi_update_values();
    else if (kind != SOLVER_PCG_IDENTITY) hessian_update_values();
  }
  void solver_set_damping(int kind, T mu, bool use_identity) {
    damping = mu; damping_identity = use_identity;
    if (kind == SOLVER_PCG) block_jacobi_set_damping(mu, use_identity);
    else if (kind != SOLVER_PCG_IDENTITY) apply_damping(mu, use_identity);
  }
  bool solver_solve(const LMOptions &opt, T *x) {
    switch (opt.solver) {
    case SOLVER_PCG_SCHUR: return solve_pcg_schur(x, opt.pcg_max_iter, (T)opt.pcg_tol, (T)opt.pcg_rejection_ratio);
    case SOLVER_PCG: return solve_pcg(x, opt.pcg_max_iter, (T)opt.pcg_tol, (T)opt.pcg_rejection_ratio, false);
    case SOLVER_PCG_IDENTITY: return solve_pcg(x, opt.pcg_max_iter, (T)opt.pcg_tol, (T)opt.pcg_rejection_ratio, true);
    case SOLVER_LDLT: last_pcg_iters = 0; return solve_ldlt_full(x);
    case SOLVER_LDLT_SCHUR: last_pcg_iters = 0; return solve_ldlt_schur(x);
    }
    return false;
  }

  // optimizer/levenberg_marquardt.hpp:110-242 (and levenberg_marquardt2, :255-418, with opt.early_stop:
  // the same iteration plus the ORB-SLAM-style termination of :404-414).  chi2_trace gets the "Current
  // Chi2" column (one entry per LM iteration), lambda_trace the damping after it.
  bool levenberg_marquardt(const LMOptions &opt, std::vector<double> &chi2_trace,
                           std::vector<double> &lambda_trace, LMStats &st) {
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    T mu = static_cast<T>(opt.initial_damping);
    T nu = 2;
    ldlt_ready = false;
    linearize();
    solver_update_values(opt.solver);
    T chi2v = chi2();
    std::vector<T> dx(n, 0);
    bool run = true;
    int num_bad = 0;
    st = LMStats();
    st.setup_seconds = std::chrono::duration<double>(clk::now() - t0).count();
    chi2_trace.clear(); lambda_trace.clear();
    chi2_trace.push_back((double)chi2v); lambda_trace.push_back((double)mu);
    auto tl = clk::now();
    for (int i = 0; i < opt.iterations && run; ++i) {
      solver_set_damping(opt.solver, mu, opt.use_identity != 0);
      auto ts = clk::now();
      const bool solve_ok = solver_solve(opt, dx.data());
      st.solve_seconds += std::chrono::duration<double>(clk::now() - ts).count();
      st.pcg_iterations += last_pcg_iters;
      backup();
      apply_update(dx.data());
      compute_error();
      T new_chi2 = chi2();
      if (!solve_ok) new_chi2 = std::numeric_limits<T>::max();
      // compute_rho :20-47
      T num = chi2v - new_chi2, denom = 1.0;
      if (solve_ok) {
        denom = tree_sum<T>(0, n, [&](size_t k) { return dx[k] * (mu * dx[k] + b[k]); });
        denom += T(1.0e-3);
      }
      const T rho = num / denom;
      const T initial_chi2 = chi2v;
      bool step_accepted = false;
      if (solve_ok && std::isfinite(new_chi2) && rho > 0) {
        step_accepted = true;
        double alpha = 1.0 - std::pow(2.0 * rho - 1.0, 3);
        alpha = std::max(std::min(alpha, 2.0 / 3.0), 1.0 / 3.0);
        mu *= static_cast<T>(alpha);
        nu = 2;
        linearize();
        solver_update_values(opt.solver);
        st.accepted++;
      } else {
        revert();
        compute_error();
        chi2();
        mu *= nu;
        nu *= 2;
        new_chi2 = chi2v;
      }
      chi2v = new_chi2;
      st.iterations_run++;
      chi2_trace.push_back((double)chi2v); lambda_trace.push_back((double)mu);
      if (!std::isfinite(mu)) run = false;
      if (rho == 0) break;
      if (opt.early_stop && step_accepted) { // :404-414
        if (((initial_chi2 - chi2v) * 1.0e3) < initial_chi2) num_bad++;
        else num_bad = 0;
        if (num_bad >= 3) break;
      }
    }
    st.loop_seconds = std::chrono::duration<double>(clk::now() - tl).count();
    return run;
  }
};

} // namespace gro
