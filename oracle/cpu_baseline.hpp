// ORACLE / CPU baseline — test and measurement infrastructure only (see bal_model.hpp header).
//
// The timed CPU comparator of SURVEY §8(d): the oracle's Levenberg-Marquardt (bal_pipeline.hpp, same
// algorithm, same options, same stopping rule) with
//   (i)   the full-H direct solve  (EigenLDLTSolver,      solver/eigen.hpp:49-98,  src/eigen_solver.cpp:8-30),
//   (ii)  the Schur direct solve   (EigenSchurLDLTSolver, solver/eigen_schur.hpp:71-108),
//   (iii) either with linearisation / Hessian assembly / Schur reduction OpenMP-parallel over ALL host
//         cores (the reference runs those stages on the GPU and ONLY the LDL^T on one CPU thread, so the
//         LDL^T seconds are reported on their own: they are what the reference's "eigen_solver CPU path"
//         costs on this host whatever the GPU does),
//   (iv)  the matrix-free block-Jacobi PCG (PCGSolver) with every stage parallel: the like-for-like
//         comparator of the GPU bench line.
// Ordering of the simplicial LDL^T: Eigen::SimplicialLDLT orders with AMD.  Here: points first (they have
// the lowest degrees of a bundle-adjustment Hessian and no minimum-degree heuristic does anything else),
// then the cameras by exact minimum degree on the camera co-observation graph (ordering = 1, the AMD role)
// or by reverse Cuthill-McKee (ordering = 0, what the oracle itself uses), or — ordering = 2, the reference's own choice —
// AMD of the scalar pattern of the matrix that is factorised (amd.hpp: Eigen::SimplicialLDLT's default AMDOrdering).
// The parallel stages sum per vertex in observation order, like the sequential oracle; only the Schur
// products are grouped by destination block column instead of by point (rounding-level differences).
#pragma once
#include "bal_pipeline.hpp"
#include "amd.hpp"
#include <omp.h>

namespace gro {

struct BaselineTimes {
  double linearize = 0, hessian = 0, schur = 0, export_csc = 0, ldlt_analyze = 0, ldlt_factor = 0, ldlt_solve = 0,
         backsub = 0, pcg = 0, update_chi2 = 0, loop = 0, setup = 0;
  int64_t ldlt_nnz = 0;
};

template <typename T> struct CpuBaseline : BalOracle<T> {
  using B = BalOracle<T>;
  using B::Nc; using B::Np; using B::No; using B::n; using B::pose_dim; using B::cams; using B::pts; using B::obs;
  using B::cam_idx; using B::pt_idx; using B::res; using B::Jc; using B::Jp; using B::chi2_vec; using B::dchi2;
  using B::scales; using B::b; using B::Hcc; using B::Hcp; using B::Hll; using B::prev_diag; using B::pt_ptr;
  using B::pt_obs; using B::S; using B::S_colptr; using B::S_row; using B::b_schur; using B::Hll_inv;
  using B::bj_blocks; using B::bj_scalar; using B::bj_inv; using B::damping; using B::damping_identity;
  using B::ldlt; using B::csc_p; using B::csc_i; using B::csc_x; using B::loss_kind; using B::loss_delta;

  int threads = 1, ordering = 1;
  std::vector<int64_t> cam_ptr, cam_obs; // camera CSR over observations, ascending observation id
  BaselineTimes tm;
  bool analyzed = false;
  int analyzed_kind = -1;

  void prepare(int threads_, int ordering_) {
    threads = std::max(1, threads_);
    ordering = ordering_;
    cam_ptr.assign(Nc + 1, 0);
    for (size_t o = 0; o < No; ++o) cam_ptr[cam_idx[o] + 1]++;
    for (size_t c = 0; c < Nc; ++c) cam_ptr[c + 1] += cam_ptr[c];
    cam_obs.assign(No, 0);
    std::vector<int64_t> w(cam_ptr.begin(), cam_ptr.end() - 1);
    for (size_t o = 0; o < No; ++o) cam_obs[w[cam_idx[o]]++] = (int64_t)o;
    analyzed = false;
  }

  // exact minimum degree on the camera co-observation graph (bitset rows; ties -> lowest id)
  std::vector<int64_t> camera_min_degree() const {
    const size_t W = (Nc + 63) / 64;
    std::vector<uint64_t> adj(Nc * W, 0);
    auto set = [&](size_t i, size_t j) { adj[i * W + j / 64] |= 1ull << (j % 64); };
    for (size_t j = 0; j < Nc; ++j)
      for (int64_t k = S_colptr[j]; k < S_colptr[j + 1]; ++k)
        if (S_row[k] != (int64_t)j) { set(j, S_row[k]); set(S_row[k], j); }
    std::vector<int64_t> deg(Nc), order;
    std::vector<char> alive(Nc, 1);
    auto popc = [&](size_t i) { int64_t d = 0; for (size_t w = 0; w < W; ++w) d += __builtin_popcountll(adj[i * W + w]); return d; };
    for (size_t i = 0; i < Nc; ++i) deg[i] = popc(i);
    order.reserve(Nc);
    for (size_t step = 0; step < Nc; ++step) {
      int64_t v = -1;
      for (size_t i = 0; i < Nc; ++i) if (alive[i] && (v < 0 || deg[i] < deg[v])) v = (int64_t)i;
      order.push_back(v);
      alive[v] = 0;
      const uint64_t *rv = &adj[(size_t)v * W];
      for (size_t w = 0; w < W; ++w) {
        uint64_t bits = rv[w];
        while (bits) {
          const size_t u = w * 64 + __builtin_ctzll(bits);
          bits &= bits - 1;
          uint64_t *ru = &adj[u * W];
          for (size_t q = 0; q < W; ++q) ru[q] |= rv[q];
          ru[(size_t)v / 64] &= ~(1ull << ((size_t)v % 64));
          ru[u / 64] &= ~(1ull << (u % 64));
          deg[u] = popc(u);
        }
      }
    }
    return order;
  }
  std::vector<int64_t> camera_order() const { return ordering == 1 ? camera_min_degree() : B::camera_rcm(); }

  // ---- parallel stages (same arithmetic as bal_pipeline.hpp, per-vertex sums in observation order) ----
  T chi2_mt() {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t o = 0; o < (int64_t)No; ++o) {
      const T raw = res[2 * o] * res[2 * o] + res[2 * o + 1] * res[2 * o + 1];
      chi2_vec[o] = loss_value(loss_kind, loss_delta, raw);
      dchi2[o] = loss_derivative(loss_kind, loss_delta, raw);
    }
    return tree_sum<T>(0, No, [&](size_t o) { return chi2_vec[o]; });
  }
  void compute_error_mt() {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t o = 0; o < (int64_t)No; ++o) bal_residual(&cams[9 * cam_idx[o]], &pts[3 * pt_idx[o]], &obs[2 * o], &res[2 * o]);
  }
  void linearize_mt() { // graph.hpp:236-290
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t o = 0; o < (int64_t)No; ++o)
      bal_residual_jacobian(&cams[9 * cam_idx[o]], &pts[3 * pt_idx[o]], &obs[2 * o], &res[2 * o], &Jc[18 * o], &Jp[6 * o]);
    chi2_mt();
    if (this->scale_system) {
#pragma omp parallel num_threads(threads)
      {
#pragma omp for schedule(dynamic, 16) nowait
        for (int64_t c = 0; c < (int64_t)Nc; ++c) {
          T d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int64_t q = cam_ptr[c]; q < cam_ptr[c + 1]; ++q) {
            const int64_t o = cam_obs[q];
            for (int k = 0; k < 9; ++k) d[k] += (Jc[18 * o + 2 * k] * Jc[18 * o + 2 * k] + Jc[18 * o + 2 * k + 1] * Jc[18 * o + 2 * k + 1]) * dchi2[o];
          }
          for (int k = 0; k < 9; ++k) scales[9 * c + k] = (T)(1.0 / (std::numeric_limits<double>::epsilon() + std::sqrt((double)d[k])));
        }
#pragma omp for schedule(static)
        for (int64_t l = 0; l < (int64_t)Np; ++l) {
          T d[3] = {0, 0, 0};
          for (int64_t q = pt_ptr[l]; q < pt_ptr[l + 1]; ++q) {
            const int64_t o = pt_obs[q];
            for (int k = 0; k < 3; ++k) d[k] += (Jp[6 * o + 2 * k] * Jp[6 * o + 2 * k] + Jp[6 * o + 2 * k + 1] * Jp[6 * o + 2 * k + 1]) * dchi2[o];
          }
          for (int k = 0; k < 3; ++k) scales[pose_dim + 3 * l + k] = (T)(1.0 / (std::numeric_limits<double>::epsilon() + std::sqrt((double)d[k])));
        }
      }
#pragma omp parallel for num_threads(threads) schedule(static)
      for (int64_t o = 0; o < (int64_t)No; ++o) {
        const T *sc = &scales[9 * cam_idx[o]], *sp = &scales[pose_dim + 3 * pt_idx[o]];
        for (int c = 0; c < 9; ++c) { Jc[18 * o + 2 * c] *= sc[c]; Jc[18 * o + 2 * c + 1] *= sc[c]; }
        for (int c = 0; c < 3; ++c) { Jp[6 * o + 2 * c] *= sp[c]; Jp[6 * o + 2 * c + 1] *= sp[c]; }
      }
    } else std::fill(scales.begin(), scales.end(), T(1));
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(dynamic, 16) nowait
      for (int64_t c = 0; c < (int64_t)Nc; ++c) {
        T v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t q = cam_ptr[c]; q < cam_ptr[c + 1]; ++q) {
          const int64_t o = cam_obs[q];
          const T x0 = dchi2[o] * res[2 * o], x1 = dchi2[o] * res[2 * o + 1];
          for (int k = 0; k < 9; ++k) v[k] -= Jc[18 * o + 2 * k] * x0 + Jc[18 * o + 2 * k + 1] * x1;
        }
        for (int k = 0; k < 9; ++k) b[9 * c + k] = v[k];
      }
#pragma omp for schedule(static)
      for (int64_t l = 0; l < (int64_t)Np; ++l) {
        T v[3] = {0, 0, 0};
        for (int64_t q = pt_ptr[l]; q < pt_ptr[l + 1]; ++q) {
          const int64_t o = pt_obs[q];
          const T x0 = dchi2[o] * res[2 * o], x1 = dchi2[o] * res[2 * o + 1];
          for (int k = 0; k < 3; ++k) v[k] -= Jp[6 * o + 2 * k] * x0 + Jp[6 * o + 2 * k + 1] * x1;
        }
        for (int k = 0; k < 3; ++k) b[pose_dim + 3 * l + k] = v[k];
      }
    }
  }
  // blocks of J^T rho' J into `hc` (81 per camera) / `hl` (9 per point) and, when hcp != null, Hcp
  void blocks_mt(T *hc, T *hl, T *hcp) {
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(dynamic, 16) nowait
      for (int64_t c = 0; c < (int64_t)Nc; ++c) {
        T h[81];
        for (int k = 0; k < 81; ++k) h[k] = 0;
        for (int64_t q = cam_ptr[c]; q < cam_ptr[c + 1]; ++q) {
          const int64_t o = cam_obs[q];
          const T *jc = &Jc[18 * o];
          const T w = dchi2[o];
          for (int col = 0; col < 9; ++col)
            for (int row = 0; row < 9; ++row) h[row + 9 * col] += (jc[2 * row] * jc[2 * col] + jc[2 * row + 1] * jc[2 * col + 1]) * w;
        }
        std::copy(h, h + 81, hc + 81 * c);
      }
#pragma omp for schedule(static) nowait
      for (int64_t l = 0; l < (int64_t)Np; ++l) {
        T h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t q = pt_ptr[l]; q < pt_ptr[l + 1]; ++q) {
          const int64_t o = pt_obs[q];
          const T *jp = &Jp[6 * o];
          const T w = dchi2[o];
          for (int col = 0; col < 3; ++col)
            for (int row = 0; row < 3; ++row) h[row + 3 * col] += (jp[2 * row] * jp[2 * col] + jp[2 * row + 1] * jp[2 * col + 1]) * w;
        }
        std::copy(h, h + 9, hl + 9 * l);
      }
      if (hcp) {
#pragma omp for schedule(static)
        for (int64_t o = 0; o < (int64_t)No; ++o) {
          const T *jc = &Jc[18 * o], *jp = &Jp[6 * o];
          const T w = dchi2[o];
          for (int col = 0; col < 3; ++col)
            for (int row = 0; row < 9; ++row) hcp[27 * o + row + 9 * col] = (jc[2 * row] * jp[2 * col] + jc[2 * row + 1] * jp[2 * col + 1]) * w;
        }
      }
    }
  }
  void hessian_update_values_mt() { // hessian.hpp:290-307
    blocks_mt(Hcc.data(), Hll.data(), Hcp.data());
    for (size_t c = 0; c < Nc; ++c) for (int i = 0; i < 9; ++i) prev_diag[9 * c + i] = Hcc[81 * c + 10 * i];
    for (size_t l = 0; l < Np; ++l) for (int i = 0; i < 3; ++i) prev_diag[pose_dim + 3 * l + i] = Hll[9 * l + 4 * i];
  }
  void schur_update_values_mt() { // schur.hpp:227-235; one thread owns one block COLUMN of S
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t l = 0; l < (int64_t)Np; ++l) small_inverse(3, &Hll[9 * l], &Hll_inv[9 * l]);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
    for (int64_t j = 0; j < (int64_t)Nc; ++j) {
      for (int64_t k = S_colptr[j]; k < S_colptr[j + 1]; ++k) std::fill(&S[81 * k], &S[81 * k] + 81, T(0));
      std::copy(&Hcc[81 * j], &Hcc[81 * j] + 81, &S[81 * this->s_block(j, j)]);
      for (int64_t q = cam_ptr[j]; q < cam_ptr[j + 1]; ++q) {
        const int64_t ob = cam_obs[q], l = pt_idx[ob];
        const T *M = &Hll_inv[9 * l], *R = &Hcp[27 * ob];
        T MRt[27]; // M R^T, 3 x 9
        for (int col = 0; col < 9; ++col)
          for (int k = 0; k < 3; ++k) {
            T s = 0;
            for (int jj = 0; jj < 3; ++jj) s += M[k + 3 * jj] * R[col + 9 * jj];
            MRt[k + 3 * col] = s;
          }
        for (int64_t a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) {
          const int64_t oa = pt_obs[a];
          if (cam_idx[oa] > j) break; // pt_obs is sorted by camera inside a point
          const T *L = &Hcp[27 * oa];
          T *dst = &S[81 * this->s_block(cam_idx[oa], j)];
          for (int col = 0; col < 9; ++col)
            for (int row = 0; row < 9; ++row) {
              T v = 0;
              for (int k = 0; k < 3; ++k) v += L[row + 9 * k] * MRt[k + 3 * col];
              dst[row + 9 * col] -= v;
            }
        }
      }
    }
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
    for (int64_t c = 0; c < (int64_t)Nc; ++c) { // b_S = b_p - Hpl Hll^-1 b_l  (:901)
      T acc[9];
      for (int r = 0; r < 9; ++r) acc[r] = b[9 * c + r];
      for (int64_t q = cam_ptr[c]; q < cam_ptr[c + 1]; ++q) {
        const int64_t o = cam_obs[q], l = pt_idx[o];
        T v[3];
        for (int r = 0; r < 3; ++r) { v[r] = 0; for (int k = 0; k < 3; ++k) v[r] += Hll_inv[9 * l + r + 3 * k] * b[pose_dim + 3 * l + k]; }
        for (int r = 0; r < 9; ++r) { T s = 0; for (int k = 0; k < 3; ++k) s += Hcp[27 * o + r + 9 * k] * v[k]; acc[r] -= s; }
      }
      for (int r = 0; r < 9; ++r) b_schur[9 * c + r] = acc[r];
    }
  }
  void landmark_update_mt(const T *xp, T *xl) const {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t l = 0; l < (int64_t)Np; ++l) {
      T rhs[3] = {b[pose_dim + 3 * l], b[pose_dim + 3 * l + 1], b[pose_dim + 3 * l + 2]};
      for (int64_t a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) {
        const int64_t o = pt_obs[a];
        for (int c = 0; c < 3; ++c) { T s = 0; for (int r = 0; r < 9; ++r) s += Hcp[27 * o + r + 9 * c] * xp[9 * cam_idx[o] + r]; rhs[c] -= s; }
      }
      for (int r = 0; r < 3; ++r) { T s = 0; for (int c = 0; c < 3; ++c) s += Hll_inv[9 * l + r + 3 * c] * rhs[c]; xl[3 * l + r] = s; }
    }
  }

  // ---- matrix-free PCG, every stage parallel (solver/pcg.hpp:61-232) ----
  T dot_mt(size_t m, const T *a, const T *c) const {
    const int nt = threads;
    std::vector<T> part(nt, 0);
#pragma omp parallel num_threads(nt)
    {
      const int t = omp_get_thread_num();
      const size_t lo = m * t / nt, hi = m * (t + 1) / nt;
      part[t] = tree_sum<T>(lo, hi, [&](size_t i) { return a[i] * c[i]; });
    }
    T s = 0;
    for (int t = 0; t < nt; ++t) s += part[t];
    return s;
  }
  void JtJ_matvec_mt(const T *p, T *v2, std::vector<T> &v1) const {
    v1.resize(2 * No);
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(static)
      for (int64_t o = 0; o < (int64_t)No; ++o) {
        const T *pc = &p[9 * cam_idx[o]], *pp = &p[pose_dim + 3 * pt_idx[o]];
        for (int e = 0; e < 2; ++e) {
          T s = 0, s2 = 0;
          for (int i = 0; i < 9; ++i) s += Jc[18 * o + e + 2 * i] * pc[i];
          for (int i = 0; i < 3; ++i) s2 += Jp[6 * o + e + 2 * i] * pp[i];
          v1[2 * o + e] = s + s2;
        }
      }
#pragma omp for schedule(dynamic, 16) nowait
      for (int64_t c = 0; c < (int64_t)Nc; ++c) {
        T y[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t q = cam_ptr[c]; q < cam_ptr[c + 1]; ++q) {
          const int64_t o = cam_obs[q];
          for (int k = 0; k < 9; ++k) y[k] += (Jc[18 * o + 2 * k] * v1[2 * o] + Jc[18 * o + 2 * k + 1] * v1[2 * o + 1]) * dchi2[o];
        }
        for (int k = 0; k < 9; ++k) v2[9 * c + k] = y[k];
      }
#pragma omp for schedule(static)
      for (int64_t l = 0; l < (int64_t)Np; ++l) {
        T y[3] = {0, 0, 0};
        for (int64_t q = pt_ptr[l]; q < pt_ptr[l + 1]; ++q) {
          const int64_t o = pt_obs[q];
          for (int k = 0; k < 3; ++k) y[k] += (Jp[6 * o + 2 * k] * v1[2 * o] + Jp[6 * o + 2 * k + 1] * v1[2 * o + 1]) * dchi2[o];
        }
        for (int k = 0; k < 3; ++k) v2[pose_dim + 3 * l + k] = y[k];
      }
    }
  }
  void block_jacobi_update_values_mt() {
    bj_blocks.assign(81 * Nc + 9 * Np, 0);
    blocks_mt(bj_blocks.data(), bj_blocks.data() + 81 * Nc, nullptr);
    bj_scalar.assign(n, 0);
    for (size_t c = 0; c < Nc; ++c) for (int i = 0; i < 9; ++i) bj_scalar[9 * c + i] = bj_blocks[81 * c + 10 * i];
    for (size_t l = 0; l < Np; ++l) for (int i = 0; i < 3; ++i) bj_scalar[pose_dim + 3 * l + i] = bj_blocks[81 * Nc + 9 * l + 4 * i];
  }
  void block_jacobi_set_damping_mt(T mu, bool use_identity) {
    bj_inv.resize(81 * Nc + 9 * Np);
    auto damp = [&](T d) {
      if (use_identity) return (T)((double)d + (double)mu);
      return (T)((double)d + (double)mu * std::clamp((double)d, 1.0e-6, 1.0e32));
    };
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(static) nowait
      for (int64_t c = 0; c < (int64_t)Nc; ++c) {
        for (int i = 0; i < 9; ++i) bj_blocks[81 * c + 10 * i] = damp(bj_scalar[9 * c + i]);
        small_inverse(9, &bj_blocks[81 * c], &bj_inv[81 * c]);
      }
#pragma omp for schedule(static)
      for (int64_t l = 0; l < (int64_t)Np; ++l) {
        for (int i = 0; i < 3; ++i) bj_blocks[81 * Nc + 9 * l + 4 * i] = damp(bj_scalar[pose_dim + 3 * l + i]);
        small_inverse(3, &bj_blocks[81 * Nc + 9 * l], &bj_inv[81 * Nc + 9 * l]);
      }
    }
  }
  void block_jacobi_apply_mt(T *z, const T *r) const {
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(static) nowait
      for (int64_t c = 0; c < (int64_t)Nc; ++c)
        for (int row = 0; row < 9; ++row) { T s = 0; for (int k = 0; k < 9; ++k) s += bj_inv[81 * c + row + 9 * k] * r[9 * c + k]; z[9 * c + row] = s; }
#pragma omp for schedule(static)
      for (int64_t l = 0; l < (int64_t)Np; ++l)
        for (int row = 0; row < 3; ++row) { T s = 0; for (int k = 0; k < 3; ++k) s += bj_inv[81 * Nc + 9 * l + row + 3 * k] * r[pose_dim + 3 * l + k]; z[pose_dim + 3 * l + row] = s; }
    }
  }
  bool solve_pcg_mt(T *x, int max_iter, T tol, T rejection_ratio) {
    std::vector<T> v1, v2(n), r(b), p(n), z(n), diag(n), xb(n), y(n);
    std::fill(x, x + n, T(0));
    for (size_t i = 0; i < n; ++i) diag[i] = std::clamp(bj_scalar[i], T(1.0e-6), T(1.0e32)); // = diag(J^T rho' J), pcg.hpp:93-103
    T rnorm = std::sqrt(dot_mt(n, r.data(), r.data()));
    T scale = T(1.0 / rnorm);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) y[i] = scale * r[i];
    block_jacobi_apply_mt(z.data(), y.data());
    p = z;
    T rz = dot_mt(n, r.data(), z.data());
    T rz_0 = std::numeric_limits<T>::infinity();
    this->last_pcg_iters = 0;
    for (int k = 0; k < max_iter; ++k) {
      if (rz == 0) break;
      JtJ_matvec_mt(p.data(), v2.data(), v1);
#pragma omp parallel for num_threads(threads) schedule(static)
      for (int64_t i = 0; i < (int64_t)n; ++i) v2[i] += damping_identity ? damping * p[i] : damping * diag[i] * p[i];
      this->last_pcg_iters++;
      const T alpha = rz / dot_mt(n, p.data(), v2.data());
#pragma omp parallel for num_threads(threads) schedule(static)
      for (int64_t i = 0; i < (int64_t)n; ++i) { xb[i] = x[i]; x[i] = alpha * p[i] + x[i]; r[i] = -alpha * v2[i] + r[i]; }
      rnorm = std::sqrt(dot_mt(n, r.data(), r.data()));
      scale = T(1.0 / rnorm);
#pragma omp parallel for num_threads(threads) schedule(static)
      for (int64_t i = 0; i < (int64_t)n; ++i) y[i] = scale * r[i];
      block_jacobi_apply_mt(z.data(), y.data());
      const T rz_new = dot_mt(n, r.data(), z.data());
      if (std::abs(rz_new) > rejection_ratio * rz_0 || std::isnan(rz_new)) { std::copy(xb.begin(), xb.end(), x); break; }
      rz_0 = std::min(rz_0, std::abs(rz_new));
      const T beta = rz_new / rz;
      rz = rz_new;
#pragma omp parallel for num_threads(threads) schedule(static)
      for (int64_t i = 0; i < (int64_t)n; ++i) p[i] = beta * p[i] + z[i];
      if (std::abs(rz_new) < tol) break;
    }
    return true;
  }

  // ---- the timed LM (optimizer/levenberg_marquardt.hpp:110-242), solver = SOLVER_LDLT | SOLVER_LDLT_SCHUR | SOLVER_PCG ----
  using clk = std::chrono::steady_clock;
  static double since(clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); }

  bool solve_direct(int kind, T *x) {
    auto t = clk::now();
    if (kind == SOLVER_LDLT_SCHUR) { schur_update_values_mt(); tm.schur += since(t); t = clk::now(); B::export_schur_csc(csc_p, csc_i, csc_x); }
    else B::export_hessian_csc(csc_p, csc_i, csc_x);
    tm.export_csc += since(t);
    if (!analyzed || analyzed_kind != kind) {
      t = clk::now();
      std::vector<int64_t> perm;
      if (ordering == 2) {
        // AMD of the matrix that is factorised, as Eigen::SimplicialLDLT's default AMDOrdering does on its scalar pattern
        // (src/eigen_solver.cpp:10-13; amd.hpp) — for S and for the full H alike
        perm = amd_order((int64_t)(kind == SOLVER_LDLT ? n : pose_dim), csc_p.data(), csc_i.data());
      } else {
        if (kind == SOLVER_LDLT) for (size_t l = 0; l < Np; ++l) for (int k = 0; k < 3; ++k) perm.push_back(pose_dim + 3 * l + k);
        for (int64_t c : camera_order()) for (int k = 0; k < 9; ++k) perm.push_back(9 * c + k);
      }
      ldlt.analyze(kind == SOLVER_LDLT ? n : pose_dim, csc_p.data(), csc_i.data(), perm);
      analyzed = true; analyzed_kind = kind;
      tm.ldlt_analyze += since(t);
      tm.ldlt_nnz = (int64_t)ldlt.Li.size();
    }
    t = clk::now();
    const bool ok = ldlt.factorize(csc_x.data()); // ONE thread: the part the reference runs on the CPU (src/eigen_solver.cpp:21-29)
    tm.ldlt_factor += since(t);
    if (!ok) return false;
    t = clk::now();
    std::fill(x, x + n, T(0));
    const bool ok2 = ldlt.solve(kind == SOLVER_LDLT ? b.data() : b_schur.data(), x);
    tm.ldlt_solve += since(t);
    if (!ok2) return false;
    if (kind == SOLVER_LDLT_SCHUR) { t = clk::now(); landmark_update_mt(x, x + pose_dim); tm.backsub += since(t); }
    return true;
  }
  void update_values(int kind) {
    auto t = clk::now();
    if (kind == SOLVER_PCG) block_jacobi_update_values_mt(); else hessian_update_values_mt();
    tm.hessian += since(t);
  }

  bool levenberg_marquardt_mt(const LMOptions &opt, std::vector<double> &chi2_trace, std::vector<double> &lambda_trace, LMStats &st) {
    auto t0 = clk::now();
    tm = BaselineTimes();
    T mu = (T)opt.initial_damping, nu = 2;
    analyzed = false;
    auto t = clk::now();
    linearize_mt();
    tm.linearize += since(t);
    update_values(opt.solver);
    T chi2v = chi2_mt();
    std::vector<T> dx(n, 0);
    bool run = true;
    st = LMStats();
    st.setup_seconds = tm.setup = since(t0);
    chi2_trace.assign(1, (double)chi2v); lambda_trace.assign(1, (double)mu);
    auto tl = clk::now();
    for (int i = 0; i < opt.iterations && run; ++i) {
      damping = mu; damping_identity = opt.use_identity != 0;
      bool solve_ok;
      auto ts = clk::now();
      if (opt.solver == SOLVER_PCG) {
        block_jacobi_set_damping_mt(mu, damping_identity);
        solve_ok = solve_pcg_mt(dx.data(), opt.pcg_max_iter, (T)opt.pcg_tol, (T)opt.pcg_rejection_ratio);
        tm.pcg += since(ts);
      } else {
        B::apply_damping(mu, damping_identity);
        this->last_pcg_iters = 0;
        solve_ok = solve_direct(opt.solver, dx.data());
      }
      st.solve_seconds += since(ts);
      st.pcg_iterations += this->last_pcg_iters;
      t = clk::now();
      B::backup();
      B::apply_update(dx.data());
      compute_error_mt();
      T new_chi2 = chi2_mt();
      if (!solve_ok) new_chi2 = std::numeric_limits<T>::max();
      T num = chi2v - new_chi2, denom = 1.0;
      if (solve_ok) { denom = tree_sum<T>(0, n, [&](size_t k) { return dx[k] * (mu * dx[k] + b[k]); }); denom += T(1.0e-3); }
      const T rho = num / denom;
      tm.update_chi2 += since(t);
      if (solve_ok && std::isfinite(new_chi2) && rho > 0) {
        double alpha = 1.0 - std::pow(2.0 * rho - 1.0, 3);
        alpha = std::max(std::min(alpha, 2.0 / 3.0), 1.0 / 3.0);
        mu *= (T)alpha; nu = 2;
        t = clk::now(); linearize_mt(); tm.linearize += since(t);
        update_values(opt.solver);
        st.accepted++;
      } else {
        B::revert(); compute_error_mt(); chi2_mt();
        mu *= nu; nu *= 2; new_chi2 = chi2v;
      }
      chi2v = new_chi2;
      st.iterations_run++;
      chi2_trace.push_back((double)chi2v); lambda_trace.push_back((double)mu);
      if (!std::isfinite(mu)) run = false;
      if (rho == 0) break;
    }
    st.loop_seconds = tm.loop = since(tl);
    return run;
  }
};

} // namespace gro
