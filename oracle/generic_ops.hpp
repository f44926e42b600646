// ORACLE — test infrastructure only (see bal_model.hpp header).
//
// CPU restatement of the reference's generic per-factor device kernels, one
// function per kernel, same argument meaning.  All Jacobian slot arrays are
// "E x d column-major per factor" (ops/error.hpp:146-149); precision matrices
// are read row-major p[i*E+j] (ops/linearize.hpp:283).  Sums are sequential in
// factor order (the reference's float atomics have no defined order, so any
// order is a valid instance; optimizer/levenberg_marquardt.hpp:372).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace gro {

// active.hpp:18-21 — a vertex takes part iff its state byte is 0
// (bit0 = fixed, bit7 = not touched by an active factor).
inline bool vertex_active(const uint8_t *state, size_t id) { return state[id] == 0; }

enum LossKind : int { LOSS_DEFAULT = 0, LOSS_HUBER = 1 };

// loss.hpp:15-51
template <typename T> inline T loss_value(int kind, T delta, T x) {
  if (kind == LOSS_HUBER && !(x <= delta * delta)) return 2 * std::sqrt(x) * delta - delta * delta;
  return x;
}
template <typename T> inline T loss_derivative(int kind, T delta, T x) {
  if (kind == LOSS_HUBER && !(x <= delta * delta)) return delta / std::sqrt(x);
  return T(1);
}

// ops/chi2.hpp:10-44 — raw = r^T P r, chi2 = rho(raw), dchi2 = rho'(raw).
// NOTE the reference indexes the FIRST `n` factors, not active_indices
// (chi2.hpp:36-43) — kept.
template <typename T>
void chi2_kernel(size_t n, int E, const T *residuals, const T *pmat, const int *loss_kind,
                 const T *loss_delta, T *chi2, T *dchi2) {
  for (size_t f = 0; f < n; ++f) {
    T value = 0;
    for (int i = 0; i < E; ++i) {
      T r2 = 0;
      for (int j = 0; j < E; ++j) r2 += pmat[f * E * E + i * E + j] * residuals[f * E + j];
      value += r2 * residuals[f * E + i];
    }
    const int kind = loss_kind ? loss_kind[f] : LOSS_DEFAULT;
    const T delta = loss_delta ? loss_delta[f] : T(0);
    chi2[f] = loss_value(kind, delta, value);
    dchi2[f] = loss_derivative(kind, delta, value);
  }
}

// Common view of one vertex slot of a factor descriptor.
template <typename T> struct Slot {
  int d;                      // vertex dimension
  const T *jac;               // [nf_internal][E*d] column-major blocks
  const size_t *ids;          // ids[f*N + I] local vertex id
  size_t N, I;
  const size_t *hessian_ids;  // scalar column per local vertex
  const uint8_t *active_state;
};

template <typename T> inline T jtpj(int E, const T *Jt, const T *J, const T *P) {
  T value = 0;
  for (int i = 0; i < E; ++i) {
    T pj = 0;
    for (int j = 0; j < E; ++j) pj += P[i * E + j] * J[j];
    value += Jt[i] * pj;
  }
  return value;
}

// ops/hessian.hpp:419-474  diag[col] += rho' * J[:,c]^T P J[:,c]
template <typename T>
void scalar_diagonal_kernel(size_t n_active, const size_t *active_ids, int E, const Slot<T> &s,
                            const T *pmat, const T *dchi2, T *diagonal) {
  for (size_t a = 0; a < n_active; ++a) {
    const size_t f = active_ids[a];
    const size_t v = s.ids[f * s.N + s.I];
    if (!vertex_active(s.active_state, v)) continue;
    const T *J = s.jac + f * E * s.d;
    for (int c = 0; c < s.d; ++c)
      diagonal[s.hessian_ids[v] + c] += jtpj(E, J + c * E, J + c * E, pmat + f * E * E) * dchi2[f];
  }
}

// ops/hessian.hpp:171-270  per-vertex d x d block (column-major) += rho' J^T P J
template <typename T>
void block_diagonal_kernel(size_t n_active, const size_t *active_ids, int E, const Slot<T> &s,
                           const T *pmat, const T *dchi2, T *blocks) {
  for (size_t a = 0; a < n_active; ++a) {
    const size_t f = active_ids[a];
    const size_t v = s.ids[f * s.N + s.I];
    if (!vertex_active(s.active_state, v)) continue;
    const T *J = s.jac + f * E * s.d;
    for (int col = 0; col < s.d; ++col)
      for (int row = 0; row < s.d; ++row)
        blocks[v * s.d * s.d + row + col * s.d] +=
            jtpj(E, J + row * E, J + col * E, pmat + f * E * E) * dchi2[f];
  }
}

// ops/linearize.hpp:142-180  J[:,c] *= scale[col]
template <typename T>
void scale_jacobians_kernel(size_t n_active, const size_t *active_ids, int E, int d, T *jac,
                            const size_t *ids, size_t N, size_t I, const size_t *hessian_ids,
                            const uint8_t *active_state, const T *scales) {
  for (size_t a = 0; a < n_active; ++a) {
    const size_t f = active_ids[a];
    const size_t v = ids[f * N + I];
    if (!vertex_active(active_state, v)) continue;
    for (int c = 0; c < d; ++c)
      for (int i = 0; i < E; ++i) jac[f * E * d + c * E + i] *= scales[hessian_ids[v] + c];
  }
}

// ops/linearize.hpp:240-303  b[col] -= J[:,c]^T (rho' P r)
template <typename T>
void compute_b_kernel(size_t n_active, const size_t *active_ids, int E, const Slot<T> &s,
                      const T *residuals, const T *pmat, const T *dchi2, T *b) {
  std::vector<T> x2(E);
  for (size_t a = 0; a < n_active; ++a) {
    const size_t f = active_ids[a];
    const size_t v = s.ids[f * s.N + s.I];
    if (!vertex_active(s.active_state, v)) continue;
    for (int i = 0; i < E; ++i) {
      x2[i] = 0;
      for (int j = 0; j < E; ++j) x2[i] += dchi2[f] * pmat[f * E * E + i * E + j] * residuals[f * E + j];
    }
    for (int c = 0; c < s.d; ++c) {
      T value = 0;
      for (int i = 0; i < E; ++i) value -= s.jac[f * E * s.d + c * E + i] * x2[i];
      b[s.hessian_ids[v] + c] += value;
    }
  }
}

// ops/product.hpp:51-100  y[a*E+e] += J[e,:] x[col:]   (y indexed by THREAD id,
// i.e. position in the active list, product.hpp:66,97 — kept)
template <typename T>
void Jv_kernel(size_t n_active, const size_t *active_ids, int E, const Slot<T> &s, const T *x, T *y) {
  for (size_t a = 0; a < n_active; ++a) {
    const size_t f = active_ids[a];
    const size_t v = s.ids[f * s.N + s.I];
    if (!vertex_active(s.active_state, v)) continue;
    for (int e = 0; e < E; ++e) {
      T value = 0;
      for (int i = 0; i < s.d; ++i) value += s.jac[f * E * s.d + e + i * E] * x[s.hessian_ids[v] + i];
      y[a * E + e] += value;
    }
  }
}

// ops/product.hpp:228-289  y[col+c] += rho' J[:,c]^T P x_f   (x indexed by factor id)
template <typename T>
void JtPv_kernel(size_t n_active, const size_t *active_ids, int E, const Slot<T> &s, const T *pmat,
                 const T *dchi2, const T *x, T *y) {
  for (size_t a = 0; a < n_active; ++a) {
    const size_t f = active_ids[a];
    const size_t v = s.ids[f * s.N + s.I];
    if (!vertex_active(s.active_state, v)) continue;
    for (int c = 0; c < s.d; ++c) {
      T value = 0;
      for (int i = 0; i < E; ++i) {
        T x2 = 0;
        for (int j = 0; j < E; ++j) x2 += pmat[f * E * E + i * E + j] * x[f * E + j];
        value += s.jac[f * E * s.d + c * E + i] * x2;
      }
      y[s.hessian_ids[v] + c] += value * dchi2[f];
    }
  }
}

// ops/hessian.hpp:10-78  one (slot i, slot j) pass of the upper-triangular
// block Hessian: block += rho' Ji^T P Jj, transposed when column(vi) > column(vj).
// block_offsets[a] = value offset of the destination block for active factor a.
template <typename T>
void hessian_block_kernel(size_t n_active, const size_t *active_ids, int E, const Slot<T> &si,
                          const Slot<T> &sj, const size_t *block_offsets, const T *pmat,
                          const T *dchi2, T *hessian) {
  for (size_t a = 0; a < n_active; ++a) {
    const size_t f = active_ids[a];
    const size_t vi = si.ids[f * si.N + si.I], vj = sj.ids[f * sj.N + sj.I];
    if (!(vertex_active(si.active_state, vi) && vertex_active(sj.active_state, vj))) continue;
    const bool transposed = si.hessian_ids[vi] > sj.hessian_ids[vj];
    const T *ji = si.jac + f * E * si.d;
    const T *jj = sj.jac + f * E * sj.d;
    int di = si.d, dj = sj.d;
    if (transposed) { std::swap(ji, jj); std::swap(di, dj); }
    for (int col = 0; col < dj; ++col)
      for (int row = 0; row < di; ++row)
        hessian[block_offsets[a] + row + col * di] +=
            jtpj(E, ji + row * E, jj + col * E, pmat + f * E * E) * dchi2[f];
  }
}

// ops/hessian.hpp:81-110  D <- d + mu*clamp(d,1e-6,1e32)  (or d + mu), in double
template <typename T>
void augment_block_diagonal_kernel(size_t nv, int D, T *blocks, const T *scalar_diag, T mu,
                                   bool use_identity, const uint8_t *active_state) {
  for (size_t v = 0; v < nv; ++v) {
    if (!vertex_active(active_state, v)) continue;
    for (int i = 0; i < D; ++i) {
      const double diag = static_cast<double>(scalar_diag[v * D + i]);
      double nd = diag;
      if (use_identity) nd += static_cast<double>(mu);
      else nd += static_cast<double>(mu) * std::clamp(diag, 1.0e-6, 1.0e32);
      blocks[v * D * D + i * D + i] = static_cast<T>(nd);
    }
  }
}

// ops/hessian.hpp:128-153  z[col+row] = sum_i block[row + i*D] r[col+i]
template <typename T>
void apply_block_jacobi_kernel(size_t nv, int D, T *z, const T *r, const T *blocks,
                               const size_t *hessian_ids, const uint8_t *active_state) {
  for (size_t v = 0; v < nv; ++v) {
    if (!vertex_active(active_state, v)) continue;
    for (int row = 0; row < D; ++row) {
      T value = 0;
      for (int i = 0; i < D; ++i) value += blocks[v * D * D + row + i * D] * r[hessian_ids[v] + i];
      z[hessian_ids[v] + row] = value;
    }
  }
}

// ops/update.hpp:11-31 with the additive `update` used by every traits in the
// reference's tests/examples (vertex += delta*scale)
template <typename T>
void apply_update_kernel(size_t nv, int D, T *params, const T *delta_x, const T *scales,
                         const size_t *hessian_ids, const uint8_t *active_state) {
  for (size_t v = 0; v < nv; ++v) {
    if (!vertex_active(active_state, v)) continue;
    for (int i = 0; i < D; ++i)
      params[v * D + i] += delta_x[hessian_ids[v] + i] * scales[hessian_ids[v] + i];
  }
}

// General small inverse (the role of cublas<t>matinvBatched, schur.hpp:1101;
// CUDA 12.8 un-vendored).  Gauss-Jordan with partial pivoting, column-major
// in/out.  Returns false on a zero pivot.
template <typename T> bool small_inverse(int n, const T *A, T *Ainv) {
  std::vector<double> M(n * 2 * n);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) {
      M[r * 2 * n + c] = static_cast<double>(A[r + c * n]);
      M[r * 2 * n + n + c] = (r == c) ? 1.0 : 0.0;
    }
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int r = k + 1; r < n; ++r)
      if (std::fabs(M[r * 2 * n + k]) > std::fabs(M[piv * 2 * n + k])) piv = r;
    if (M[piv * 2 * n + k] == 0.0) return false;
    if (piv != k)
      for (int c = 0; c < 2 * n; ++c) std::swap(M[k * 2 * n + c], M[piv * 2 * n + c]);
    const double inv = 1.0 / M[k * 2 * n + k];
    for (int c = 0; c < 2 * n; ++c) M[k * 2 * n + c] *= inv;
    for (int r = 0; r < n; ++r) {
      if (r == k) continue;
      const double fct = M[r * 2 * n + k];
      if (fct == 0.0) continue;
      for (int c = 0; c < 2 * n; ++c) M[r * 2 * n + c] -= fct * M[k * 2 * n + c];
    }
  }
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) Ainv[r + c * n] = static_cast<T>(M[r * 2 * n + n + c]);
  return true;
}

} // namespace gro
