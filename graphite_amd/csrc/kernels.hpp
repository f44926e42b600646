// HIP kernels of the BAL hot path, written for gfx950 (wave64, 256 CUs).
//
// Data layout in HBM (all arrays resident for the whole optimisation):
//   cams[Nc][9], pts[Np][3]                 vertex values (T)
//   pack[Nc][24]                            per-camera R,t,f,k1,k2,Jr (bal_device.hpp)
//   point-major ("pm") observation order: observations sorted by point, then camera
//     obs_pm[No][2], cam_pm[No], pt_pm[No], pt_ptr[Np+1]
//   camera-major ("cm") order: the same observations sorted by camera, then point
//     obs_cm[No][2], pt_cm[No], pos_cm[No] (-> pm position), cam_ptr[Nc+1]
//   UNSCALED blocks of J^T rho' J and -J^T rho' r:
//     Hcc[Nc][81], bc[Nc][9], Hll[Np][9], bl[Np][3], Hcp[No][27] (pm order, 9x3 col-major)
//   scales[n]  column scales s = 1/(eps + sqrt(diag))           (graph.hpp:262-270)
// The reference scales the stored Jacobians in place and multiplies afterwards
// (graph.hpp:254-287); here every kernel works on UNSCALED blocks and the
// diagonal congruence H = D H^u D, b = D b^u is folded into the small
// per-vertex matrices (3x3 / 9-vectors) where it is needed.  Two sorted copies
// of the observation list let every reduction be a segmented sum (no float
// atomics on the hot path; results are run-to-run deterministic).
#pragma once
#include "bal_device.hpp"
#include "common.hpp"
#include <cfloat>

namespace gr {

constexpr int TPB = 256;

template <typename T> struct Vec2T;
template <> struct Vec2T<float> { using type = float2; };
template <> struct Vec2T<double> { using type = double2; };

// ---------------------------------------------------------------------------
// camera pack: Nc threads
template <typename T>
__global__ void k_campack(int Nc, const T *__restrict__ cams, T *__restrict__ pack) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Nc) return;
  T cam[9], pk[PACK];
#pragma unroll
  for (int i = 0; i < 9; ++i) cam[i] = cams[9 * c + i];
  make_campack(cam, pk);
#pragma unroll
  for (int i = 0; i < PACK; ++i) pack[PACK * c + i] = pk[i];
}

// ---------------------------------------------------------------------------
// Point pass of Graph::linearize + Hessian::update_values (A4-A10 of SURVEY §8a):
// one thread per point walks its observations (pm order) and produces
//   Hll^u (3x3), bl^u (3), Hcp^u per observation (optional), chi2 partial.
template <typename T, bool WRITE_HCP>
__global__ void __launch_bounds__(TPB)
k_point_linearize(int Np, const int *__restrict__ pt_ptr, const int *__restrict__ cam_pm,
                  const T *__restrict__ obs_pm, const T *__restrict__ pts,
                  const T *__restrict__ pack, int loss_kind, T loss_delta, T *__restrict__ Hll,
                  T *__restrict__ bl, T *__restrict__ Hcp, double *__restrict__ chi2_partial) {
  __shared__ double red[4];
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  double chi2 = 0.0;
  if (l < Np) {
    const T X = pts[3 * l], Y = pts[3 * l + 1], Z = pts[3 * l + 2];
    T h00 = 0, h01 = 0, h02 = 0, h11 = 0, h12 = 0, h22 = 0, b0 = 0, b1 = 0, b2 = 0;
    const int beg = pt_ptr[l], end = pt_ptr[l + 1];
    for (int a = beg; a < end; ++a) {
      const int c = cam_pm[a];
      T pk[PACK];
#pragma unroll
      for (int i = 0; i < PACK; ++i) pk[i] = pack[PACK * c + i];
      const typename Vec2T<T>::type o = reinterpret_cast<const typename Vec2T<T>::type *>(obs_pm)[a];
      T e0, e1, Jc[18], Jp[6];
      bal_linearize(pk, X, Y, Z, o.x, o.y, e0, e1, Jc, Jp);
      const T raw = e0 * e0 + e1 * e1;
      const T w = loss_drho(loss_kind, loss_delta, raw);
      chi2 += (double)loss_rho(loss_kind, loss_delta, raw);
      const T wp0x = w * Jp[0], wp0y = w * Jp[1], wp1x = w * Jp[2], wp1y = w * Jp[3], wp2x = w * Jp[4], wp2y = w * Jp[5];
      h00 += wp0x * Jp[0] + wp0y * Jp[1];
      h01 += wp0x * Jp[2] + wp0y * Jp[3];
      h02 += wp0x * Jp[4] + wp0y * Jp[5];
      h11 += wp1x * Jp[2] + wp1y * Jp[3];
      h12 += wp1x * Jp[4] + wp1y * Jp[5];
      h22 += wp2x * Jp[4] + wp2y * Jp[5];
      b0 -= wp0x * e0 + wp0y * e1;
      b1 -= wp1x * e0 + wp1y * e1;
      b2 -= wp2x * e0 + wp2y * e1;
      if (WRITE_HCP) {
        T *h = Hcp + 27 * (size_t)a;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          h[r] = Jc[2 * r] * wp0x + Jc[2 * r + 1] * wp0y;
          h[r + 9] = Jc[2 * r] * wp1x + Jc[2 * r + 1] * wp1y;
          h[r + 18] = Jc[2 * r] * wp2x + Jc[2 * r + 1] * wp2y;
        }
      }
    }
    T *H = Hll + 9 * (size_t)l;
    H[0] = h00; H[1] = h01; H[2] = h02; H[3] = h01; H[4] = h11; H[5] = h12; H[6] = h02; H[7] = h12; H[8] = h22;
    bl[3 * l] = b0; bl[3 * l + 1] = b1; bl[3 * l + 2] = b2;
  }
  const double tot = block_sum_256(chi2, red);
  if (threadIdx.x == 0) chi2_partial[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// Camera pass: one 256-thread block per camera walks its observations (cm
// order), recomputes residual + Jc, and block-reduces Hcc^u (45 unique) and
// bc^u (9).  The pack address is block-uniform, so it sits in SGPRs.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_camera_linearize(const int *__restrict__ cam_ptr, const int *__restrict__ pt_cm,
                   const T *__restrict__ obs_cm, const T *__restrict__ pts,
                   const T *__restrict__ pack, int loss_kind, T loss_delta, T *__restrict__ Hcc,
                   T *__restrict__ bc) {
  __shared__ T red[4][54];
  const int c = blockIdx.x;
  T pk[PACK];
#pragma unroll
  for (int i = 0; i < PACK; ++i) pk[i] = pack[PACK * c + i];
  T acc[54];
#pragma unroll
  for (int i = 0; i < 54; ++i) acc[i] = T(0);
  const int beg = cam_ptr[c], end = cam_ptr[c + 1];
  for (int j = beg + threadIdx.x; j < end; j += TPB) {
    const int l = pt_cm[j];
    const T X = pts[3 * l], Y = pts[3 * l + 1], Z = pts[3 * l + 2];
    const typename Vec2T<T>::type o = reinterpret_cast<const typename Vec2T<T>::type *>(obs_cm)[j];
    T e0, e1, Jc[18], Jp[6];
    bal_linearize(pk, X, Y, Z, o.x, o.y, e0, e1, Jc, Jp);
    const T w = loss_drho(loss_kind, loss_delta, e0 * e0 + e1 * e1);
    int k = 0;
#pragma unroll
    for (int col = 0; col < 9; ++col) {
      const T wx = w * Jc[2 * col], wy = w * Jc[2 * col + 1];
#pragma unroll
      for (int row = 0; row <= col; ++row) acc[k++] += Jc[2 * row] * wx + Jc[2 * row + 1] * wy;
      acc[45 + col] -= wx * e0 + wy * e1;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 54; ++i) {
    const T v = wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 81) {
    const int row = threadIdx.x % 9, col = threadIdx.x / 9;
    const int r = row < col ? row : col, cc = row < col ? col : row;
    const int k = cc * (cc + 1) / 2 + r;
    Hcc[81 * (size_t)c + threadIdx.x] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
  } else if (threadIdx.x < 90) {
    const int k = 45 + threadIdx.x - 81;
    bc[9 * (size_t)c + threadIdx.x - 81] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
  }
}

// ---------------------------------------------------------------------------
// Column scales from the diagonals (graph.hpp:254-270) + final chi2 reduction.
template <typename T>
__global__ void k_scales(int Nc, int Np, int scale_system, const T *__restrict__ Hcc,
                         const T *__restrict__ Hll, T *__restrict__ scales, int n_partials,
                         const double *__restrict__ chi2_partial, double *__restrict__ chi2_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t pose_dim = 9 * (size_t)Nc, n = pose_dim + 3 * (size_t)Np;
  if (i < n) {
    T diag;
    if (i < pose_dim) diag = Hcc[81 * (i / 9) + 10 * (i % 9)];
    else { const size_t q = i - pose_dim; diag = Hll[9 * (q / 3) + 4 * (q % 3)]; }
    scales[i] = scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)diag))) : T(1);
  }
  if (blockIdx.x == 0 && chi2_out) {
    __shared__ double red[4];
    double s = 0;
    for (int k = threadIdx.x; k < n_partials; k += blockDim.x) s += chi2_partial[k];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) *chi2_out = s;
  }
}

// chi2 only (Graph::compute_error + Graph::chi2 on a trial step): one thread per
// observation in pm order, fully coalesced.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_chi2(int No, const int *__restrict__ cam_pm, const int *__restrict__ pt_pm,
       const T *__restrict__ obs_pm, const T *__restrict__ pts, const T *__restrict__ pack,
       int loss_kind, T loss_delta, double *__restrict__ chi2_partial, T *__restrict__ res_out) {
  __shared__ double red[4];
  double chi2 = 0;
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < No; a += gridDim.x * blockDim.x) {
    const int c = cam_pm[a], l = pt_pm[a];
    const T *pk = pack + PACK * (size_t)c;
    const typename Vec2T<T>::type o = reinterpret_cast<const typename Vec2T<T>::type *>(obs_pm)[a];
    T e0, e1;
    bal_residual(pk, pts[3 * l], pts[3 * l + 1], pts[3 * l + 2], o.x, o.y, e0, e1);
    chi2 += (double)loss_rho(loss_kind, loss_delta, e0 * e0 + e1 * e1);
    if (res_out) { res_out[2 * (size_t)a] = e0; res_out[2 * (size_t)a + 1] = e1; }
  }
  const double tot = block_sum_256(chi2, red);
  if (threadIdx.x == 0) chi2_partial[blockIdx.x] = tot;
}

template <typename T>
__global__ void k_reduce_partials(int n, const double *__restrict__ partial, double *__restrict__ out) {
  __shared__ double red[4];
  double s = 0;
  for (int k = threadIdx.x; k < n; k += blockDim.x) s += partial[k];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) *out = s;
}

// ---------------------------------------------------------------------------
// x <- backup + (delta * scale)    (ops/update.hpp:11-31 with additive update)
template <typename T>
__global__ void k_apply_update(size_t n, T *__restrict__ x, const T *__restrict__ dx,
                               const T *__restrict__ scales) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += dx[i] * scales[i];
}

// out = a .* b
template <typename T>
__global__ void k_mul(size_t n, T *__restrict__ out, const T *__restrict__ a, const T *__restrict__ b) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
}

// compute_rho denominator: sum dx (mu dx + b), b = s .* b^u  (levenberg_marquardt.hpp:34-41)
template <typename T>
__global__ void k_rho_denom(size_t n, const T *__restrict__ dx, const T *__restrict__ bu,
                            const T *__restrict__ scales, double mu, double *__restrict__ out) {
  __shared__ double red[4];
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T x = dx[i];
    s += (double)(x * ((T)mu * x + scales[i] * bu[i]));
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) atomicAdd(out, s);
}

// ---------------------------------------------------------------------------
// Scaled exports for parity checks (reference value layouts).
template <typename T>
__global__ void k_export_blocks(size_t nblocks, int rows, int cols, const T *__restrict__ src,
                                const T *__restrict__ srow, const T *__restrict__ scol,
                                const int *__restrict__ rowmap, const int *__restrict__ colmap,
                                const int *__restrict__ srcmap, T *__restrict__ dst) {
  // dst block q = diag(s[rowmap[q]*rows..]) src[srcmap[q]] diag(s[colmap[q]*cols..])
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t bs = (size_t)rows * cols;
  if (i >= nblocks * bs) return;
  const size_t q = i / bs;
  const int e = (int)(i % bs), r = e % rows, c = e / rows;
  const size_t sb = srcmap ? (size_t)srcmap[q] : q;
  const size_t rb = rowmap ? (size_t)rowmap[q] : q, cb = colmap ? (size_t)colmap[q] : q;
  dst[i] = srow[rb * rows + r] * src[sb * bs + e] * scol[cb * cols + c];
}

// ===========================================================================
// Schur path (PCGSchurSolver)
// ===========================================================================

// Per point: scaled + damped Hll, its inverse (scaled space, kept for
// back-substitution and parity), M' = Dp Hll^-1 Dp and v = M' bl^u.
// (schur.hpp:1067-1114 batched inverse; hessian.hpp:136-176 damping)
template <typename T>
__global__ void k_point_prepare(int Np, int Nc, const T *__restrict__ Hll, const T *__restrict__ bl,
                                const T *__restrict__ scales, double mu, int use_identity,
                                T *__restrict__ Hll_inv, T *__restrict__ Mp, T *__restrict__ vl) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= Np) return;
  const T *s = scales + 9 * (size_t)Nc + 3 * (size_t)l;
  const T s0 = s[0], s1 = s[1], s2 = s[2];
  const T *H = Hll + 9 * (size_t)l;
  double A[9];
  const T sc[3] = {s0, s1, s2};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const T v = sc[r] * H[r + 3 * c] * sc[c]; // scaled block in T, as the reference forms it
      A[r + 3 * c] = (r == c) ? (double)damp_diag(v, mu, use_identity) : (double)v;
    }
  spd_inverse<3>(A);
  T inv[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { inv[i] = (T)A[i]; Hll_inv[9 * (size_t)l + i] = inv[i]; }
  T m[9];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) { m[r + 3 * c] = sc[r] * inv[r + 3 * c] * sc[c]; Mp[9 * (size_t)l + r + 3 * c] = m[r + 3 * c]; }
  const T b0 = bl[3 * l], b1 = bl[3 * l + 1], b2 = bl[3 * l + 2];
#pragma unroll
  for (int r = 0; r < 3; ++r) vl[3 * (size_t)l + r] = m[r] * b0 + m[r + 3] * b1 + m[r + 6] * b2;
}

// S = Dc (Hcc^u + damping - sum_l Hcp_il M'_l Hcp_jl^T) Dc, upper blocks.
// Products are pre-sorted by destination block (symbolic phase), so each S
// block is a segmented sum: 9 threads per block (one per column), block order
// sorted by descending segment length so that a wave's 7 blocks have similar
// trip counts.  No atomics (the reference uses 81 atomicAdd per product,
// ops/schur.hpp:155-188).
template <typename T>
__global__ void __launch_bounds__(TPB)
k_schur_products(int nnzb, const int *__restrict__ blk_order, const int *__restrict__ prod_ptr,
                 const int *__restrict__ prod_a, const int *__restrict__ prod_b,
                 const int *__restrict__ S_rowi, const int *__restrict__ S_coli,
                 const int *__restrict__ pt_pm, const T *__restrict__ Hcp, const T *__restrict__ Mp,
                 const T *__restrict__ Hcc, const T *__restrict__ scales, double mu, int use_identity,
                 T *__restrict__ S) {
  const int gid = blockIdx.x * 28 + (threadIdx.x / 9); // 28 S-blocks per 256-thread block (252 lanes)
  const int c = threadIdx.x % 9;
  if (threadIdx.x >= 252 || gid >= nnzb) return;
  const int blk = blk_order[gid];
  const int i = S_rowi[blk], j = S_coli[blk];
  T acc[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) acc[r] = T(0);
  const int qb = prod_ptr[blk], qe = prod_ptr[blk + 1];
  for (int q = qb; q < qe; ++q) {
    const int a = prod_a[q], b = prod_b[q];
    const T *m = Mp + 9 * (size_t)pt_pm[a];
    const T *hb = Hcp + 27 * (size_t)b + c;
    const T hb0 = hb[0], hb1 = hb[9], hb2 = hb[18];
    const T u0 = m[0] * hb0 + m[3] * hb1 + m[6] * hb2;
    const T u1 = m[1] * hb0 + m[4] * hb1 + m[7] * hb2;
    const T u2 = m[2] * hb0 + m[5] * hb1 + m[8] * hb2;
    const T *ha = Hcp + 27 * (size_t)a;
#pragma unroll
    for (int r = 0; r < 9; ++r) acc[r] += ha[r] * u0 + ha[r + 9] * u1 + ha[r + 18] * u2;
  }
  const T *si = scales + 9 * (size_t)i, *sj = scales + 9 * (size_t)j;
  const T sjc = sj[c];
  T *out = S + 81 * (size_t)blk + 9 * c;
  if (i == j) {
    const T *H = Hcc + 81 * (size_t)i + 9 * c;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      T h = si[r] * H[r] * sjc;
      if (r == c) h = damp_diag(h, mu, use_identity);
      out[r] = h - si[r] * acc[r] * sjc;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 9; ++r) out[r] = -(si[r] * acc[r] * sjc);
  }
}

// b_S = Dc (bc^u - sum_obs Hcp^u v_l)      (schur.hpp:901-920), block per camera
template <typename T>
__global__ void __launch_bounds__(TPB)
k_bschur(const int *__restrict__ cam_ptr, const int *__restrict__ pt_cm, const int *__restrict__ pos_cm,
         const T *__restrict__ Hcp, const T *__restrict__ vl, const T *__restrict__ bc,
         const T *__restrict__ scales, T *__restrict__ b_schur) {
  __shared__ T red[4][9];
  const int c = blockIdx.x;
  T acc[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) acc[r] = T(0);
  for (int j = cam_ptr[c] + threadIdx.x; j < cam_ptr[c + 1]; j += TPB) {
    const T *v = vl + 3 * (size_t)pt_cm[j];
    const T v0 = v[0], v1 = v[1], v2 = v[2];
    const T *h = Hcp + 27 * (size_t)pos_cm[j];
#pragma unroll
    for (int r = 0; r < 9; ++r) acc[r] += h[r] * v0 + h[r + 9] * v1 + h[r + 18] * v2;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const T v = wave_sum(acc[r]);
    if (lane == 0) red[wave][r] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int r = threadIdx.x;
    const T y = red[0][r] + red[1][r] + red[2][r] + red[3][r];
    b_schur[9 * (size_t)c + r] = scales[9 * (size_t)c + r] * (bc[9 * (size_t)c + r] - y);
  }
}

// x_l = Hll^-1 (b_l - Hpl^T x_p)     (schur.hpp:279-302), thread per point
template <typename T>
__global__ void k_backsub(int Np, int Nc, const int *__restrict__ pt_ptr, const int *__restrict__ cam_pm,
                          const T *__restrict__ Hcp, const T *__restrict__ Hll_inv,
                          const T *__restrict__ bl, const T *__restrict__ scales,
                          const T *__restrict__ xp, T *__restrict__ xl) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= Np) return;
  T t0 = 0, t1 = 0, t2 = 0;
  for (int a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) {
    const int c = cam_pm[a];
    const T *h = Hcp + 27 * (size_t)a;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const T xs = scales[9 * (size_t)c + r] * xp[9 * (size_t)c + r];
      t0 += h[r] * xs; t1 += h[r + 9] * xs; t2 += h[r + 18] * xs;
    }
  }
  const T *s = scales + 9 * (size_t)Nc + 3 * (size_t)l;
  const T r0 = s[0] * (bl[3 * l] - t0), r1 = s[1] * (bl[3 * l + 1] - t1), r2 = s[2] * (bl[3 * l + 2] - t2);
  const T *inv = Hll_inv + 9 * (size_t)l;
#pragma unroll
  for (int r = 0; r < 3; ++r) xl[3 * (size_t)l + r] = inv[r] * r0 + inv[r + 3] * r1 + inv[r + 6] * r2;
}

// Inverse of the 9x9 diagonal blocks (block-Jacobi of S, block_jacobi_schur.hpp:114-150;
// or of damped D Hcc^u D for the matrix-free PCG, block_jacobi.hpp:120-172).
// MODE 0: src = S blocks via diag index.  MODE 1: src = Hcc^u, scaled + damped here;
// also writes the clamped scalar diagonal used by the operator damping (pcg.hpp:93-103).
template <typename T, int MODE>
__global__ void __launch_bounds__(64) k_inv9(int Nc, const T *__restrict__ src, const int *__restrict__ diag_blk,
                       const T *__restrict__ scales, double mu, int use_identity,
                       T *__restrict__ Minv, T *__restrict__ diag_clamped) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Nc) return;
  double A[81];
  if (MODE == 0) {
    const T *B = src + 81 * (size_t)diag_blk[c];
#pragma unroll
    for (int i = 0; i < 81; ++i) A[i] = (double)B[i];
  } else {
    const T *B = src + 81 * (size_t)c;
    const T *s = scales + 9 * (size_t)c;
#pragma unroll
    for (int col = 0; col < 9; ++col)
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const T v = s[r] * B[r + 9 * col] * s[col];
        if (r == col) {
          A[r + 9 * col] = (double)damp_diag(v, mu, use_identity);
          diag_clamped[9 * (size_t)c + r] = (T)clampd((double)v, 1.0e-6, 1.0e32);
        } else A[r + 9 * col] = (double)v;
      }
  }
  spd_inverse<9>(A);
#pragma unroll
  for (int i = 0; i < 81; ++i) Minv[81 * (size_t)c + i] = (T)A[i];
}

// 3x3 point blocks of the matrix-free block-Jacobi (block_jacobi.hpp:120-172)
template <typename T>
__global__ void k_inv3_points(int Np, int Nc, const T *__restrict__ Hll, const T *__restrict__ scales,
                              double mu, int use_identity, T *__restrict__ Minv,
                              T *__restrict__ diag_clamped) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= Np) return;
  const T *s = scales + 9 * (size_t)Nc + 3 * (size_t)l;
  const T *H = Hll + 9 * (size_t)l;
  double A[9];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const T v = s[r] * H[r + 3 * c] * s[c];
      if (r == c) {
        A[r + 3 * c] = (double)damp_diag(v, mu, use_identity);
        diag_clamped[9 * (size_t)Nc + 3 * (size_t)l + r] = (T)clampd((double)v, 1.0e-6, 1.0e32);
      } else A[r + 3 * c] = (double)v;
    }
  spd_inverse<3>(A);
#pragma unroll
  for (int i = 0; i < 9; ++i) Minv[9 * (size_t)l + i] = (T)A[i];
}

// ---------------------------------------------------------------------------
// PCG on the explicit Schur complement (solver/pcg_schur.hpp:79-168).
// Scalars never visit the host: every iteration k owns a slot in `sc`:
//   rz[k]   = r.z at the start of iteration k (slot 0 filled by the init kernel)
//   den[k]  = p.Ap of iteration k
//   rz0[k]  = running min of |rz_new| before iteration k (inf at k = 0)
//   done[k] = loop already left before iteration k
// Kernels of iteration k only READ slots <= k written by earlier launches and
// accumulate into slot k / k+1, so there are no intra-launch races.
struct PcgScalars {
  double *rz, *den, *rz0, *rr; // rr used by the matrix-free variant
  int *done, *iters;
};

// zero all slots, rz0[0] = +inf.  `cap` = number of slots per array.
__global__ void k_pcg_scalars_init(PcgScalars sc, int cap) {
  for (int i = threadIdx.x; i < cap; i += blockDim.x) {
    sc.rz[i] = 0.0; sc.den[i] = 0.0; sc.rr[i] = 0.0; sc.done[i] = 0;
    sc.rz0[i] = (i == 0) ? __builtin_inf() : 0.0;
  }
  if (threadIdx.x == 0) sc.iters[0] = 0;
}

__device__ __forceinline__ bool pcg_active(const PcgScalars &sc, int k) {
  return !sc.done[k] && sc.rz[k] != 0.0;
}

// y = S p for one block ROW per wave: 7 groups of 9 lanes stride over the
// row's block list (upper blocks as A, lower as A^T from the stored upper
// block), partial sums combined with wave shuffles.  Also accumulates p.Ap.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_schur_matvec(int Nc, const int *__restrict__ row_ptr, const int *__restrict__ row_blk,
               const int *__restrict__ row_col, const T *__restrict__ S, const T *__restrict__ x,
               T *__restrict__ y, PcgScalars sc, int k) {
  if (k >= 0 && !pcg_active(sc, k)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= Nc) return;
  const int g = lane / 9, r = lane % 9;
  T acc = 0;
  if (g < 7) {
    for (int e = row_ptr[i] + g; e < row_ptr[i + 1]; e += 7) {
      const int blk = row_blk[e], j = row_col[e];
      const bool transposed = blk < 0; // encoded as ~blk
      const T *A = S + 81 * (size_t)(transposed ? ~blk : blk);
      const T *xj = x + 9 * (size_t)j;
      if (!transposed) {
#pragma unroll
        for (int c = 0; c < 9; ++c) acc += A[r + 9 * c] * xj[c];
      } else {
#pragma unroll
        for (int c = 0; c < 9; ++c) acc += A[c + 9 * r] * xj[c];
      }
    }
  }
  T tot = acc;
#pragma unroll
  for (int gg = 1; gg < 7; ++gg) tot += __shfl(acc, gg * 9 + r, 64);
  T dot = 0;
  if (lane < 9) {
    y[9 * (size_t)i + r] = tot;
    dot = tot * x[9 * (size_t)i + r];
  }
  if (k >= 0) {
    dot = wave_sum(dot);
    if (lane == 0) atomicAdd(&sc.den[k], (double)dot);
  }
}

// init: r = b_S, z = Minv r, p = z, rz[0] = r.z      (pcg_schur.hpp:93-105)
template <typename T>
__global__ void __launch_bounds__(TPB)
k_pcgs_init(int Nc, const T *__restrict__ b, const T *__restrict__ Minv, T *__restrict__ r,
            T *__restrict__ z, T *__restrict__ p, PcgScalars sc) {
  __shared__ double red[4];
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double part = 0;
  if (t < 9 * (size_t)Nc) {
    const size_t c = t / 9;
    const int row = (int)(t % 9);
    const T *M = Minv + 81 * c;
    T s = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) s += M[row + 9 * q] * b[9 * c + q];
    r[t] = b[t]; z[t] = s; p[t] = s;
    part = (double)(b[t] * s);
  }
  part = block_sum_256(part, red);
  if (threadIdx.x == 0) atomicAdd(&sc.rz[0], part);
}

// x_backup = x; x += alpha p; r -= alpha Ap; z = Minv r; rz[k+1] += r.z   (:125-142)
template <typename T>
__global__ void __launch_bounds__(TPB)
k_pcgs_update(int Nc, T *__restrict__ x, T *__restrict__ xb, T *__restrict__ r, T *__restrict__ z,
              const T *__restrict__ p, const T *__restrict__ Ap, const T *__restrict__ Minv,
              PcgScalars sc, int k) {
  if (!pcg_active(sc, k)) return;
  const double den = sc.den[k];
  if (den == 0.0 || den != den) return;
  __shared__ double red[4];
  __shared__ T rs[TPB];
  const T alpha = (T)(sc.rz[k] / den);
  const size_t base = (size_t)blockIdx.x * 252; // 28 cameras per block
  const size_t t = base + threadIdx.x;
  const bool on = threadIdx.x < 252 && t < 9 * (size_t)Nc;
  T rn = 0;
  if (on) {
    const T xo = x[t];
    xb[t] = xo;
    x[t] = alpha * p[t] + xo;
    rn = -alpha * Ap[t] + r[t];
    r[t] = rn;
  }
  rs[threadIdx.x] = rn;
  __syncthreads();
  double part = 0;
  if (on) {
    const size_t c = t / 9;
    const int row = (int)(t % 9);
    const T *M = Minv + 81 * c;
    const T *rc = rs + (threadIdx.x / 9) * 9;
    T s = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) s += M[row + 9 * q] * rc[q];
    z[t] = s;
    part = (double)(rn * s);
  }
  part = block_sum_256(part, red);
  if (threadIdx.x == 0) atomicAdd(&sc.rz[k + 1], part);
}

// rejection / beta / p update / tolerance (:143-163); thread 0 publishes slot k+1
template <typename T>
__global__ void k_pcgs_direction(int Nc, T *__restrict__ x, const T *__restrict__ xb, T *__restrict__ p,
                                 const T *__restrict__ z, PcgScalars sc, int k, double tol,
                                 double rejection_ratio) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool first = (t == 0);
  const double rz = sc.rz[k], den = sc.den[k], rz0 = sc.rz0[k];
  bool active = !sc.done[k] && rz != 0.0;
  bool stepped = active && !(den == 0.0 || den != den);
  if (!stepped) {
    if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; }
    return;
  }
  // T-precision scalars, as the reference keeps them in T on the host
  const T rz_new = (T)sc.rz[k + 1];
  const bool reject = (fabs((double)rz_new) > rejection_ratio * rz0) || (rz_new != rz_new);
  if (reject) {
    if (t < 9 * (size_t)Nc) x[t] = xb[t];
    if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; sc.iters[0] = k + 1; }
    return;
  }
  const T beta = rz_new / (T)rz;
  if (t < 9 * (size_t)Nc) p[t] = beta * p[t] + z[t];
  if (first) {
    sc.rz0[k + 1] = fmin(rz0, fabs((double)rz_new));
    sc.done[k + 1] = (fabs((double)rz_new) < tol) ? 1 : 0;
    sc.iters[0] = k + 1;
  }
}

// ===========================================================================
// Matrix-free PCG (PCGSolver, solver/pcg.hpp:61-232) — operator kernels
// ===========================================================================
// v2 = D J^T rho' J D p + mu diag .* p, with J RECOMPUTED from the camera pack
// (the reference streams the stored J twice per iteration, pcg.hpp:143-163).
// ps = s .* p is prepared by the direction kernel.

// point rows: thread per point
template <typename T>
__global__ void __launch_bounds__(TPB)
k_op_points(int Np, int Nc, const int *__restrict__ pt_ptr, const int *__restrict__ cam_pm,
            const T *__restrict__ obs_pm, const T *__restrict__ pts, const T *__restrict__ pack,
            int loss_kind, T loss_delta, const T *__restrict__ scales, const T *__restrict__ ps,
            const T *__restrict__ p, const T *__restrict__ diag, double mu, int use_identity,
            T *__restrict__ v2, PcgScalars sc, int k) {
  if (!pcg_active(sc, k)) return;
  __shared__ double red[4];
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  double part = 0;
  if (l < Np) {
    const size_t off = 9 * (size_t)Nc + 3 * (size_t)l;
    const T X = pts[3 * l], Y = pts[3 * l + 1], Z = pts[3 * l + 2];
    const T q0 = ps[off], q1 = ps[off + 1], q2 = ps[off + 2];
    T a0 = 0, a1 = 0, a2 = 0;
    for (int a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) {
      const int c = cam_pm[a];
      T pk[PACK];
#pragma unroll
      for (int i = 0; i < PACK; ++i) pk[i] = pack[PACK * (size_t)c + i];
      const typename Vec2T<T>::type o = reinterpret_cast<const typename Vec2T<T>::type *>(obs_pm)[a];
      T e0, e1, Jc[18], Jp[6];
      bal_linearize(pk, X, Y, Z, o.x, o.y, e0, e1, Jc, Jp);
      const T w = loss_drho(loss_kind, loss_delta, e0 * e0 + e1 * e1);
      const T *pc = ps + 9 * (size_t)c;
      T u0 = Jp[0] * q0 + Jp[2] * q1 + Jp[4] * q2;
      T u1 = Jp[1] * q0 + Jp[3] * q1 + Jp[5] * q2;
#pragma unroll
      for (int i = 0; i < 9; ++i) { const T pv = pc[i]; u0 += Jc[2 * i] * pv; u1 += Jc[2 * i + 1] * pv; }
      u0 *= w; u1 *= w;
      a0 += Jp[0] * u0 + Jp[1] * u1;
      a1 += Jp[2] * u0 + Jp[3] * u1;
      a2 += Jp[4] * u0 + Jp[5] * u1;
    }
    const T acc[3] = {a0, a1, a2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const T pv = p[off + i];
      const T damp = use_identity ? (T)mu * pv : (T)mu * diag[off + i] * pv;
      const T out = scales[off + i] * acc[i] + damp;
      v2[off + i] = out;
      part += (double)(out * pv);
    }
  }
  part = block_sum_256(part, red);
  if (threadIdx.x == 0) atomicAdd(&sc.den[k], part);
}

// camera rows: block per camera
template <typename T>
__global__ void __launch_bounds__(TPB)
k_op_cameras(int Nc, const int *__restrict__ cam_ptr, const int *__restrict__ pt_cm,
             const T *__restrict__ obs_cm, const T *__restrict__ pts, const T *__restrict__ pack,
             int loss_kind, T loss_delta, const T *__restrict__ scales, const T *__restrict__ ps,
             const T *__restrict__ p, const T *__restrict__ diag, double mu, int use_identity,
             T *__restrict__ v2, PcgScalars sc, int k) {
  if (!pcg_active(sc, k)) return;
  __shared__ T red[4][9];
  const int c = blockIdx.x;
  T pk[PACK], pc[9];
#pragma unroll
  for (int i = 0; i < PACK; ++i) pk[i] = pack[PACK * (size_t)c + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) pc[i] = ps[9 * (size_t)c + i];
  T acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = T(0);
  for (int j = cam_ptr[c] + threadIdx.x; j < cam_ptr[c + 1]; j += TPB) {
    const int l = pt_cm[j];
    const T X = pts[3 * l], Y = pts[3 * l + 1], Z = pts[3 * l + 2];
    const size_t off = 9 * (size_t)Nc + 3 * (size_t)l;
    const T q0 = ps[off], q1 = ps[off + 1], q2 = ps[off + 2];
    const typename Vec2T<T>::type o = reinterpret_cast<const typename Vec2T<T>::type *>(obs_cm)[j];
    T e0, e1, Jc[18], Jp[6];
    bal_linearize(pk, X, Y, Z, o.x, o.y, e0, e1, Jc, Jp);
    const T w = loss_drho(loss_kind, loss_delta, e0 * e0 + e1 * e1);
    T u0 = Jp[0] * q0 + Jp[2] * q1 + Jp[4] * q2;
    T u1 = Jp[1] * q0 + Jp[3] * q1 + Jp[5] * q2;
#pragma unroll
    for (int i = 0; i < 9; ++i) { u0 += Jc[2 * i] * pc[i]; u1 += Jc[2 * i + 1] * pc[i]; }
    u0 *= w; u1 *= w;
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] += Jc[2 * i] * u0 + Jc[2 * i + 1] * u1;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const T v = wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int i = threadIdx.x;
    const size_t off = 9 * (size_t)c + i;
    const T pv = p[off];
    const T damp = use_identity ? (T)mu * pv : (T)mu * diag[off] * pv;
    const T out = scales[off] * (red[0][i] + red[1][i] + red[2][i] + red[3][i]) + damp;
    v2[off] = out;
    // 9 lanes of wave 0: reduce with shuffles, one atomic per camera
    T d = out * pv;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { const T other = __shfl_down(d, o, 64); if (i + o < 9) d += other; }
    if (i == 0) atomicAdd(&sc.den[k], (double)d);
  }
}

// z' = Minv r for the full system (9x9 camera blocks then 3x3 point blocks);
// mode 0 (init): r = s.*b^u first.  mode 1: x_backup = x; x += alpha p; r -= alpha v2 first.
// Accumulates rr[slot] = r.r and rz[slot] = r.z'  (the reference applies the
// preconditioner to r/||r||, pcg.hpp:108-118,171-183; Minv is linear so
// z = z'/||r|| and r.z = (r.z')/||r|| are formed from these two sums).
template <typename T, int MODE, bool IDENTITY>
__global__ void __launch_bounds__(TPB)
k_pcg_update(int Nc, int Np, const T *__restrict__ bu, const T *__restrict__ scales, T *__restrict__ x,
             T *__restrict__ xb, T *__restrict__ r, T *__restrict__ zt, const T *__restrict__ p,
             const T *__restrict__ v2, const T *__restrict__ MinvC, const T *__restrict__ MinvP,
             PcgScalars sc, int k) {
  T alpha = 0;
  if (MODE == 1) {
    if (!pcg_active(sc, k)) return;
    // reference rz = r.z with z = Minv (r/||r||)  ->  (r.z')/||r||
    const T rz = (T)sc.rz[k] * (T)(1.0 / (double)(T)sqrt((double)(T)sc.rr[k]));
    alpha = rz / (T)sc.den[k];
  }
  __shared__ double red[4];
  __shared__ T rs[TPB];
  const size_t pose_dim = 9 * (size_t)Nc, n = pose_dim + 3 * (size_t)Np;
  // 252 scalars per block: 28 cameras or 84 points; camera blocks first
  const size_t cam_blocks = (pose_dim + 251) / 252;
  size_t t;
  bool is_cam;
  if (blockIdx.x < cam_blocks) { t = (size_t)blockIdx.x * 252 + threadIdx.x; is_cam = true; }
  else { t = pose_dim + ((size_t)blockIdx.x - cam_blocks) * 252 + threadIdx.x; is_cam = false; }
  const bool on = threadIdx.x < 252 && (is_cam ? t < pose_dim : t < n);
  T rn = 0;
  if (on) {
    if (MODE == 0) { rn = scales[t] * bu[t]; x[t] = T(0); }
    else {
      const T xo = x[t];
      xb[t] = xo;
      x[t] = alpha * p[t] + xo;
      rn = -alpha * v2[t] + r[t];
    }
    r[t] = rn;
  }
  rs[threadIdx.x] = rn;
  __syncthreads();
  double prr = 0, prz = 0;
  if (on) {
    T s = 0;
    if (IDENTITY) s = rn;
    else if (is_cam) {
      const T *M = MinvC + 81 * (t / 9);
      const int row = (int)(t % 9);
      const T *rc = rs + (threadIdx.x / 9) * 9;
#pragma unroll
      for (int q = 0; q < 9; ++q) s += M[row + 9 * q] * rc[q];
    } else {
      const size_t q0 = t - pose_dim;
      const T *M = MinvP + 9 * (q0 / 3);
      const int row = (int)(q0 % 3);
      const T *rc = rs + (threadIdx.x / 3) * 3;
      s = M[row] * rc[0] + M[row + 3] * rc[1] + M[row + 6] * rc[2];
    }
    zt[t] = s;
    prr = (double)(rn * rn);
    prz = (double)(rn * s);
  }
  const int slot = (MODE == 0) ? 0 : k + 1;
  prr = block_sum_256(prr, red);
  if (threadIdx.x == 0) atomicAdd(&sc.rr[slot], prr);
  prz = block_sum_256(prz, red);
  if (threadIdx.x == 0) atomicAdd(&sc.rz[slot], prz);
}

// Direction kernel of the matrix-free PCG (pcg.hpp:108-127 for k = -1, :184-217 otherwise).
// Here sc.rz[k] holds r.z' and sc.rr[k] holds r.r; the reference's rz is rz'/sqrt(rr).
template <typename T>
__global__ void k_pcg_direction(size_t n, T *__restrict__ x, const T *__restrict__ xb, T *__restrict__ p,
                                T *__restrict__ ps, const T *__restrict__ zt, const T *__restrict__ scales,
                                PcgScalars sc, int k, double tol, double rejection_ratio) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool first = (t == 0);
  if (k < 0) { // p = z = z'/||r||
    const T scale = (T)(1.0 / (double)(T)sqrt((double)(T)sc.rr[0]));
    if (t < n) { const T v = scale * zt[t]; p[t] = v; ps[t] = scales[t] * v; }
    return;
  }
  const bool active = !sc.done[k] && sc.rz[k] != 0.0;
  if (!active) {
    if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = sc.rz0[k]; }
    return;
  }
  const T scale_old = (T)(1.0 / (double)(T)sqrt((double)(T)sc.rr[k]));
  const T scale_new = (T)(1.0 / (double)(T)sqrt((double)(T)sc.rr[k + 1]));
  const T rz = (T)sc.rz[k] * scale_old;
  const T rz_new = (T)sc.rz[k + 1] * scale_new;
  const double rz0 = sc.rz0[k];
  const bool reject = (fabs((double)rz_new) > rejection_ratio * rz0) || (rz_new != rz_new);
  if (reject) {
    if (t < n) x[t] = xb[t];
    if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; sc.iters[0] = k + 1; }
    return;
  }
  const T beta = rz_new / rz;
  if (t < n) { const T v = beta * p[t] + scale_new * zt[t]; p[t] = v; ps[t] = scales[t] * v; }
  if (first) {
    sc.rz0[k + 1] = fmin(rz0, fabs((double)rz_new));
    sc.done[k + 1] = (fabs((double)rz_new) < tol) ? 1 : 0;
    sc.iters[0] = k + 1;
  }
}

} // namespace gr
