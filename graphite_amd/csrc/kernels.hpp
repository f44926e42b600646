// HIP kernels of the BAL hot path, written for gfx950 (wave64, 256 CUs).
//
// Data layout in HBM (all arrays resident for the whole optimisation):
//   cams[Nc][9], pts[Np][3]                 vertex values (T)
//   pack[Nc][24]                            per-camera R,t,f,k1,k2,Jr (bal_device.hpp)
//   point-major ("pm") observation order: observations sorted by point, then camera
//     obs_pm[No][2], cam_pm[No], pt_pm[No], pt_ptr[Np+1]
//   camera-major ("cm") order: the same observations sorted by camera, then point
//     obs_cm[No][2], pt_cm[No], pos_cm[No] (-> pm position), cam_ptr[Nc+1]
//     cut into CHUNKs of <= 128 consecutive observations of ONE camera:
//     chunk_cam[nch], chunk_beg[nch+1], cam_chunk_ptr[Nc+1]
//   UNSCALED blocks of J^T rho' J and -J^T rho' r:
//     Hcc[Nc][81], bu = [bc[Nc][9]; bl[Np][3]], Hll[Np][9], Hcp[No][27] (pm order, 9x3 col-major)
//   scales[n]  column scales s = 1/(eps + sqrt(diag))           (graph.hpp:262-270)
//
// The reference scales the stored Jacobians in place and multiplies afterwards
// (graph.hpp:254-287); here every kernel works on UNSCALED blocks and the
// diagonal congruence H = D H^u D, b = D b^u is folded into the small
// per-vertex matrices where it is needed.
//
// Parallelisation: ONE THREAD PER OBSERVATION everywhere (No is 4-7x the number
// of points, and the chip wants >= 0.5 M threads).  Point-side sums are wave-level
// segmented scans over the pm order (a point's observations are consecutive);
// camera-side sums are wave reductions over cm-order chunks followed by a
// fixed-order sum of the chunk partials.  The only float atomics left are the two
// halves of a point whose observations straddle a wave boundary (commutative, so
// still deterministic) and the slotted dot-product accumulators.
#pragma once
#include "bal_device.hpp"
#include "common.hpp"
#include <cfloat>

namespace gr {

constexpr int TPB = 256;
constexpr int CHUNK = 128; // observations of one camera handled by one wave (2 per lane)

template <typename T> struct Vec2T;
template <> struct Vec2T<float> { using type = float2; };
template <> struct Vec2T<double> { using type = double2; };

template <typename T> __device__ __forceinline__ void load_pack(const T *__restrict__ pack, int c, T *pk) {
  const T *src = pack + PACK * (size_t)c;
#pragma unroll
  for (int i = 0; i < PACK; ++i) pk[i] = src[i];
}

// ---------------------------------------------------------------------------
// camera pack (+ optional x += dx .* s on the cameras first): Nc threads
template <typename T>
__global__ void k_campack(int Nc, T *__restrict__ cams, T *__restrict__ pack, const T *__restrict__ dx,
                          const T *__restrict__ scales) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Nc) return;
  T cam[9], pk[PACK];
#pragma unroll
  for (int i = 0; i < 9; ++i) cam[i] = cams[9 * c + i];
  if (dx) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { cam[i] += dx[9 * c + i] * scales[9 * c + i]; cams[9 * c + i] = cam[i]; }
  }
  make_campack(cam, pk);
#pragma unroll
  for (int i = 0; i < PACK; ++i) pack[PACK * (size_t)c + i] = pk[i];
}

// ---------------------------------------------------------------------------
// Graph::linearize + Hessian::update_values (A4-A10 of SURVEY §8a): ONE pass over the
// observations in CAMERA-major order, one wave per chunk (<= 128 observations of one
// camera).  The camera pack is wave-uniform (scalar loads); the only divergent
// accesses are the point gather and the per-observation scatter below — in
// point-major order every lane would need its own 192-byte pack, and the 24 x 64
// cache-line requests per wave were measured to dominate the kernel.
//   camera side : 45 + 9 sums reduced across the wave (transpose-sum) -> cam_partial[ch][54]
//   point side  : per-observation [w Jp^T Jp (6), -w Jp^T e (3)] -> g9[pm position][9]
//                 (summed per point by k_linearize_finalize; consecutive in pm order)
//   Hcp^u       : per-observation 9x3 block -> Hcp[pm position][27]  (Schur solvers only)
//   chi2        : block partial
template <typename T, bool WRITE_HCP>
__global__ void __launch_bounds__(TPB)
k_linearize(int nch, const int *__restrict__ chunk_cam, const int *__restrict__ chunk_beg,
            const int *__restrict__ pt_cm, const int *__restrict__ pos_cm, const T *__restrict__ obs_cm,
            const T *__restrict__ pts, const T *__restrict__ pack, int loss_kind, T loss_delta,
            T *__restrict__ g9, T *__restrict__ Hcp, T *__restrict__ cam_partial,
            double *__restrict__ chi2_partial) {
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  using V2 = typename Vec2T<T>::type;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  double chi2 = 0.0;
  if (ch < nch) {
    const int c = __builtin_amdgcn_readfirstlane(chunk_cam[ch]);
    T pk[PACK];
    load_pack(pack, c, pk);
    T acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = T(0);
    const int beg = __builtin_amdgcn_readfirstlane(chunk_beg[ch]), end = __builtin_amdgcn_readfirstlane(chunk_beg[ch + 1]);
    for (int j = beg + lane; j < end; j += 64) {
      const int l = pt_cm[j];
      const size_t a = (size_t)pos_cm[j];
      const V2 o = reinterpret_cast<const V2 *>(obs_cm)[j];
      T e0, e1, Jc[18], Jp[6];
      bal_linearize(pk, pts[3 * (size_t)l], pts[3 * (size_t)l + 1], pts[3 * (size_t)l + 2], o.x, o.y, e0, e1, Jc, Jp);
      const T raw = e0 * e0 + e1 * e1;
      const T w = loss_drho(loss_kind, loss_delta, raw);
      chi2 += (double)loss_rho(loss_kind, loss_delta, raw);
      int k = 0;
#pragma unroll
      for (int col = 0; col < 9; ++col) {
        const T wx = w * Jc[2 * col], wy = w * Jc[2 * col + 1];
#pragma unroll
        for (int row = 0; row <= col; ++row) acc[k++] += Jc[2 * row] * wx + Jc[2 * row + 1] * wy;
        acc[45 + col] -= wx * e0 + wy * e1;
      }
      const T wp0x = w * Jp[0], wp0y = w * Jp[1], wp1x = w * Jp[2], wp1y = w * Jp[3], wp2x = w * Jp[4], wp2y = w * Jp[5];
      T *g = g9 + 9 * a;
      g[0] = wp0x * Jp[0] + wp0y * Jp[1];
      g[1] = wp0x * Jp[2] + wp0y * Jp[3];
      g[2] = wp0x * Jp[4] + wp0y * Jp[5];
      g[3] = wp1x * Jp[2] + wp1y * Jp[3];
      g[4] = wp1x * Jp[4] + wp1y * Jp[5];
      g[5] = wp2x * Jp[4] + wp2y * Jp[5];
      g[6] = -(wp0x * e0 + wp0y * e1);
      g[7] = -(wp1x * e0 + wp1y * e1);
      g[8] = -(wp2x * e0 + wp2y * e1);
      if (WRITE_HCP) {
        T *h = Hcp + 27 * a;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          h[r] = Jc[2 * r] * wp0x + Jc[2 * r + 1] * wp0y;
          h[r + 9] = Jc[2 * r] * wp1x + Jc[2 * r + 1] * wp1y;
          h[r + 18] = Jc[2 * r] * wp2x + Jc[2 * r + 1] * wp2y;
        }
      }
    }
    const T tot = wave_transpose_sum<T, 64>(acc, lane);
    if (lane < 54) cam_partial[54 * (size_t)ch + lane] = tot;
  }
  chi2 = block_sum_256(chi2, red);
  if (threadIdx.x == 0) chi2_partial[blockIdx.x] = chi2;
}

// Finalise a linearisation:
//   threads [0, 90 Nc)        fixed-order sum of the chunk partials -> Hcc^u, bc^u, camera scales
//   threads [90 Nc, +Np)      one per point: sum of its observations' g9 -> Hll^u, bl^u, point scales
//   block 0                   chi2 total
// (column scales: graph.hpp:254-270)
template <typename T>
__global__ void __launch_bounds__(TPB)
k_linearize_finalize(int Nc, int Np, int scale_system, const int *__restrict__ cam_chunk_ptr,
                     const T *__restrict__ cam_partial, const int *__restrict__ pt_ptr,
                     const T *__restrict__ g9, T *__restrict__ Hcc, T *__restrict__ bc, T *__restrict__ Hll,
                     T *__restrict__ bl, T *__restrict__ scales, int n_partials,
                     const double *__restrict__ chi2_partial, double *__restrict__ chi2_out) {
  const unsigned t = blockIdx.x * TPB + threadIdx.x;
  const unsigned ncam = 90u * (unsigned)Nc;
  if (t < ncam) {
    const unsigned c = t / 90u, e = t % 90u;
    int idx;
    unsigned row = 0, col = 0;
    if (e < 81u) {
      row = e % 9u; col = e / 9u;
      const unsigned r = row < col ? row : col, cc = row < col ? col : row;
      idx = (int)(cc * (cc + 1) / 2 + r);
    } else idx = 45 + (int)(e - 81u);
    T s = 0;
    for (int ch = cam_chunk_ptr[c]; ch < cam_chunk_ptr[c + 1]; ++ch) s += cam_partial[54 * (size_t)ch + idx];
    if (e < 81u) {
      Hcc[81 * (size_t)c + e] = s;
      if (row == col) scales[9 * c + row] = scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)s))) : T(1);
    } else bc[9 * c + (e - 81u)] = s;
  } else if (t < ncam + (unsigned)Np) {
    const unsigned l = t - ncam;
    T v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = T(0);
    for (int a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) {
      const T *g = g9 + 9 * (size_t)a;
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] += g[i];
    }
    T *H = Hll + 9 * (size_t)l;
    H[0] = v[0]; H[1] = v[1]; H[2] = v[2]; H[3] = v[1]; H[4] = v[3]; H[5] = v[4]; H[6] = v[2]; H[7] = v[4]; H[8] = v[5];
    bl[3 * (size_t)l] = v[6]; bl[3 * (size_t)l + 1] = v[7]; bl[3 * (size_t)l + 2] = v[8];
    T *s = scales + 9 * (size_t)Nc + 3 * (size_t)l;
    s[0] = scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)v[0]))) : T(1);
    s[1] = scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)v[3]))) : T(1);
    s[2] = scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)v[5]))) : T(1);
  }
  if (blockIdx.x == 0 && chi2_out) {
    __shared__ double red[4];
    double s = 0;
    for (int k = threadIdx.x; k < n_partials; k += TPB) s += chi2_partial[k];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) *chi2_out = s;
  }
}

// chi2 of a trial step (Graph::compute_error + Graph::chi2) fused with compute_rho's
// denominator sum dx (mu dx + b) (levenberg_marquardt.hpp:34-41).  One thread per
// observation in CAMERA-major order (neighbouring lanes share the camera pack);
// dscal[0] = chi2, dscal[1] = rho denominator are produced by the last block to
// finish (ticket), so no extra reduce launch.  res_out (optional) is indexed by pm position.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_chi2(int No, unsigned n, const int *__restrict__ cam_cm, const int *__restrict__ pt_cm,
       const int *__restrict__ pos_cm, const T *__restrict__ obs_cm, const T *__restrict__ pts,
       const T *__restrict__ pack, int loss_kind, T loss_delta, const T *__restrict__ dx,
       const T *__restrict__ bu, const T *__restrict__ scales, double mu, double *__restrict__ partial,
       unsigned *__restrict__ ticket, double *__restrict__ dscal, T *__restrict__ res_out) {
  __shared__ double red[4];
  __shared__ bool last;
  using V2 = typename Vec2T<T>::type;
  double chi2 = 0, rho = 0;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < No; j += gridDim.x * TPB) {
    const int c = cam_cm[j], l = pt_cm[j];
    const T *pk = pack + PACK * (size_t)c;
    const V2 o = reinterpret_cast<const V2 *>(obs_cm)[j];
    T e0, e1;
    bal_residual(pk, pts[3 * (size_t)l], pts[3 * (size_t)l + 1], pts[3 * (size_t)l + 2], o.x, o.y, e0, e1);
    chi2 += (double)loss_rho(loss_kind, loss_delta, e0 * e0 + e1 * e1);
    if (res_out) { const size_t a = (size_t)pos_cm[j]; res_out[2 * a] = e0; res_out[2 * a + 1] = e1; }
  }
  if (dx) {
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
      const T x = dx[i];
      rho += (double)(x * ((T)mu * x + scales[i] * bu[i]));
    }
  }
  chi2 = block_sum_256(chi2, red);
  rho = block_sum_256(rho, red);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partial[2 * blockIdx.x], chi2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partial[2 * blockIdx.x + 1], rho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = atomicAdd(ticket, 1u);
    last = (tk == gridDim.x - 1);
  }
  __syncthreads();
  if (last) {
    double s0 = 0, s1 = 0;
    for (unsigned k = threadIdx.x; k < gridDim.x; k += TPB) {
      s0 += __hip_atomic_load(&partial[2 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s1 += __hip_atomic_load(&partial[2 * k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s0 = block_sum_256(s0, red);
    s1 = block_sum_256(s1, red);
    if (threadIdx.x == 0) { dscal[0] = s0; dscal[1] = s1; *ticket = 0u; }
  }
}

// ---------------------------------------------------------------------------
// x += dx .* s     (ops/update.hpp:11-31 with additive update)
template <typename T>
__global__ void k_apply_update(unsigned n, T *__restrict__ x, const T *__restrict__ dx,
                               const T *__restrict__ scales) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += dx[i] * scales[i];
}

// out = a .* b
template <typename T>
__global__ void k_mul(unsigned n, T *__restrict__ out, const T *__restrict__ a, const T *__restrict__ b) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
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
  const T sc[3] = {s[0], s[1], s[2]};
  const T *H = Hll + 9 * (size_t)l;
  double A[9];
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
  const T b0 = bl[3 * (size_t)l], b1 = bl[3 * (size_t)l + 1], b2 = bl[3 * (size_t)l + 2];
#pragma unroll
  for (int r = 0; r < 3; ++r) vl[3 * (size_t)l + r] = m[r] * b0 + m[r + 3] * b1 + m[r + 6] * b2;
}

// S^u accumulation: sum_l Hcp_il M'_l Hcp_jl^T for every upper block (i,j).
// Products are pre-sorted by destination block (symbolic phase) and cut into
// work items of <= 56 products of ONE block; one wave per work item: 7 groups
// of 9 lanes (lane = output column) take 8 products each, group partials are
// combined with shuffles.  Single-item blocks are finished in place (scaling,
// Hcc, damping); multi-item blocks (diagonal blocks, popular camera pairs)
// accumulate raw sums with atomics into the zeroed S and are finished by
// k_schur_fixup.  (The reference issues 81 atomicAdd per product,
// ops/schur.hpp:155-188.)
template <typename T>
__device__ __forceinline__ void schur_epilogue(int i, int j, int c, const T *acc, const T *__restrict__ Hcc,
                                               const T *__restrict__ scales, double mu, int use_identity,
                                               T *__restrict__ out) {
  const T *si = scales + 9 * (size_t)i;
  const T sjc = scales[9 * (size_t)j + c];
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

template <typename T>
__global__ void __launch_bounds__(TPB)
k_schur_products(int nitems, const int *__restrict__ item_blk, const int *__restrict__ item_beg,
                 const int *__restrict__ item_end, const int *__restrict__ item_single,
                 const int *__restrict__ prod_a, const int *__restrict__ prod_b,
                 const int *__restrict__ S_rowi, const int *__restrict__ S_coli,
                 const int *__restrict__ pt_pm, const T *__restrict__ Hcp, const T *__restrict__ Mp,
                 const T *__restrict__ Hcc, const T *__restrict__ scales, double mu, int use_identity,
                 T *__restrict__ S) {
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= nitems) return;
  const int lane = threadIdx.x & 63;
  const int g = lane / 9, c = lane % 9;
  const int blk = item_blk[item];
  T acc[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) acc[r] = T(0);
  if (g < 7) {
    for (int q = item_beg[item] + g; q < item_end[item]; q += 7) {
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
  }
  // combine the 7 groups: lanes 0..8 collect
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    T tot = acc[r];
#pragma unroll
    for (int gg = 1; gg < 7; ++gg) tot += __shfl(acc[r], gg * 9 + c, 64);
    acc[r] = tot;
  }
  if (lane < 9) {
    T *out = S + 81 * (size_t)blk + 9 * c;
    if (item_single[item]) schur_epilogue<T>(S_rowi[blk], S_coli[blk], c, acc, Hcc, scales, mu, use_identity, out);
    else {
#pragma unroll
      for (int r = 0; r < 9; ++r) atomicAdd(&out[r], acc[r]);
    }
  }
}

// zero the multi-item blocks (mode 0) / finish them (mode 1): thread per (block, column)
template <typename T, int MODE>
__global__ void k_schur_multi(int nmulti, const int *__restrict__ multi_blk, const int *__restrict__ S_rowi,
                              const int *__restrict__ S_coli, const T *__restrict__ Hcc,
                              const T *__restrict__ scales, double mu, int use_identity, T *__restrict__ S) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nmulti * 9) return;
  const int blk = multi_blk[t / 9], c = t % 9;
  T *out = S + 81 * (size_t)blk + 9 * c;
  if (MODE == 0) {
#pragma unroll
    for (int r = 0; r < 9; ++r) out[r] = T(0);
  } else {
    T acc[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) acc[r] = out[r];
    schur_epilogue<T>(S_rowi[blk], S_coli[blk], c, acc, Hcc, scales, mu, use_identity, out);
  }
}

// b_S = Dc (bc^u - sum_obs Hcp^u v_l)      (schur.hpp:901-920): wave per camera chunk
// writes 9 partials, k_bschur_finalize sums them in fixed order.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_bschur_partial(int nch, const int *__restrict__ chunk_beg, const int *__restrict__ pt_cm,
                 const int *__restrict__ pos_cm, const T *__restrict__ Hcp, const T *__restrict__ vl,
                 T *__restrict__ partial9) {
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= nch) return;
  const int lane = threadIdx.x & 63;
  T acc[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) acc[r] = T(0);
  for (int j = chunk_beg[ch] + lane; j < chunk_beg[ch + 1]; j += 64) {
    const T *v = vl + 3 * (size_t)pt_cm[j];
    const T v0 = v[0], v1 = v[1], v2 = v[2];
    const T *h = Hcp + 27 * (size_t)pos_cm[j];
#pragma unroll
    for (int r = 0; r < 9; ++r) acc[r] += h[r] * v0 + h[r + 9] * v1 + h[r + 18] * v2;
  }
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const T s = wave_sum(acc[r]);
    if (lane == 0) partial9[9 * (size_t)ch + r] = s;
  }
}
template <typename T>
__global__ void k_bschur_finalize(int Nc, const int *__restrict__ cam_chunk_ptr, const T *__restrict__ partial9,
                                  const T *__restrict__ bc, const T *__restrict__ scales, T *__restrict__ b_schur) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 9u * (unsigned)Nc) return;
  const unsigned c = t / 9u, r = t % 9u;
  T y = 0;
  for (int ch = cam_chunk_ptr[c]; ch < cam_chunk_ptr[c + 1]; ++ch) y += partial9[9 * (size_t)ch + r];
  b_schur[t] = scales[t] * (bc[t] - y);
}

// x_l = Hll^-1 (b_l - Hpl^T x_p)     (schur.hpp:279-302): thread per observation (pm),
// segmented scan per point, the tail lane finishes the point.  Points that straddle
// a wave boundary go through `xl_acc` (zero on entry, re-zeroed by the fixup).
template <typename T>
__global__ void __launch_bounds__(TPB)
k_backsub(int No, int Nc, const int *__restrict__ pt_ptr, const int *__restrict__ cam_pm,
          const int *__restrict__ pt_pm, const T *__restrict__ Hcp, const T *__restrict__ Hll_inv,
          const T *__restrict__ bl, const T *__restrict__ scales, const T *__restrict__ xp,
          T *__restrict__ xl, T *__restrict__ xl_acc, int *__restrict__ boundary_flag) {
  const int a = blockIdx.x * TPB + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = a < No;
  T v[3] = {T(0), T(0), T(0)};
  int l = -1;
  if (valid) {
    const int c = cam_pm[a];
    l = pt_pm[a];
    const T *h = Hcp + 27 * (size_t)a;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const T xs = scales[9 * (size_t)c + r] * xp[9 * (size_t)c + r];
      v[0] += h[r] * xs; v[1] += h[r + 9] * xs; v[2] += h[r + 18] * xs;
    }
  }
  seg_scan<T, 3>(v, l, lane);
  const int ln = __shfl_down(l, 1, 64);
  if (valid && (lane == 63 || ln != l)) {
    const int wbase = a - lane;
    const bool complete = pt_ptr[l] >= wbase && pt_ptr[l + 1] <= wbase + 64;
    if (complete) {
      const T *s = scales + 9 * (size_t)Nc + 3 * (size_t)l;
      const T r0 = s[0] * (bl[3 * (size_t)l] - v[0]), r1 = s[1] * (bl[3 * (size_t)l + 1] - v[1]), r2 = s[2] * (bl[3 * (size_t)l + 2] - v[2]);
      const T *inv = Hll_inv + 9 * (size_t)l;
#pragma unroll
      for (int r = 0; r < 3; ++r) xl[3 * (size_t)l + r] = inv[r] * r0 + inv[r + 3] * r1 + inv[r + 6] * r2;
    } else {
      atomicAdd(&xl_acc[3 * (size_t)l], v[0]); atomicAdd(&xl_acc[3 * (size_t)l + 1], v[1]); atomicAdd(&xl_acc[3 * (size_t)l + 2], v[2]);
      boundary_flag[l] = 1;
    }
  }
}
template <typename T>
__global__ void k_backsub_fixup(int Np, int Nc, const T *__restrict__ Hll_inv, const T *__restrict__ bl,
                                const T *__restrict__ scales, T *__restrict__ xl, T *__restrict__ xl_acc,
                                int *__restrict__ boundary_flag) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= Np || !boundary_flag[l]) return;
  boundary_flag[l] = 0;
  const T *s = scales + 9 * (size_t)Nc + 3 * (size_t)l;
  T *v = xl_acc + 3 * (size_t)l;
  const T r0 = s[0] * (bl[3 * (size_t)l] - v[0]), r1 = s[1] * (bl[3 * (size_t)l + 1] - v[1]), r2 = s[2] * (bl[3 * (size_t)l + 2] - v[2]);
  v[0] = 0; v[1] = 0; v[2] = 0;
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
// PCG scalars never visit the host.  Iteration k owns slot k of every array:
//   rz[k]  (NS partials) r.z at the start of iteration k (slot 0 filled by the init kernel)
//   den[k] (NS partials) p.Ap of iteration k
//   rr[k]  (NS partials) r.r (matrix-free variant only), pdp[k]: p.D.p
//   rz0[k] running min of |rz_new| before iteration k (inf at k = 0); done[k] loop left before k
// Kernels of iteration k only READ slots written by earlier launches and accumulate
// into slot k / k+1, so there are no intra-launch races.
struct PcgScalars {
  double *rz, *den, *rr, *pdp; // [cap][NS]
  double *rz0;                 // [cap]
  int *done, *iters;           // [cap], [1]
};

// zero all slots, rz0[0] = +inf
__global__ void k_pcg_scalars_init(PcgScalars sc, int cap) {
  for (int i = threadIdx.x; i < cap * NS; i += blockDim.x) { sc.rz[i] = 0.0; sc.den[i] = 0.0; sc.rr[i] = 0.0; sc.pdp[i] = 0.0; }
  for (int i = threadIdx.x; i < cap; i += blockDim.x) { sc.done[i] = 0; sc.rz0[i] = (i == 0) ? __builtin_inf() : 0.0; }
  if (threadIdx.x == 0) sc.iters[0] = 0;
}

// ===========================================================================
// PCG on the explicit Schur complement (solver/pcg_schur.hpp:79-168)
// ===========================================================================

// y = S p for one block ROW per wave: 7 groups of 9 lanes stride over the
// row's block list (upper blocks as A, lower as A^T from the stored upper
// block), partial sums combined with wave shuffles.  Also accumulates p.Ap.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_schur_matvec(int Nc, const int *__restrict__ row_ptr, const int *__restrict__ row_blk,
               const int *__restrict__ row_col, const T *__restrict__ S, const T *__restrict__ x,
               T *__restrict__ y, PcgScalars sc, int k) {
  if (k >= 0) {
    if (sc.done[k]) return;
    if (slot_sum(sc.rz, k) == 0.0) return;
  }
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
    if (lane == 0) slot_add(sc.den, k, (double)dot);
  }
}

// init: r = b_S, z = Minv r, p = z, x = 0, rz[0] = r.z (pcg_schur.hpp:93-105).
// 252 scalars (28 cameras) per block.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_pcgs_init(int Nc, const T *__restrict__ b, const T *__restrict__ Minv, T *__restrict__ r,
            T *__restrict__ z, T *__restrict__ p, T *__restrict__ x, PcgScalars sc) {
  __shared__ double red[4];
  __shared__ T rs[TPB];
  const unsigned t = blockIdx.x * 252u + threadIdx.x;
  const bool on = threadIdx.x < 252 && t < 9u * (unsigned)Nc;
  const T bv = on ? b[t] : T(0);
  rs[threadIdx.x] = bv;
  __syncthreads();
  double part = 0;
  if (on) {
    const T *M = Minv + 81 * (size_t)(t / 9u);
    const int row = (int)(t % 9u);
    const T *rc = rs + (threadIdx.x / 9) * 9;
    T s = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) s += M[row + 9 * q] * rc[q];
    r[t] = bv; z[t] = s; p[t] = s; x[t] = T(0);
    part = (double)(bv * s);
  }
  part = block_sum_256(part, red);
  if (threadIdx.x == 0) slot_add(sc.rz, 0, part);
}

// x_backup = x; x += alpha p; r -= alpha Ap; z = Minv r; rz[k+1] += r.z   (:125-142)
template <typename T>
__global__ void __launch_bounds__(TPB)
k_pcgs_update(int Nc, T *__restrict__ x, T *__restrict__ xb, T *__restrict__ r, T *__restrict__ z,
              const T *__restrict__ p, const T *__restrict__ Ap, const T *__restrict__ Minv,
              PcgScalars sc, int k) {
  if (sc.done[k]) return;
  const double rz = slot_sum(sc.rz, k);
  if (rz == 0.0) return;
  const double den = slot_sum(sc.den, k);
  if (den == 0.0 || den != den) return;
  __shared__ double red[4];
  __shared__ T rs[TPB];
  const T alpha = (T)rz / (T)den;
  const unsigned t = blockIdx.x * 252u + threadIdx.x; // 28 cameras per block
  const bool on = threadIdx.x < 252 && t < 9u * (unsigned)Nc;
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
    const T *M = Minv + 81 * (size_t)(t / 9u);
    const int row = (int)(t % 9u);
    const T *rc = rs + (threadIdx.x / 9) * 9;
    T s = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) s += M[row + 9 * q] * rc[q];
    z[t] = s;
    part = (double)(rn * s);
  }
  part = block_sum_256(part, red);
  if (threadIdx.x == 0) slot_add(sc.rz, k + 1, part);
}

// rejection / beta / p update / tolerance (:143-163); thread 0 publishes slot k+1
template <typename T>
__global__ void __launch_bounds__(TPB)
k_pcgs_direction(int Nc, T *__restrict__ x, const T *__restrict__ xb, T *__restrict__ p,
                 const T *__restrict__ z, PcgScalars sc, int k, double tol, double rejection_ratio) {
  const unsigned t = blockIdx.x * TPB + threadIdx.x;
  const bool first = (t == 0);
  const double rz0 = sc.rz0[k];
  if (sc.done[k]) { if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; } return; }
  const double rz = slot_sum(sc.rz, k);
  const double den = slot_sum(sc.den, k);
  if (rz == 0.0 || den == 0.0 || den != den) { if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; } return; }
  // T-precision scalars, as the reference keeps them in T on the host
  const T rz_new = (T)slot_sum(sc.rz, k + 1);
  const bool reject = (fabs((double)rz_new) > rejection_ratio * rz0) || (rz_new != rz_new);
  if (reject) {
    if (t < 9u * (unsigned)Nc) x[t] = xb[t];
    if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; sc.iters[0] = k + 1; }
    return;
  }
  const T beta = rz_new / (T)rz;
  if (t < 9u * (unsigned)Nc) p[t] = beta * p[t] + z[t];
  if (first) {
    sc.rz0[k + 1] = fmin(rz0, fabs((double)rz_new));
    sc.done[k + 1] = (fabs((double)rz_new) < tol) ? 1 : 0;
    sc.iters[0] = k + 1;
  }
}

// ===========================================================================
// Matrix-free PCG (PCGSolver, solver/pcg.hpp:61-232)
// ===========================================================================
// Operator (J^T rho' J) applied to ps = s .* p with J RECOMPUTED from the camera
// pack (the reference streams the stored J twice per iteration, pcg.hpp:143-163).
// ONE pass in camera-major order, one wave per chunk:
//   u = J ps, w = rho' u;   den += rho' |u|^2
//   camera rows: Jc^T w reduced over the wave     -> op_partial[ch][9]
//   point rows : per-observation Jp^T w           -> g3[pm position][3]
// p.A.p = sum_obs rho' |J ps|^2 + mu p.D.p, so the dot product needs no second pass
// over v2; scaling, damping and the chunk / per-point sums are applied by k_pcg_update.
// VAR (diagnostic builds only, GR_DIAG): 1 = no g3 scatter, 2 = no point gather, 4 = no ps_l gather,
// 8 = no Jacobian math, 16 = no wave reduction.  VAR = 0 is the product kernel.
template <typename T, int VAR = 0>
__global__ void __launch_bounds__(TPB)
k_pcg_operator(int Nc, int nch, const int *__restrict__ chunk_cam, const int *__restrict__ chunk_beg,
               const int *__restrict__ pt_cm, const int *__restrict__ pos_cm, const T *__restrict__ obs_cm,
               const T *__restrict__ pts, const T *__restrict__ pack, int loss_kind, T loss_delta,
               const T *__restrict__ ps, T *__restrict__ g3, T *__restrict__ op_partial, PcgScalars sc, int k) {
  if (sc.done[k]) return;
  if (slot_sum(sc.rz, k) == 0.0) return;
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  using V2 = typename Vec2T<T>::type;
  const size_t pose_dim = 9 * (size_t)Nc;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  double den = 0;
  if (ch < nch) {
    const int c = __builtin_amdgcn_readfirstlane(chunk_cam[ch]);
    T pk[PACK], pc[9];
    load_pack(pack, c, pk);
#pragma unroll
    for (int i = 0; i < 9; ++i) pc[i] = ps[9 * (size_t)c + i];
    T acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = T(0);
    const int beg = __builtin_amdgcn_readfirstlane(chunk_beg[ch]), end = __builtin_amdgcn_readfirstlane(chunk_beg[ch + 1]);
    for (int j = beg + lane; j < end; j += 64) {
      const int l = pt_cm[j];
      const size_t a = (size_t)pos_cm[j];
      const T *pl = ps + pose_dim + 3 * (size_t)((VAR & 4) ? (j & 1023) : l);
      const V2 o = reinterpret_cast<const V2 *>(obs_cm)[j];
      T e0, e1, Jc[18], Jp[6];
      const size_t lp = (VAR & 2) ? (size_t)(j & 1023) : (size_t)l;
      if (VAR & 8) {
        e0 = o.x; e1 = o.y;
#pragma unroll
        for (int i = 0; i < 18; ++i) Jc[i] = pts[3 * lp + (i % 3)] + pk[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) Jp[i] = pts[3 * lp + (i % 3)] - pk[i];
      } else
        bal_linearize(pk, pts[3 * lp], pts[3 * lp + 1], pts[3 * lp + 2], o.x, o.y, e0, e1, Jc, Jp);
      const T w = loss_drho(loss_kind, loss_delta, e0 * e0 + e1 * e1);
      T u0 = Jp[0] * pl[0] + Jp[2] * pl[1] + Jp[4] * pl[2];
      T u1 = Jp[1] * pl[0] + Jp[3] * pl[1] + Jp[5] * pl[2];
#pragma unroll
      for (int i = 0; i < 9; ++i) { u0 += Jc[2 * i] * pc[i]; u1 += Jc[2 * i + 1] * pc[i]; }
      den += (double)(w * (u0 * u0 + u1 * u1));
      u0 *= w; u1 *= w;
#pragma unroll
      for (int i = 0; i < 9; ++i) acc[i] += Jc[2 * i] * u0 + Jc[2 * i + 1] * u1;
      T *g = g3 + 3 * ((VAR & 1) ? (size_t)j : a);
      g[0] = Jp[0] * u0 + Jp[1] * u1;
      g[1] = Jp[2] * u0 + Jp[3] * u1;
      g[2] = Jp[4] * u0 + Jp[5] * u1;
    }
    const T tot = (VAR & 16) ? acc[lane & 7] + acc[8] : wave_transpose_sum<T, 16>(acc, lane);
    if ((lane & 3) == 0 && (lane >> 2) < 9) op_partial[9 * (size_t)ch + (lane >> 2)] = tot;
  }
  den = block_sum_256(den, red);
  if (threadIdx.x == 0) slot_add(sc.den, k, den);
}

// z' = Minv r for the full system (9x9 camera blocks then 3x3 point blocks).
// MODE 0 (init): r = s .* b^u, x = 0.
// MODE 1: v2 = s .* (operator sums) + mu d .* p formed on the fly (camera rows: fixed-order
//         sum of the chunk partials), x_backup = x; x += alpha p; r -= alpha v2.
// Accumulates rr[slot] = r.r and rz[slot] = r.z'  (the reference applies the
// preconditioner to r/||r||, pcg.hpp:108-118,171-183; Minv is linear, so
// z = z'/||r|| and r.z = (r.z')/||r|| are formed from these two sums).
// 252 scalars per block: 28 cameras or 84 points; camera blocks first.
template <typename T, int MODE, bool IDENTITY>
__global__ void __launch_bounds__(TPB)
k_pcg_update(int Nc, int Np, const T *__restrict__ bu, const T *__restrict__ scales, T *__restrict__ x,
             T *__restrict__ xb, T *__restrict__ r, T *__restrict__ zt, const T *__restrict__ p,
             const T *__restrict__ g3, const int *__restrict__ pt_ptr, const T *__restrict__ op_partial,
             const int *__restrict__ cam_chunk_ptr, const T *__restrict__ diag, double mu, int use_identity,
             const T *__restrict__ MinvC, const T *__restrict__ MinvP, PcgScalars sc, int k) {
  T alpha = 0;
  if (MODE == 1) {
    if (sc.done[k]) return;
    const double rzs = slot_sum(sc.rz, k);
    if (rzs == 0.0) return;
    // reference rz = r.z with z = Minv (r/||r||)  ->  (r.z')/||r||
    const T rz = (T)rzs * (T)(1.0 / (double)(T)sqrt((double)(T)slot_sum(sc.rr, k)));
    // p.A.p = sum_obs rho'|J ps|^2 + mu p.D.p
    const T den = (T)(slot_sum(sc.den, k) + mu * slot_sum(sc.pdp, k));
    alpha = rz / den;
  }
  __shared__ double red[4];
  __shared__ T rs[TPB];
  const unsigned pose_dim = 9u * (unsigned)Nc, n = pose_dim + 3u * (unsigned)Np;
  const unsigned cam_blocks = (pose_dim + 251u) / 252u;
  unsigned t;
  bool is_cam;
  if (blockIdx.x < cam_blocks) { t = blockIdx.x * 252u + threadIdx.x; is_cam = true; }
  else { t = pose_dim + (blockIdx.x - cam_blocks) * 252u + threadIdx.x; is_cam = false; }
  const bool on = threadIdx.x < 252 && (is_cam ? t < pose_dim : t < n);
  T rn = 0;
  if (on) {
    if (MODE == 0) { rn = scales[t] * bu[t]; x[t] = T(0); }
    else {
      T raw;
      if (is_cam) {
        const unsigned c = t / 9u, i = t % 9u;
        raw = 0;
        for (int ch = cam_chunk_ptr[c]; ch < cam_chunk_ptr[c + 1]; ++ch) raw += op_partial[9 * (size_t)ch + i];
      } else {
        const unsigned q0 = t - pose_dim, l = q0 / 3u, i = q0 % 3u;
        raw = 0;
        for (int a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) raw += g3[3 * (size_t)a + i];
      }
      const T pv = p[t];
      const T v2 = scales[t] * raw + (use_identity ? (T)mu * pv : (T)mu * diag[t] * pv);
      const T xo = x[t];
      xb[t] = xo;
      x[t] = alpha * pv + xo;
      rn = -alpha * v2 + r[t];
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
      const T *M = MinvC + 81 * (size_t)(t / 9u);
      const int row = (int)(t % 9u);
      const T *rc = rs + (threadIdx.x / 9) * 9;
#pragma unroll
      for (int q = 0; q < 9; ++q) s += M[row + 9 * q] * rc[q];
    } else {
      const unsigned q0 = t - pose_dim;
      const T *M = MinvP + 9 * (size_t)(q0 / 3u);
      const int row = (int)(q0 % 3u);
      const T *rc = rs + (threadIdx.x / 3) * 3;
      s = M[row] * rc[0] + M[row + 3] * rc[1] + M[row + 6] * rc[2];
    }
    zt[t] = s;
    prr = (double)(rn * rn);
    prz = (double)(rn * s);
  }
  const int slot = (MODE == 0) ? 0 : k + 1;
  prr = block_sum_256(prr, red);
  if (threadIdx.x == 0) slot_add(sc.rr, slot, prr);
  prz = block_sum_256(prz, red);
  if (threadIdx.x == 0) slot_add(sc.rz, slot, prz);
}

// Direction kernel of the matrix-free PCG (pcg.hpp:108-127 for k = -1, :184-217 otherwise).
// sc.rz[k] holds r.z' and sc.rr[k] holds r.r; the reference's rz is rz'/sqrt(rr).
// Also: ps = s .* p for the operator and pdp[k+1] = p.D.p.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_pcg_direction(unsigned n, unsigned pose_dim, T *__restrict__ x, const T *__restrict__ xb, T *__restrict__ p,
                T *__restrict__ ps, const T *__restrict__ zt, const T *__restrict__ scales,
                const T *__restrict__ diag, int use_identity, PcgScalars sc, int k,
                double tol, double rejection_ratio) {
  __shared__ double red[4];
  const unsigned t = blockIdx.x * TPB + threadIdx.x;
  const bool first = (t == 0);
  T pn = 0;
  bool wrote = false;
  if (k < 0) { // p = z = z'/||r||
    const T scale = (T)(1.0 / (double)(T)sqrt((double)(T)slot_sum(sc.rr, 0)));
    if (t < n) { pn = scale * zt[t]; wrote = true; }
  } else {
    const double rz0 = sc.rz0[k];
    if (sc.done[k]) { if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; } return; }
    const double rzs = slot_sum(sc.rz, k);
    if (rzs == 0.0) { if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; } return; }
    const T scale_old = (T)(1.0 / (double)(T)sqrt((double)(T)slot_sum(sc.rr, k)));
    const T scale_new = (T)(1.0 / (double)(T)sqrt((double)(T)slot_sum(sc.rr, k + 1)));
    const T rz = (T)rzs * scale_old;
    const T rz_new = (T)slot_sum(sc.rz, k + 1) * scale_new;
    const bool reject = (fabs((double)rz_new) > rejection_ratio * rz0) || (rz_new != rz_new);
    if (reject) {
      if (t < n) x[t] = xb[t];
      if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; sc.iters[0] = k + 1; }
      return;
    }
    const T beta = rz_new / rz;
    if (t < n) { pn = beta * p[t] + scale_new * zt[t]; wrote = true; }
    if (first) {
      sc.rz0[k + 1] = fmin(rz0, fabs((double)rz_new));
      sc.done[k + 1] = (fabs((double)rz_new) < tol) ? 1 : 0;
      sc.iters[0] = k + 1;
    }
  }
  double pdp = 0;
  if (wrote) {
    p[t] = pn;
    ps[t] = scales[t] * pn;
    pdp = use_identity ? (double)(pn * pn) : (double)(diag[t] * pn * pn);
  }
  pdp = block_sum_256(pdp, red);
  if (threadIdx.x == 0) slot_add(sc.pdp, k + 1, pdp);
}

} // namespace gr
