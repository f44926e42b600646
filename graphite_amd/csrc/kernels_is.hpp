// Implicit Schur-complement PCG — gfx950.
//
// Same iterates as PCGSchurSolver + BlockJacobiSchurPreconditioner (solver/pcg_schur.hpp:79-168,
// preconditioner/block_jacobi_schur.hpp) up to rounding, but S = Hpp - Hpl Hll^-1 Hpl^T is never
// formed:   S p = Hpp p - Hpl ( Hll^-1 ( Hpl^T p ) )
// is applied with two camera-major passes over the observations that RECOMPUTE the Jacobians
// (like the matrix-free operator) and a per-point 3x3 multiply in between.  Only the 9x9 DIAGONAL
// blocks of S (the preconditioner) and b_S are accumulated, in one more pass per LM iteration.
// What this removes from the hot path: the Pi = sum k_l(k_l+1)/2 block products (3 ms per LM
// iteration on the Venice-1778 shape, 81 atomics each in the reference, ops/schur.hpp:155-188),
// the 27-scalar Hcp block per observation, and the O(Pi) symbolic phase.  The explicit S stays
// available (gr_bal_schur_update_values) for parity checks and direct solvers.
//
// Scaled space with unscaled blocks (kernels.hpp): with q = s_c .* p_c, M' = Dp (Hll_s + damp)^-1 Dp,
//   (S p)_c = s_c .* ( Hcc^u q_c - sum_obs Jc^T w Jp M'_l sum_obs' Jp^T w Jc q_c' ) + mu clamp(d_c) p_c
#pragma once
#include "kernels_mf.hpp"

namespace gr {

// Walk this block's camera-major tiles (persistent, XCD-aware, next tile's index streams
// prefetched).  body(j, valid, c, l, a, ox, oy) is called by ALL threads once per tile.
template <typename T, typename Body>
__device__ __forceinline__ void cm_tiles(int No, int ntiles, const int *__restrict__ cam_cm,
                                         const int *__restrict__ pt_cm, const int *__restrict__ pos_cm,
                                         const T *__restrict__ obs_cm, Body &&body) {
  using V2 = typename Vec2T<T>::type;
  int t0, t1, tstep;
  xcd_tile_range(ntiles, t0, t1, tstep);
  int j = t0 * TPB + threadIdx.x;
  bool valid = t0 < t1 && j < No;
  int c_n = -1, l_n = 0, a_n = 0;
  V2 o_n{};
  if (valid) { c_n = cam_cm[j]; l_n = pt_cm[j]; a_n = pos_cm ? pos_cm[j] : j; o_n = reinterpret_cast<const V2 *>(obs_cm)[j]; } // pos_cm == nullptr: slot = observation-order position
  for (int t = t0; t < t1; t += tstep) {
    const int c = c_n, l = l_n, a = a_n;
    const V2 o = o_n;
    const int jn = j + tstep * TPB;
    const bool validn = (t + tstep < t1) && jn < No;
    if (validn) { c_n = cam_cm[jn]; l_n = pt_cm[jn]; a_n = pos_cm ? pos_cm[jn] : jn; o_n = reinterpret_cast<const V2 *>(obs_cm)[jn]; }
    body(j, valid, c, l, a, o.x, o.y);
    valid = validn;
    j = jn;
  }
}

// reduce NV (16 or 64) per-lane values once per distinct camera of the wave and store the first
// NOUT of them at out[seg * NOUT + i]; build(mine, v) fills the lane's values (zero weight if !mine)
// segf = flat id of the 64-observation block's first (run, block) segment (blk_seg[j >> 6]); seg_slot maps flat ids to
// the camera-major slots that the consumers read
template <typename T, int NV, int NOUT, typename Build>
__device__ __forceinline__ void reduce_by_camera(bool valid, int c, int segf, const int *__restrict__ seg_slot, int lane, T *__restrict__ out, Build &&build) {
  unsigned long long remaining = __ballot(valid);
  while (remaining) {
    const int leader = __builtin_ctzll(remaining);
    const int cl = __builtin_amdgcn_readlane(c, leader); // leader is wave-uniform (from a ballot): v_readlane, not ds_bpermute
    const int segl = seg_slot[segf++];
    const bool mine = valid && c == cl;
    T v[NV];
    build(mine, v);
    const T tot = wave_transpose_sum<T, NV>(v, lane);
    if (NV == 64) { if (lane < NOUT) out[(size_t)NOUT * segl + lane] = tot; }
    else { if ((lane & 3) == 0 && (lane >> 2) < NOUT) out[(size_t)NOUT * segl + (lane >> 2)] = tot; }
    remaining &= ~__ballot(mine);
  }
}

// Once per solve: per (wave, camera) segment partials of
//   D   = sum_obs Hcp_o M'_l Hcp_o^T   (45 unique)      -> diagonal blocks of S
//   rhs = sum_obs Hcp_o v_l            (9),  v_l = M'_l bl^u  -> b_S
// with Hcp_o = w Jc^T Jp recomputed.  (schur.hpp:649-734 restricted to i == j, :901-920)
template <typename T, typename JT = T>
__global__ void __launch_bounds__(TPB)
k_is_prepare(int No, int ntiles, const int *__restrict__ cam_cm, const int *__restrict__ pt_cm,
             const int *__restrict__ pos_cm, const T *__restrict__ obs_cm, const int *__restrict__ blk_seg,
             const int *__restrict__ seg_slot, const T *__restrict__ pts, const T *__restrict__ pack,
             int loss_kind, T loss_delta, const T *__restrict__ Mp, const T *__restrict__ vl,
             T *__restrict__ cam_partial) {
  const int lane = threadIdx.x & 63;
  cm_tiles<T>(No, ntiles, cam_cm, pt_cm, pos_cm, obs_cm, [&](int j, bool valid, int c, int l, int, T ox, T oy) {
    T h[27], A[27], rhs[9];
#pragma unroll
    for (int i = 0; i < 27; ++i) { h[i] = T(0); A[i] = T(0); }
#pragma unroll
    for (int i = 0; i < 9; ++i) rhs[i] = T(0);
    if (valid) {
      T pk[PACK], e0, e1, Jc[18], Jp[6];
      load_pack(pack, c, pk);
      bal_linearize_j<T, JT>(pk, pts[3 * (size_t)l], pts[3 * (size_t)l + 1], pts[3 * (size_t)l + 2], ox, oy, e0, e1, Jc, Jp);
      const T w = loss_drho(loss_kind, loss_delta, e0 * e0 + e1 * e1);
      const T *m = Mp + 9 * (size_t)l;
      const T *v = vl + 3 * (size_t)l;
      const T wp[6] = {w * Jp[0], w * Jp[1], w * Jp[2], w * Jp[3], w * Jp[4], w * Jp[5]};
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        h[r] = Jc[2 * r] * wp[0] + Jc[2 * r + 1] * wp[1];
        h[r + 9] = Jc[2 * r] * wp[2] + Jc[2 * r + 1] * wp[3];
        h[r + 18] = Jc[2 * r] * wp[4] + Jc[2 * r + 1] * wp[5];
      }
      const T m0 = m[0], m1 = m[1], m2 = m[2], m4 = m[4], m5 = m[5], m8 = m[8]; // symmetric 3x3
      const T v0 = v[0], v1 = v[1], v2 = v[2];
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        A[r] = h[r] * m0 + h[r + 9] * m1 + h[r + 18] * m2;
        A[r + 9] = h[r] * m1 + h[r + 9] * m4 + h[r + 18] * m5;
        A[r + 18] = h[r] * m2 + h[r + 9] * m5 + h[r + 18] * m8;
        rhs[r] = h[r] * v0 + h[r + 9] * v1 + h[r + 18] * v2;
      }
    }
    reduce_by_camera<T, 64, 54>(valid, c, blk_seg[__builtin_amdgcn_readfirstlane(j >> 6)], seg_slot, lane, cam_partial, [&](bool mine, T(&acc)[64]) {
      const T z = mine ? T(1) : T(0);
      int k = 0;
#pragma unroll
      for (int col = 0; col < 9; ++col) {
#pragma unroll
        for (int row = 0; row <= col; ++row) acc[k++] = z * (A[row] * h[col] + A[row + 9] * h[col + 9] + A[row + 18] * h[col + 18]);
        acc[45 + col] = z * rhs[col];
      }
#pragma unroll
      for (int i = 54; i < 64; ++i) acc[i] = T(0);
    });
  });
}

// Sdiag_c = Dc (Hcc^u + damping - D) Dc  (81 per camera, column-major),  b_S = Dc (bc^u - rhs)
template <typename T>
__global__ void k_is_finalize(int Nc, const int *__restrict__ cam_seg_ptr, const T *__restrict__ cam_partial,
                              const T *__restrict__ Hcc, const T *__restrict__ bc, const T *__restrict__ scales,
                              double mu, int use_identity, T *__restrict__ Sdiag, T *__restrict__ b_schur,
                              const T *__restrict__ raw_in = nullptr, T *__restrict__ raw_out = nullptr,
                              const unsigned char *__restrict__ cam_fixed = nullptr) {
  // multi-GPU: stage 1 (raw_out) leaves this shard's 90 sums per camera for the all-reduce,
  // stage 2 (raw_in) combines the all-reduced sums with the (already global) Hcc, bc
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 90u * (unsigned)Nc) return;
  const unsigned c = t / 90u, e = t % 90u;
  int idx;
  unsigned row = 0, col = 0;
  if (e < 81u) {
    row = e % 9u; col = e / 9u;
    const unsigned r = row < col ? row : col, cc = row < col ? col : row;
    idx = (int)(cc * (cc + 1) / 2 + r);
  } else idx = 45 + (int)(e - 81u);
  T s = 0;
  if (raw_in) s = raw_in[t];
  else for (int sg = cam_seg_ptr[c]; sg < cam_seg_ptr[c + 1]; ++sg) s += cam_partial[54 * (size_t)sg + idx];
  if (cam_fixed && cam_fixed[c]) s = T(0); // a fixed camera has no Jacobian block: nothing is eliminated into its (empty) row
  if (raw_out) { raw_out[t] = s; return; }
  if (e < 81u) {
    const T sr = scales[9 * c + row], sc = scales[9 * c + col];
    T hh = sr * Hcc[81 * (size_t)c + e] * sc;
    if (row == col) hh = damp_diag(hh, mu, use_identity);
    Sdiag[81 * (size_t)c + e] = hh - sr * s * sc;
  } else {
    const unsigned r = e - 81u;
    b_schur[9 * c + r] = scales[9 * c + r] * (bc[9 * c + r] - s);
  }
}

// pass 1:  g3[pm slot] = Jp^T w (Jc q_c)      (Hpl^T p, one observation's share)
template <typename T, typename JT = T>
__global__ void __launch_bounds__(TPB)
k_is_pass1(int No, int ntiles, const int *__restrict__ cam_cm, const int *__restrict__ pt_cm,
           const int *__restrict__ pos_cm, const T *__restrict__ obs_cm, const T *__restrict__ pts,
           const T *__restrict__ pack, int loss_kind, T loss_delta, const T *__restrict__ q, T *__restrict__ g3,
           PcgScalars sc, int k) {
  if (k >= 0) {
    if (sc.done[k]) return;
    if (part_sum(sc.rz, sc.np, k) == 0.0) return;
  }
  cm_tiles<T>(No, ntiles, cam_cm, pt_cm, pos_cm, obs_cm, [&](int, bool valid, int c, int l, int a, T ox, T oy) {
    if (!valid) return;
    T pk[PACK], e0, e1, Jc[18], Jp[6];
    load_pack(pack, c, pk);
    bal_linearize_j<T, JT>(pk, pts[3 * (size_t)l], pts[3 * (size_t)l + 1], pts[3 * (size_t)l + 2], ox, oy, e0, e1, Jc, Jp);
    const T w = loss_drho(loss_kind, loss_delta, e0 * e0 + e1 * e1);
    const T *qc = q + 9 * (size_t)c;
    T t0 = 0, t1 = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { const T v = qc[i]; t0 += Jc[2 * i] * v; t1 += Jc[2 * i + 1] * v; }
    t0 *= w; t1 *= w;
    T *g = g3 + 3 * (size_t)a;
    g[0] = Jp[0] * t0 + Jp[1] * t1;
    g[1] = Jp[2] * t0 + Jp[3] * t1;
    g[2] = Jp[4] * t0 + Jp[5] * t1;
  });
}

// per point: y = sum of its observations' g3 (fixed order);
//   MODE 0: zl = M' y                                   (Hll^-1 Hpl^T p in unscaled form)
//   MODE 1: xl = Hll_inv (s_l .* (bl^u - y))            (back-substitution, schur.hpp:279-302)
// gg.gidx != nullptr: g3 is in observation order (pass 1 then stores whole lines) and a point's slots are gathered through
// gidx, the points walked tile by tile on the XCD that wrote the tile (kernels_mf.hpp G3Gather); gridDim.x % 8 == 0 then.
template <typename T, int MODE>
__global__ void k_is_points(int Np, int Nc, const int *__restrict__ pt_ptr, const T *__restrict__ g3,
                            const T *__restrict__ Mp, const T *__restrict__ Hll_inv, const T *__restrict__ bl,
                            const T *__restrict__ scales, T *__restrict__ out, PcgScalars sc, int k, G3Gather gg = G3Gather{}) {
  if (MODE == 0) {
    if (sc.done[k]) return;
    if (part_sum(sc.rz, sc.np, k) == 0.0) return;
  }
  auto point = [&](int l) {
    T y0 = 0, y1 = 0, y2 = 0;
    if (gg.gidx) {
      for (int a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) {
        const T *g = g3 + 3 * (size_t)gg.gidx[a];
        y0 += g[0]; y1 += g[1]; y2 += g[2];
      }
    } else {
      for (int a = pt_ptr[l]; a < pt_ptr[l + 1]; ++a) {
        const T *g = g3 + 3 * (size_t)a;
        y0 += g[0]; y1 += g[1]; y2 += g[2];
      }
    }
    const T *m = (MODE == 0 ? Mp : Hll_inv) + 9 * (size_t)l;
    if (MODE == 1) {
      const T *s = scales + 9 * (size_t)Nc + 3 * (size_t)l;
      y0 = s[0] * (bl[3 * (size_t)l] - y0); y1 = s[1] * (bl[3 * (size_t)l + 1] - y1); y2 = s[2] * (bl[3 * (size_t)l + 2] - y2);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) out[3 * (size_t)l + r] = m[r] * y0 + m[r + 3] * y1 + m[r + 6] * y2;
  };
  if (gg.n_ptiles > 0) {
    const int x = (int)(blockIdx.x & 7), bi = (int)(blockIdx.x >> 3), nb = (int)(gridDim.x >> 3);
    for (int q = x; q < gg.n_ptiles; q += 8) {
      const int P0 = gg.ptile_ptr[q], P1 = gg.ptile_ptr[q + 1];
      for (int base = P0 + bi * (int)blockDim.x; base < P1; base += nb * (int)blockDim.x) {
        const int l = base + (int)threadIdx.x;
        if (l < P1) point(l);
      }
    }
  } else {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < Np) point(l);
  }
}

// pass 2: per (wave, camera) segment partial of  sum_obs Jc^T w Jp z_l
template <typename T, typename JT = T>
__global__ void __launch_bounds__(TPB)
k_is_pass2(int No, int ntiles, const int *__restrict__ cam_cm, const int *__restrict__ pt_cm,
           const int *__restrict__ pos_cm, const T *__restrict__ obs_cm, const int *__restrict__ blk_seg,
           const int *__restrict__ seg_slot, const T *__restrict__ pts, const T *__restrict__ pack,
           int loss_kind, T loss_delta, const T *__restrict__ zl, T *__restrict__ op_partial, PcgScalars sc, int k) {
  if (sc.done[k]) return;
  if (part_sum(sc.rz, sc.np, k) == 0.0) return;
  const int lane = threadIdx.x & 63;
  cm_tiles<T>(No, ntiles, cam_cm, pt_cm, pos_cm, obs_cm, [&](int j, bool valid, int c, int l, int, T ox, T oy) {
    T acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = T(0);
    if (valid) {
      T pk[PACK], e0, e1, Jc[18], Jp[6];
      load_pack(pack, c, pk);
      bal_linearize_j<T, JT>(pk, pts[3 * (size_t)l], pts[3 * (size_t)l + 1], pts[3 * (size_t)l + 2], ox, oy, e0, e1, Jc, Jp);
      const T w = loss_drho(loss_kind, loss_delta, e0 * e0 + e1 * e1);
      const T *z = zl + 3 * (size_t)l;
      const T v0 = w * (Jp[0] * z[0] + Jp[2] * z[1] + Jp[4] * z[2]);
      const T v1 = w * (Jp[1] * z[0] + Jp[3] * z[1] + Jp[5] * z[2]);
#pragma unroll
      for (int i = 0; i < 9; ++i) acc[i] = Jc[2 * i] * v0 + Jc[2 * i + 1] * v1;
    }
    reduce_by_camera<T, 16, 9>(valid, c, blk_seg[__builtin_amdgcn_readfirstlane(j >> 6)], seg_slot, lane, op_partial, [&](bool mine, T(&m)[16]) {
#pragma unroll
      for (int i = 0; i < 9; ++i) m[i] = mine ? acc[i] : T(0);
#pragma unroll
      for (int i = 9; i < 16; ++i) m[i] = T(0);
    });
  });
}

// Ap_c = s_c .* (Hcc^u q_c - sum of segment partials) + mu clamp(d_c) p_c ;  den[k] += p.Ap
// 252 scalars (28 cameras) per block, q staged in LDS for the 9x9 product.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_is_apply(int Nc, const int *__restrict__ cam_seg_ptr, const T *__restrict__ op_partial,
           const T *__restrict__ Hcc, const T *__restrict__ scales, const T *__restrict__ p,
           const T *__restrict__ q, double mu, int use_identity, T *__restrict__ Ap, PcgScalars sc, int k,
           const T *__restrict__ rawc = nullptr, const unsigned char *__restrict__ cam_fixed = nullptr) {
  if (sc.done[k]) return;
  if (part_sum(sc.rz, sc.np, k) == 0.0) return;
  __shared__ double red[4];
  __shared__ T qs[TPB];
  const unsigned t = blockIdx.x * 252u + threadIdx.x;
  const bool on = threadIdx.x < 252 && t < 9u * (unsigned)Nc;
  qs[threadIdx.x] = on ? q[t] : T(0);
  __syncthreads();
  double part = 0;
  if (on) {
    const unsigned c = t / 9u, r = t % 9u;
    const T *H = Hcc + 81 * (size_t)c;
    const T *qc = qs + (threadIdx.x / 9) * 9;
    T hq = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) hq += H[r + 9 * i] * qc[i];
    T sub = 0;
    if (rawc) sub = rawc[t]; // multi-GPU: segment sums all-reduced over the landmark shards
    else for (int sg = cam_seg_ptr[c]; sg < cam_seg_ptr[c + 1]; ++sg) sub += op_partial[9 * (size_t)sg + r];
    if (cam_fixed && cam_fixed[c]) sub = T(0);
    const T s = scales[t], pv = p[t];
    const T d = s * H[10 * r] * s; // scaled diagonal of Hcc (prev_diag of hessian.hpp:102-134)
    const T damp = use_identity ? (T)mu * pv : (T)(mu * clampd((double)d, 1.0e-6, 1.0e32)) * pv;
    const T out = s * (hq - sub) + damp;
    Ap[t] = out;
    part = (double)(out * pv);
  }
  part = block_sum_256(part, red);
  if (threadIdx.x == 0) part_store(sc.den, sc.np, k, part);
}

} // namespace gr
