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
// This file: shared helpers, the Schur-complement path (PCGSchurSolver) and the small
// block inverses.  Linearisation, chi2 and the matrix-free PCG live in kernels_mf.hpp.
#pragma once
#include "bal_device.hpp"
#include "common.hpp"
#include <cfloat>

namespace gr {

constexpr int CHUNK = 128; // observations of one camera handled by one wave (2 per lane)

template <typename T> __device__ __forceinline__ void load_pack(const T *__restrict__ pack, int c, T *pk) {
  const T *src = pack + PACK * (size_t)c;
#pragma unroll
  for (int i = 0; i < PACK; ++i) pk[i] = src[i];
}

// ---------------------------------------------------------------------------
// camera pack (+ optional x += dx .* s on the cameras first): Nc threads
template <typename T>
__global__ void k_campack(int Nc, T *__restrict__ cams, T *__restrict__ pack, const T *__restrict__ dx,
                          const T *__restrict__ scales, T *__restrict__ backup, const LmDev *__restrict__ lm = nullptr) {
  if (lm && lm->stop) return;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Nc) return;
  T cam[9], pk[PACK];
#pragma unroll
  for (int i = 0; i < 9; ++i) cam[i] = cams[9 * c + i];
  if (dx) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (backup) backup[9 * c + i] = cam[i]; // Graph::backup_parameters fused (graph.hpp:302-309)
      cam[i] += dx[9 * c + i] * scales[9 * c + i];
      cams[9 * c + i] = cam[i];
    }
  }
  make_campack(cam, pk);
#pragma unroll
  for (int i = 0; i < PACK; ++i) pack[PACK * (size_t)c + i] = pk[i];
}

// ---------------------------------------------------------------------------
// x += dx .* s     (ops/update.hpp:11-31 with additive update)
template <typename T>
__global__ void k_apply_update(unsigned n, T *__restrict__ x, const T *__restrict__ dx,
                               const T *__restrict__ scales, T *__restrict__ backup) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const T xo = x[i];
    if (backup) backup[i] = xo;
    x[i] = xo + dx[i] * scales[i];
  }
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

// ---------------------------------------------------------------------------
// PCG scalars never visit the host.  Iteration k owns slot k of every array:
//   rz[k]  (NS partials) r.z at the start of iteration k (slot 0 filled by the init kernel)
//   den[k] (NS partials) p.Ap of iteration k
//   rr[k]  (NS partials) r.r (matrix-free variant only), pdp[k]: p.D.p
//   rz0[k] running min of |rz_new| before iteration k (inf at k = 0); done[k] loop left before k
// Kernels of iteration k only READ slots written by earlier launches and accumulate
// into slot k / k+1, so there are no intra-launch races.
// The partial sums of rz / den are ONE PER BLOCK of the producing kernel, stored (not atomically added)
// at partial index blockIdx.x and re-summed by every consumer wave in a fixed order: the scalars are then
// bit-reproducible, and identical on every rank of a landmark-sharded run whose camera-space vectors are
// replicated — the loop decisions (and with them the collectives each rank enqueues) cannot diverge.
struct PcgScalars {
  double *rz, *den;            // [cap][np] per-block partials
  int np;                      // partials per scalar: multiple of 64, >= the largest producer grid
  double *rz0;                 // [cap]
  int *done, *iters;           // [cap], [1]
  volatile int *hflag;         // pinned host memory [cap]: 1 = iteration finished, 2 = loop left (may be null)
  volatile int *hiters;        // pinned host mirror of iters (may be null)
};

__device__ __forceinline__ void part_store(double *base, int np, int k, double v) { base[(size_t)k * np + blockIdx.x] = v; }
__device__ __forceinline__ double part_sum(const double *base, int np, int k) { // whole wave must call; fixed order
  double s = 0;
  const double *q = base + (size_t)k * np;
  for (int i = threadIdx.x & 63; i < np; i += 64) s += q[i];
  return wave_allsum(s);
}

// Per point: scaled + damped Hll, its inverse (scaled space, kept for
// back-substitution and parity), M' = Dp Hll^-1 Dp and v = M' bl^u.
// (schur.hpp:1067-1114 batched inverse; hessian.hpp:136-176 damping)
template <typename T>
__global__ void k_point_prepare(int Np, int Nc, const T *__restrict__ Hll, const T *__restrict__ bl,
                                const T *__restrict__ scales, double mu, int use_identity,
                                T *__restrict__ Hll_inv, T *__restrict__ Mp, T *__restrict__ vl,
                                PcgScalars pcg = PcgScalars{}, int cap = 0,
                                // fixed points: eliminated with a zero "inverse", i.e. they add nothing to S and b_S, the implicit
                                // operator skips them and their back-substituted step is 0
                                const unsigned char *__restrict__ pt_fixed = nullptr) {
  if (pcg.rz && blockIdx.x == gridDim.x - 1) { // one extra block: reset of the PCG scalars of the solve that follows
    for (int i = threadIdx.x; i < cap * pcg.np; i += blockDim.x) { pcg.rz[i] = 0.0; pcg.den[i] = 0.0; }
    for (int i = threadIdx.x; i < cap; i += blockDim.x) { pcg.done[i] = 0; pcg.rz0[i] = (i == 0) ? __builtin_inf() : 0.0; }
    if (threadIdx.x == 0) pcg.iters[0] = 0;
    return;
  }
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= Np) return;
  const T *s = scales + 9 * (size_t)Nc + 3 * (size_t)l;
  const T *H = Hll + 9 * (size_t)l;
  // the whole chain scale -> scaled damped block -> inverse -> M' -> v in double from the stored sums (scale_hat): the inversion
  // amplifies entry errors by the block's condition number (1e3-1e4 for weakly observed points), so in fp32 the block it sees must
  // not carry roundings of its own; only the outputs are rounded to T
  const double sc[3] = {scale_hat(s[0], H[0]), scale_hat(s[1], H[4]), scale_hat(s[2], H[8])};
  double A[9];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double v = sc[r] * (double)H[r + 3 * c] * sc[c];
      A[r + 3 * c] = (r == c) ? (use_identity ? v + mu : v + mu * clampd(v, 1.0e-6, 1.0e32)) : v;
    }
  spd_inverse<3>(A);
  const bool fixed = pt_fixed && pt_fixed[l];
  if (fixed) {
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) Hll_inv[9 * (size_t)l + i] = (T)A[i];
  double m[9];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) { m[r + 3 * c] = sc[r] * A[r + 3 * c] * sc[c]; Mp[9 * (size_t)l + r + 3 * c] = (T)m[r + 3 * c]; }
  const double b0 = (double)bl[3 * (size_t)l], b1 = (double)bl[3 * (size_t)l + 1], b2 = (double)bl[3 * (size_t)l + 2];
#pragma unroll
  for (int r = 0; r < 3; ++r) vl[3 * (size_t)l + r] = (T)(m[r] * b0 + m[r + 3] * b1 + m[r + 6] * b2);
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
  if (i == j && Hcc) { // Hcc == nullptr: a landmark shard other than rank 0 (the camera blocks are added once)
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

#ifndef SCHUR_WAVES
#define SCHUR_WAVES 4
#endif
template <typename T>
__global__ void __launch_bounds__(TPB, SCHUR_WAVES)
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
  // Operands through LDS.  Every lane of a group needs the SAME 27 scalars of Ha and 9 of M (plus its own column of Hb): fetched
  // redundantly they are 39 live fp64 registers per lane and every round of 7 products waits one full memory latency for them at
  // 3 waves per SIMD (352-369 us on Ladybug-1723).  Here the 9 lanes of a group fetch the 36 shared scalars ONCE (4 per lane, the
  // NEXT round's while the current one is multiplied), park them in the group's LDS strip and read them back as broadcasts.  One
  // wave owns a strip: a wave-level fence (wave_lds_fence) orders its writes, its reads and the next round's writes — no
  // workgroup barrier.
  __shared__ T strip[4][7][40]; // [wave][group][27 Ha | 9 M | pad]
  if (g < 7) {
    T *sg = strip[threadIdx.x >> 6][g];
    const int q_end = item_end[item];
    int q = item_beg[item] + g;
    int a_n = 0, b_n = 0, pm_n = 0;
    T h_n[3], hb_n[3], m_n = T(0);
    auto fetch = [&](int qq) { // indices (dependent chain) and, from them, this lane's share of the operands of product qq
      a_n = prod_a[qq]; b_n = prod_b[qq]; pm_n = pt_pm[a_n];
      const T *ha = Hcp + 27 * (size_t)a_n + c, *hb = Hcp + 27 * (size_t)b_n + c;
      h_n[0] = ha[0]; h_n[1] = ha[9]; h_n[2] = ha[18];
      hb_n[0] = hb[0]; hb_n[1] = hb[9]; hb_n[2] = hb[18];
      m_n = Mp[9 * (size_t)pm_n + c];
    };
    if (q < q_end) fetch(q);
    for (; q < q_end; q += 7) {
      sg[c] = h_n[0]; sg[c + 9] = h_n[1]; sg[c + 18] = h_n[2]; sg[27 + c] = m_n;
      const T hb0 = hb_n[0], hb1 = hb_n[1], hb2 = hb_n[2];
      if (q + 7 < q_end) fetch(q + 7);
      wave_lds_fence(); // the strip is written and read by different lanes of this wave
      T mv[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) mv[i] = sg[27 + i];
      const T u0 = mv[0] * hb0 + mv[3] * hb1 + mv[6] * hb2;
      const T u1 = mv[1] * hb0 + mv[4] * hb1 + mv[7] * hb2;
      const T u2 = mv[2] * hb0 + mv[5] * hb1 + mv[8] * hb2;
#pragma unroll
      for (int r = 0; r < 9; ++r) acc[r] += sg[r] * u0 + sg[r + 9] * u1 + sg[r + 18] * u2;
      wave_lds_fence(); // ... and rewritten by the next round
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
                                  const T *__restrict__ bc, const T *__restrict__ scales, T *__restrict__ b_schur,
                                  int cam_weight = 1) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 9u * (unsigned)Nc) return;
  const unsigned c = t / 9u, r = t % 9u;
  T y = 0;
  for (int ch = cam_chunk_ptr[c]; ch < cam_chunk_ptr[c + 1]; ++ch) y += partial9[9 * (size_t)ch + r];
  b_schur[t] = scales[t] * ((cam_weight ? bc[t] : T(0)) - y);
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
               T *__restrict__ y, PcgScalars sc, int k, const LmDev *__restrict__ lm = nullptr) {
  if (lm && lm->stop) return; // device-decided LM loop (kernels_sf.hpp): the trial step this head belongs to was not accepted
  if (k >= 0) {
    if (sc.done[k]) return;
    if (part_sum(sc.rz, sc.np, k) == 0.0) return;
  }
  __shared__ double wdot[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  const int g = lane / 9, r = lane % 9;
  T acc = 0;
  if (g < 7 && i < Nc) {
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
  if (lane < 9 && i < Nc) {
    y[9 * (size_t)i + r] = tot;
    dot = tot * x[9 * (size_t)i + r];
  }
  if (k >= 0) { // one partial per block (4 rows), the waves summed in fixed order
    const double dw = wave_sum((double)dot);
    if (lane == 0) wdot[wave] = dw;
    __syncthreads();
    if (threadIdx.x == 0) part_store(sc.den, sc.np, k, wdot[0] + wdot[1] + wdot[2] + wdot[3]);
  }
}


// One launch, thread per camera, for everything between "S is complete" and the first iteration:
//   MODE 0: b_S = Dc (bc^u - chunk partials) (schur.hpp:901-920), M = (S diagonal block)^-1
//   MODE 1: b_S given (all-reduced over landmark shards), M = (S diagonal block)^-1
//   MODE 2: b_S given (implicit Schur), M = Sdiag_c^-1
// then r = b_S, z = p = M r, x = 0, q = s .* p (implicit only), rz[0] += r.z
// (block_jacobi_schur.hpp:114-178, pcg_schur.hpp:79-104).  The PCG scalars were reset by k_point_prepare.
template <typename T, int MODE>
__global__ void __launch_bounds__(64)
k_schur_pcg_prepare(int Nc, const T *__restrict__ Ssrc, const int *__restrict__ diag_blk,
                    const int *__restrict__ cam_chunk_ptr, const T *__restrict__ partial9, const T *__restrict__ bc,
                    const T *__restrict__ scales, T *__restrict__ b_schur, T *__restrict__ Minv, T *__restrict__ r,
                    T *__restrict__ z, T *__restrict__ p, T *__restrict__ x, T *__restrict__ q, PcgScalars sc, const LmDev *__restrict__ lm = nullptr) {
  if (lm && lm->stop) return;
  const int c = blockIdx.x * 64 + threadIdx.x;
  double part = 0;
  if (c < Nc) {
    T b[9];
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) b[i] = T(0);
      for (int ch = cam_chunk_ptr[c]; ch < cam_chunk_ptr[c + 1]; ++ch)
#pragma unroll
        for (int i = 0; i < 9; ++i) b[i] += partial9[9 * (size_t)ch + i];
#pragma unroll
      for (int i = 0; i < 9; ++i) { b[i] = scales[9 * (size_t)c + i] * (bc[9 * (size_t)c + i] - b[i]); b_schur[9 * (size_t)c + i] = b[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) b[i] = b_schur[9 * (size_t)c + i];
    }
    double A[81];
    const T *B = Ssrc + 81 * (size_t)(MODE != 2 ? diag_blk[c] : c);
#pragma unroll
    for (int i = 0; i < 81; ++i) A[i] = (double)B[i];
    spd_inverse<9>(A);
#pragma unroll
    for (int row = 0; row < 9; ++row) {
      T s = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) s += (T)A[row + 9 * k] * b[k];
      const size_t t = 9 * (size_t)c + row;
      r[t] = b[row]; z[t] = s; p[t] = s; x[t] = T(0);
      if (q) q[t] = scales[t] * s;
      part += (double)(b[row] * s);
    }
#pragma unroll
    for (int i = 0; i < 81; ++i) Minv[81 * (size_t)c + i] = (T)A[i];
  }
  part = wave_sum(part);
  if (threadIdx.x == 0) part_store(sc.rz, sc.np, 0, part);
}

// x_backup = x; x += alpha p; r -= alpha Ap; z = Minv r; rz[k+1] += r.z   (:125-142)
template <typename T>
__global__ void __launch_bounds__(TPB)
k_pcgs_update(int Nc, T *__restrict__ x, T *__restrict__ xb, T *__restrict__ r, T *__restrict__ z,
              const T *__restrict__ p, const T *__restrict__ Ap, const T *__restrict__ Minv,
              PcgScalars sc, int k, const LmDev *__restrict__ lm = nullptr) {
  if (lm && lm->stop) return;
  if (sc.done[k]) return;
  const double rz = part_sum(sc.rz, sc.np, k);
  if (rz == 0.0) return;
  const double den = part_sum(sc.den, sc.np, k);
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
  if (threadIdx.x == 0) part_store(sc.rz, sc.np, k + 1, part);
}

// rejection / beta / p update / tolerance (:143-163); thread 0 publishes slot k+1
template <typename T>
__global__ void __launch_bounds__(TPB)
k_pcgs_direction(int Nc, T *__restrict__ x, const T *__restrict__ xb, T *__restrict__ p,
                 const T *__restrict__ z, T *__restrict__ q, const T *__restrict__ scales, PcgScalars sc,
                 int k, double tol, double rejection_ratio, const LmDev *__restrict__ lm = nullptr) {
  const unsigned t = blockIdx.x * TPB + threadIdx.x;
  const bool first = (t == 0);
  const double rz0 = sc.rz0[k];
  auto publish = [&](int flag) { if (sc.hflag) { sc.hflag[k] = flag; __threadfence_system(); } };
  if (lm && lm->stop) { if (first) publish(2); return; } // the host's loop over the exit flags still ends
  if (sc.done[k]) { if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; publish(2); } return; }
  const double rz = part_sum(sc.rz, sc.np, k);
  const double den = part_sum(sc.den, sc.np, k);
  if (rz == 0.0 || den == 0.0 || den != den) { if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; publish(2); } return; }
  // T-precision scalars, as the reference keeps them in T on the host
  const T rz_new = (T)part_sum(sc.rz, sc.np, k + 1);
  const bool reject = (fabs((double)rz_new) > rejection_ratio * rz0) || (rz_new != rz_new);
  if (reject) {
    if (t < 9u * (unsigned)Nc) x[t] = xb[t];
    if (first) { sc.done[k + 1] = 1; sc.rz0[k + 1] = rz0; sc.iters[0] = k + 1; if (sc.hiters) *sc.hiters = k + 1; publish(2); }
    return;
  }
  const T beta = rz_new / (T)rz;
  if (t < 9u * (unsigned)Nc) {
    const T pn = beta * p[t] + z[t];
    p[t] = pn;
    if (q) q[t] = scales[t] * pn;
  }
  if (first) {
    const int dn = (fabs((double)rz_new) < tol) ? 1 : 0;
    sc.rz0[k + 1] = fmin(rz0, fabs((double)rz_new));
    sc.done[k + 1] = dn;
    sc.iters[0] = k + 1;
    if (sc.hiters) *sc.hiters = k + 1;
    publish(dn ? 2 : 1);
  }
}

// reset of the Schur-PCG scalars (the extra block of k_point_prepare in the host-driven form)
__global__ void k_pcgs_reset(PcgScalars pcg, int cap) {
  for (int i = threadIdx.x; i < cap * pcg.np; i += blockDim.x) { pcg.rz[i] = 0.0; pcg.den[i] = 0.0; }
  for (int i = threadIdx.x; i < cap; i += blockDim.x) { pcg.done[i] = 0; pcg.rz0[i] = (i == 0) ? __builtin_inf() : 0.0; }
  if (threadIdx.x == 0) pcg.iters[0] = 0;
}

// Landmark shards: the two LM scalars (trial chi2, rho denominator) have just been summed over the ranks in device memory;
// mirror them into pinned host memory and raise the sequence word the host polls (same hand-over as the single-GPU kernels'
// own publication: no device-to-host copy, no stream synchronisation on the accept path)
__global__ void k_publish_scalars(const double *__restrict__ d, int count, volatile double *hres, volatile int *hres_seq, int seq) {
  for (int i = 0; i < count; ++i) hres[i] = d[i];
  __threadfence_system();
  *hres_seq = seq;
  __threadfence_system();
}
} // namespace gr
