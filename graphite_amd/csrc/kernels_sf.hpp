// Device-decided LM iteration for PCGSchurSolver on graphs whose reduced camera system is SMALL (Ladybug-49: 49 cameras,
// S = 441 x 441) — gfx950.  solver/pcg_schur.hpp:79-168 + optimizer/levenberg_marquardt.hpp:110-242.
//
// The host-driven form of this solver runs ~18 launches of 2.5-20 us and two host decisions per LM iteration; on such graphs
// every kernel is a latency chain, so the iteration costs what its launches and round trips cost.  Here it is FIVE launches and
// the host only observes:
//   k_linearize<WRITE_HCP>      (kernels_mf.hpp)  residuals, Jacobian blocks, Hcp, segment / point partials, chi2 partials
//   k_finalize_schur                               accept decision of the PREVIOUS trial step (LmDecide, as k_finalize_bj), segment
//                                                  sums -> Hcc, bc, camera scales; point sums -> Hll, bl, point scales, damped
//                                                  inverse, M' = D Hll^-1 D, v = M' bl  (k_linearize_finalize + k_point_prepare)
//   k_schur_reduce                                 S^u products per destination block, multi-item blocks finished by their LAST
//                                                  ARRIVER in item order (no atomics on values, no zero / fix-up launches), and
//                                                  the b_S chunk partials in the same launch (k_schur_multi x 2 + k_schur_products
//                                                  + k_bschur_partial)
//   k_schur_pcg_coop                               b_S, block-Jacobi inverses and ALL inner iterations of the PCG on S: one wave
//                                                  per camera row, three grid barriers per iteration (k_schur_pcg_prepare +
//                                                  (k_schur_matvec, k_pcgs_update, k_pcgs_direction) x iterations + the host's
//                                                  exit-flag round trips)
//   k_backsub_apply                                x_l = Hll^-1 (b_l - Hpl^T x_p), then the trial step: backup, update, camera packs,
//                                                  rho-denominator partials (k_backsub + k_backsub_fixup + k_apply_update_rho)
#pragma once
#include "kernels_mf.hpp"

namespace gr {

// accept decision of the trial step the pending linearisation evaluated — the prologue of k_finalize_bj (kernels_mf.hpp), same
// sums in the same order; returns false when the launch has nothing more to do (step not accepted)
__device__ __forceinline__ bool lm_decide_prologue(const LmDecide &dec, double &mu, double *s_sum /* __shared__ [2] */) {
  if (!dec.seq) return true;
  if (threadIdx.x < 64) {
    const int ln = threadIdx.x;
    double c0 = 0, r0 = 0;
    for (int base = 0; base < dec.n_chi2; base += 512) {
      double q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = base + 64 * u + ln; q[u] = i < dec.n_chi2 ? dec.chi2_partial[i] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) c0 += q[u];
    }
    for (int base = 0; base < dec.n_rho; base += 512) {
      double q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = base + 64 * u + ln; q[u] = i < dec.n_rho ? dec.rho_partial[i] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) r0 += q[u];
    }
    c0 = wave_allsum(c0); r0 = wave_allsum(r0);
    if (ln == 0) { s_sum[0] = c0; s_sum[1] = r0; }
  }
  __syncthreads();
  const double cs = s_sum[0], rs = s_sum[1];
  if (dec.report_only) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (dec.dscal) dec.dscal[0] = cs;
      dec.hres[0] = cs;
      __threadfence_system();
      *dec.hres_seq = dec.seq;
    }
    return true;
  }
  // levenberg_marquardt.hpp:184-197, scalars in the graph precision like the host loop: the caller passes chi2 / mu already rounded
  return true;
}

// LmDecide's arithmetic in T (optimizer/levenberg_marquardt.hpp:20-47,184-197): ok = accepted; mun = the damping that follows
template <typename T> __device__ __forceinline__ bool lm_accept(const LmDecide &dec, double cs, double rs, double &mun) {
  const T chi2v = (T)dec.chi2_cur, new_chi2 = (T)cs;
  const T denom = (T)rs + (T)1.0e-3;
  const T rho = (chi2v - new_chi2) / denom;
  const bool ok = isfinite((double)new_chi2) && rho > T(0);
  double alpha = 1.0 - pow(2.0 * (double)rho - 1.0, 3.0);
  alpha = fmax(fmin(alpha, 2.0 / 3.0), 1.0 / 3.0);
  mun = (double)((T)dec.mu_cur * (T)alpha);
  return ok;
}

struct CoopState {           // k_schur_pcg_coop's cross-workgroup state (device memory)
  unsigned *barrier;         // [1] arrivals, cleared by k_finalize_schur
  double *part;              // [(cap) * 3][Nc] per-row partials: phase 0 r.z (start / new), 1 p.Sp
  int *iters;                // [1] inner iterations run
  volatile int *hiters;      // pinned mirror
  volatile int *fail;        // pinned host word, sticky: != 0 once a grid barrier has timed out (workgroups not co-resident)
  long long *ts;             // pinned [2] wall-clock stamps around the loop (solve_seconds), may be null
  int *fail_dev = nullptr;   // the same word in device memory: the NEXT head's decision reads it (a step computed behind a timed-out barrier is never accepted)
  unsigned absent = 0;       // TEST ONLY (GR_TEST_COOP_TIMEOUT): the barriers wait for this many workgroups more than the launch has — they time out
  double *recs = nullptr;    // [2][rec_cap] 16-byte flagged records {partial, tag} of coop_exchange (round 6); zero at allocation
  int rec_cap = 0;
  unsigned launch_id = 0;    // unique per launch of a handle: the upper half of the records' tags
};

// A REJECTED STEP DOES NOT STOP THE HEAD (round 5, built-in model).  What a rejection needs of the current point is still in memory — Hcc,
// bc, the scales and the per-point sums are only ever written by this kernel — except the camera-point blocks, which the trial
// linearisation overwrites: those get a second buffer (LmDev::hsel).  So on "not accepted" this launch takes the vertices back
// (k_revert_pack's work), keeps the sums, recomputes what depends on the damping (Hll^-1, M', v) from the stored per-point sums with
// mu * nu, and the rest of the head runs as the first head of the next LM iteration: no four no-op launches, no host round trip, no
// revert launch, no re-linearisation (Ladybug-49: ~50 us per rejected step, every fourth step of the bench line).
template <typename T> struct RejectCont {
  T *cams = nullptr, *pts = nullptr;             // != nullptr: this form is on, and the vertices are the library's (taken back here)
  int model_cont = 0;                            // user-traits problems: this form is on; the vertices are the user's — LmDev::hsel bit 1 tells
                                                 // the step launch behind this head (gr_model_ops.step) to restore its backup first
  const T *cams_bak = nullptr, *pts_bak = nullptr;
  T *pack = nullptr;
  double *vsum = nullptr;                        // [Np][9] per-point sums of the current linearisation, as summed (double)
  LmDev *lm = nullptr;                           // hsel lives here (also for heads that take no decision)
  double nu_cur = 2.0;                           // the damping factor a rejection applies (levenberg_marquardt.hpp:206-207)
};
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(TPB)
k_finalize_schur(int Nc, int Np, int nbc, int scale_system, const int *__restrict__ cam_seg_ptr, const T *__restrict__ cam_partial,
                 const int *__restrict__ pt_ptr, const T *__restrict__ g9, T *__restrict__ Hcc, T *__restrict__ bu, T *__restrict__ Hll,
                 T *__restrict__ scales, double mu, int use_identity, T *__restrict__ Hll_inv, T *__restrict__ Mp, T *__restrict__ vl,
                 LmDecide dec, const unsigned char *__restrict__ cam_fixed, const unsigned char *__restrict__ pt_fixed, CoopState cs,
                 PcgScalars pcg = PcgScalars{}, int pcg_cap = 0 /* > 0: the last workgroup resets the scalars of the per-iteration PCG kernels */,
                 RejectCont<T> rc = RejectCont<T>{}) {
  __shared__ double s_sum[2];
  bool cont = false; // the step was not accepted and this head goes on from the current point
  if (dec.seq) {
    (void)lm_decide_prologue(dec, mu, s_sum);
    if (!dec.report_only) {
      double mun;
      const bool ok = lm_accept<T>(dec, s_sum[0], s_sum[1], mun) && !(cs.fail_dev && *cs.fail_dev);
      cont = !ok && (rc.cams != nullptr || rc.model_cont);
      if (cont) mun = (double)((T)dec.mu_cur * (T)rc.nu_cur);
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        dec.lm->mu = mun; dec.lm->stop = (ok || cont) ? 0 : 2;
        if (dec.dscal) { dec.dscal[0] = s_sum[0]; dec.dscal[1] = s_sum[1]; }
        dec.hres[0] = s_sum[0]; dec.hres[1] = s_sum[1]; dec.hres[2] = mun; dec.hres[3] = ok ? 1.0 : 0.0;
        __threadfence_system();
        *dec.hres_seq = dec.seq;
      }
      if (!ok && !cont) return;
      mu = mun;
    }
  }
  // this launch takes a NEW linearisation over unless it continues from the current point: the other block buffer becomes the current one
  // (nothing in this kernel reads the blocks or hsel)
  if (rc.lm && blockIdx.x == 0 && threadIdx.x == 0) rc.lm->hsel = cont ? (rc.lm->hsel | (rc.model_cont ? 2 : 0)) : ((rc.lm->hsel ^ 1) & 1);
  const int b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0 && cs.barrier) *cs.barrier = 0u;
  if (pcg_cap > 0 && b == (int)gridDim.x - 1) { // an extra workgroup
    for (int i = threadIdx.x; i < pcg_cap * pcg.np; i += TPB) { pcg.rz[i] = 0.0; pcg.den[i] = 0.0; }
    for (int i = threadIdx.x; i < pcg_cap; i += TPB) { pcg.done[i] = 0; pcg.rz0[i] = (i == 0) ? __builtin_inf() : 0.0; }
    if (threadIdx.x == 0) pcg.iters[0] = 0;
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (b < nbc) {
    // cameras: lane j of a 9-lane group sums column j of the camera's 45 + 9 segment partials (as k_finalize_bj)
    const int g = lane / 9, j = lane - 9 * g;
    const int c = (b * 4 + wave) * 7 + g;
    const bool on = g < 7 && c < Nc;
    if (!on) return;
    if (cont && !rc.cams) return; // (user-traits problems: the sums stand, the vertices are the step launch's business)
    if (cont) { // the camera's sums stand; its parameters and pack go back to the backup
      const T v = rc.cams_bak[9 * (size_t)c + j];
      rc.cams[9 * (size_t)c + j] = v;
      T cam[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) cam[i] = __shfl(v, 9 * g + i, 64);
      if (j == 0) {
        T pk[PACK];
        make_campack(cam, pk);
#pragma unroll
        for (int k = 0; k < PACK; ++k) rc.pack[PACK * (size_t)c + k] = pk[k];
      }
      return;
    }
    T a[9], bj = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = T(0);
    int idx[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { const int lo = i < j ? i : j, hi = i < j ? j : i; idx[i] = hi * (hi + 1) / 2 + lo; }
    int sg = cam_seg_ptr[c];
    const int sg1 = cam_seg_ptr[c + 1];
    for (; sg + 4 <= sg1; sg += 4) {
      T q[4][10];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const T *cp = cam_partial + 54 * (size_t)(sg + u);
#pragma unroll
        for (int i = 0; i < 9; ++i) q[u][i] = cp[idx[i]];
        q[u][9] = cp[45 + j];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 9; ++i) a[i] += q[u][i];
        bj += q[u][9];
      }
    }
    if (sg < sg1) {
      const int nleft = sg1 - sg, last = sg1 - 1;
      T q[3][10];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const T *cp = cam_partial + 54 * (size_t)(sg + u < last ? sg + u : last);
#pragma unroll
        for (int i = 0; i < 9; ++i) q[u][i] = cp[idx[i]];
        q[u][9] = cp[45 + j];
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (u < nleft) {
#pragma unroll
          for (int i = 0; i < 9; ++i) a[i] += q[u][i];
          bj += q[u][9];
        }
    }
    const bool fixed = cam_fixed && cam_fixed[c];
    if (fixed) {
      bj = T(0);
#pragma unroll
      for (int i = 0; i < 9; ++i) a[i] = T(0);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) Hcc[81 * (size_t)c + i + 9 * j] = a[i];
    bu[9 * (size_t)c + j] = bj;
    T ajj = a[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) ajj = (i == j) ? a[i] : ajj;
    scales[9 * (size_t)c + j] = (scale_system && !fixed) ? (T)(1.0 / (DBL_EPSILON + sqrt((double)ajj))) : T(1);
    return;
  }
  // points: FIN_PL lanes per point sum its observations' records (fixed order), then k_point_prepare's arithmetic
  using V2 = typename Vec2T<T>::type;
  constexpr int PPB = TPB / FIN_PL;
  const int ntile = (Np + PPB - 1) / PPB;
  const unsigned jl = threadIdx.x % FIN_PL;
  const int npw = (int)gridDim.x - nbc - (pcg_cap > 0 ? 1 : 0); // point workgroups
  for (int tile = b - nbc; tile < ntile; tile += npw) {
    const int l = tile * PPB + (int)(threadIdx.x / FIN_PL);
    const bool on = l < Np;
    double v[9]; // sums, scales, block, inverse, M', v in double whatever T is (see k_finalize_bj): only the outputs are rounded
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = 0.0;
    if (on && !cont) {
      for (int a = pt_ptr[l] + (int)jl; a < pt_ptr[l + 1]; a += FIN_PL) {
        const V2 *gq = reinterpret_cast<const V2 *>(g9 + 8 * (size_t)a);
        const V2 q0 = gq[0], q1 = gq[1], q2 = gq[2], qe = gq[3];
        const double c0x = q0.x, c0y = q0.y, c1x = q1.x, c1y = q1.y, c2x = q2.x, c2y = q2.y, ex = qe.x, ey = qe.y;
        v[0] += c0x * c0x + c0y * c0y;
        v[1] += c0x * c1x + c0y * c1y;
        v[2] += c0x * c2x + c0y * c2y;
        v[3] += c1x * c1x + c1y * c1y;
        v[4] += c1x * c2x + c1y * c2y;
        v[5] += c2x * c2x + c2y * c2y;
        v[6] -= c0x * ex + c0y * ey;
        v[7] -= c1x * ex + c1y * ey;
        v[8] -= c2x * ex + c2y * ey;
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) { v[i] += lane_xor<1>(v[i]); v[i] += lane_xor<2>(v[i]); }
    if (!on || jl != 0) continue;
    if (cont) { // the sums as they were taken, the point back at its backup
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] = rc.vsum[9 * (size_t)l + i];
      if (rc.pts) {
#pragma unroll
        for (int q = 0; q < 3; ++q) rc.pts[3 * (size_t)l + q] = rc.pts_bak[3 * (size_t)l + q];
      }
    } else if (rc.vsum) {
#pragma unroll
      for (int i = 0; i < 9; ++i) rc.vsum[9 * (size_t)l + i] = v[i];
    }
    const bool pfixed = pt_fixed && pt_fixed[l];
    if (pfixed) {
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] = 0.0;
    }
    const bool sc_on = scale_system && !pfixed;
    const double sc[3] = {sc_on ? 1.0 / (DBL_EPSILON + sqrt(v[0])) : 1.0, sc_on ? 1.0 / (DBL_EPSILON + sqrt(v[3])) : 1.0, sc_on ? 1.0 / (DBL_EPSILON + sqrt(v[5])) : 1.0};
    const double H[9] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5]};
    double A[9];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double q = sc[r] * H[r + 3 * c] * sc[c];
        A[r + 3 * c] = (r == c) ? (use_identity ? q + mu : q + mu * clampd(q, 1.0e-6, 1.0e32)) : q;
      }
    spd_inverse<3>(A);
    const size_t t0 = 9 * (size_t)Nc + 3 * (size_t)l;
    double m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { if (pfixed) A[i] = 0.0; Hll[9 * (size_t)l + i] = (T)H[i]; Hll_inv[9 * (size_t)l + i] = (T)A[i]; }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r) { m[r + 3 * c] = sc[r] * A[r + 3 * c] * sc[c]; Mp[9 * (size_t)l + r + 3 * c] = (T)m[r + 3 * c]; }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      bu[t0 + r] = (T)v[6 + r]; scales[t0 + r] = (T)sc[r];
      vl[3 * (size_t)l + r] = (T)(m[r] * v[6] + m[r + 3] * v[7] + m[r + 6] * v[8]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// S^u products + b_S chunk partials in one launch.  Workgroups [0, nwg_items): 4 work items each (k_schur_products' arithmetic and
// LDS strips); multi-item blocks: every item leaves its 81 partial sums in slab[item] (written through), arrives at the block's
// counter, and the LAST arriver adds the slabs in item order and applies the epilogue — fixed order, no float atomics, no zeroing.
// Workgroups [nwg_items, ...): k_bschur_partial's chunks.
template <typename T>
__global__ void __launch_bounds__(TPB, SCHUR_WAVES)
k_schur_reduce(int nitems, int nwg_items, const int *__restrict__ item_blk, const int *__restrict__ item_beg,
               const int *__restrict__ item_end, const int *__restrict__ item_multi /* -1: the block's only item, else its multi-block index */,
               const int *__restrict__ multi_first, const int *__restrict__ multi_n, unsigned *__restrict__ multi_cnt, T *__restrict__ slab,
               const int *__restrict__ prod_a, const int *__restrict__ prod_b, const int *__restrict__ S_rowi, const int *__restrict__ S_coli,
               const int *__restrict__ prod_pm /* landmark of each product */, const T *__restrict__ Hcp, const T *__restrict__ Mp, const T *__restrict__ Hcc,
               const T *__restrict__ scales, double mu, int use_identity, T *__restrict__ S,
               int nch, const int *__restrict__ chunk_beg, const int *__restrict__ pt_cm, const int *__restrict__ pos_cm, const T *__restrict__ vl,
               T *__restrict__ partial9, const LmDev *__restrict__ lm, const T *__restrict__ Hcp_alt = nullptr, const LmDev *__restrict__ hs = nullptr) {
  if (lm) { if (lm->stop) return; mu = lm->mu; }
  if (hs && (hs->hsel & 1)) Hcp = Hcp_alt; // (RejectCont: the buffer that holds the current point's blocks)
  const int lane = threadIdx.x & 63;
  if ((int)blockIdx.x >= nwg_items) { // ---- b_S partials: one wave per camera chunk
    const int ch = ((int)blockIdx.x - nwg_items) * 4 + (threadIdx.x >> 6);
    if (ch >= nch) return;
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
    return;
  }
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= nitems) return;
  const int g = lane / 9, c = lane % 9;
  const int blk = item_blk[item];
  T acc[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) acc[r] = T(0);
  __shared__ T strip[4][7][40];
  if (g < 7) {
    T *sg = strip[threadIdx.x >> 6][g];
    const int q_end = item_end[item];
    int q = item_beg[item] + g;
    // two rounds of look-ahead: the product's three indices (observation a, observation b, landmark — the landmark used to be a
    // dependent load through a's point id) are fetched TWO rounds ahead, its operands one round ahead from indices that are already in
    // registers: a round's loads no longer wait for a chain index -> point id -> operand (Ladybug-49: k_schur_reduce 16.6 us of 8 rounds)
    int a_n = 0, b_n = 0, pm_n = 0;
    T h_n[3], hb_n[3], m_n = T(0);
    auto fetch_idx = [&](int qq) { a_n = prod_a[qq]; b_n = prod_b[qq]; pm_n = prod_pm[qq]; };
    auto fetch = [&]() {
      const T *ha = Hcp + 27 * (size_t)a_n + c, *hb = Hcp + 27 * (size_t)b_n + c;
      h_n[0] = ha[0]; h_n[1] = ha[9]; h_n[2] = ha[18];
      hb_n[0] = hb[0]; hb_n[1] = hb[9]; hb_n[2] = hb[18];
      m_n = Mp[9 * (size_t)pm_n + c];
    };
    if (q < q_end) { fetch_idx(q); fetch(); }
    if (q + 7 < q_end) fetch_idx(q + 7);
    for (; q < q_end; q += 7) {
      sg[c] = h_n[0]; sg[c + 9] = h_n[1]; sg[c + 18] = h_n[2]; sg[27 + c] = m_n;
      const T hb0 = hb_n[0], hb1 = hb_n[1], hb2 = hb_n[2];
      if (q + 7 < q_end) fetch();             // operands of the next round, from the indices fetched a round ago
      if (q + 14 < q_end) fetch_idx(q + 14);  // indices of the round after it
      wave_lds_fence();
      T mv[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) mv[i] = sg[27 + i];
      const T u0 = mv[0] * hb0 + mv[3] * hb1 + mv[6] * hb2;
      const T u1 = mv[1] * hb0 + mv[4] * hb1 + mv[7] * hb2;
      const T u2 = mv[2] * hb0 + mv[5] * hb1 + mv[8] * hb2;
#pragma unroll
      for (int r = 0; r < 9; ++r) acc[r] += sg[r] * u0 + sg[r + 9] * u1 + sg[r + 18] * u2;
      wave_lds_fence();
    }
  }
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    T tot = acc[r];
#pragma unroll
    for (int gg = 1; gg < 7; ++gg) tot += __shfl(acc[r], gg * 9 + c, 64);
    acc[r] = tot;
  }
  const int m = item_multi[item];
  if (m < 0) {
    if (lane < 9) schur_epilogue<T>(S_rowi[blk], S_coli[blk], c, acc, Hcc, scales, mu, use_identity, S + 81 * (size_t)blk + 9 * c);
    return;
  }
  // multi-item block: slab out (written through), arrive; the last arriver finishes the block
  if (lane < 9) {
    T *out = slab + 81 * (size_t)item + 9 * c;
#pragma unroll
    for (int r = 0; r < 9; ++r) __hip_atomic_store(&out[r], acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int last = 0;
  if (lane == 0) {
    const unsigned n = (unsigned)multi_n[m];
    const unsigned old = __hip_atomic_fetch_add(&multi_cnt[m], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == n) { last = 1; __hip_atomic_store(&multi_cnt[m], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  {
    // the slabs in item order, SEVEN in flight — one per lane group, nine registers per lane — and added in item order through
    // shuffles: the heaviest blocks (a camera's own: one product per observation, a dozen items on Ladybug-49) used to cost their last
    // arriver six dependent round trips at two slabs each, the tail of the launch.  (Eight slabs in nine lanes' registers: 72 more
    // VGPRs, Ladybug-1723's launch 266 -> 317 us.)
    T tot[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) tot[r] = T(0);
    const int i0 = multi_first[m], n = multi_n[m];
    for (int it = 0; it < n; it += 7) {
      T q[9];
      const int idx = it + (g < 7 ? g : 0);
      const T *su = slab + 81 * (size_t)(i0 + (idx < n ? idx : n - 1)) + 9 * c;
#pragma unroll
      for (int r = 0; r < 9; ++r) q[r] = __hip_atomic_load(&su[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 7; ++u)
        if (it + u < n) {
#pragma unroll
          for (int r = 0; r < 9; ++r) tot[r] += __shfl(q[r], 9 * u + c, 64);
        }
    }
    if (lane < 9)
    schur_epilogue<T>(S_rowi[blk], S_coli[blk], c, tot, Hcc, scales, mu, use_identity, S + 81 * (size_t)blk + 9 * c);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// grid barrier of the cooperative PCG: monotonic counter, every workgroup one wave.  Everything that crosses workgroups is
// written through (agent-scope stores) and drained before the arrival (cdna_hip_programming.md G16, form R1) and read back with
// agent-scope loads.  Bounded: a launch whose workgroups are not all resident would wait for ever otherwise.
__device__ __forceinline__ bool coop_barrier(unsigned *counter, unsigned target, volatile int *fail, int *fail_dev = nullptr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int ok = 1;
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 200000000ll) { // 2 s of the 100 MHz clock: the launch's workgroups are not all resident
        *fail = 1; // pinned host word: the host takes the step back and leaves the cooperative form (Engine::lm)
        if (fail_dev) *fail_dev = 1;
        __threadfence_system();
        ok = 0;
        break;
      }
    }
  }
  return __builtin_amdgcn_readfirstlane(ok) != 0;
}
// sum of the Nc per-row partials of one phase, the same order in every workgroup (whole wave must call)
__device__ __forceinline__ double coop_sum(const double *part, int Nc) {
  double s = 0;
  for (int i = threadIdx.x & 63; i < Nc; i += 64) s += __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return wave_allsum(s);
}

// Round 6: the barrier AND the sum over the rows as ONE exchange of flagged records (the rendezvous of include/graphite/engine_pose.hpp):
// a workgroup stores {its partial, tag} with one 16-byte write-through store, then polls the records of all workgroups until each carries
// the tag (launch << 32 | phase); the sum runs in the order of coop_sum (lane-strided, then the butterfly): the same bits as the counter
// barrier + partial reads it replaces, one store -> load hop instead of store -> counter -> poll -> partial reads.  DRAIN: the phase
// published p (write-through stores that must be acknowledged before the record announces them).  Bounded like coop_barrier.
typedef unsigned sf_u32x4 __attribute__((ext_vector_type(4)));
template <bool DRAIN> __device__ __forceinline__ bool coop_exchange(const CoopState &cs, unsigned phase, int i, int Nc, double mine, double &total) {
  const int lane = threadIdx.x & 63, n = Nc + (int)cs.absent, cap = cs.rec_cap;
  const unsigned long long tag = ((unsigned long long)cs.launch_id << 32) | phase;
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(cs.recs, 0, 2 * cap * 16, 0x00020000);
  const int set = (int)(phase & 1u) * cap;
  if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
    // (the second half is tag XOR the value's bits: halves of different stores — should a 16-byte access ever be split — carry no valid tag)
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, mine), mark = tag ^ bits;
    sf_u32x4 a;
    a.x = (unsigned)bits; a.y = (unsigned)(bits >> 32); a.z = (unsigned)mark; a.w = (unsigned)(mark >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(a, rr, (set + i) * 16, 0, 16);
  }
  double acc = 0;
  bool ok = true;
  const int nper = (n + 63) / 64;
  for (int c0 = 0; c0 < nper && ok; c0 += 4) {
    sf_u32x4 rec[4];
    unsigned pend = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (c0 + u < nper && lane + 64 * (c0 + u) < n) pend |= 1u << u;
    const long long t0 = wall_clock64();
    while (true) {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (pend & (1u << u)) rec[u] = __builtin_amdgcn_raw_buffer_load_b128(rr, (set + lane + 64 * (c0 + u)) * 16, 0, 16);
#pragma unroll
      for (int u = 0; u < 4; ++u) if ((pend & (1u << u)) && ((((unsigned long long)rec[u].w << 32) | rec[u].z) ^ (((unsigned long long)rec[u].y << 32) | rec[u].x)) == tag) pend &= ~(1u << u);
      if (!__any(pend != 0)) break;
      if (wall_clock64() - t0 > 200000000ll) { ok = false; break; } // 2 s of the 100 MHz clock: the launch's workgroups are not all resident
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) break;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c0 + u < nper && lane + 64 * (c0 + u) < Nc) acc += __builtin_bit_cast(double, ((unsigned long long)rec[u].y << 32) | rec[u].x);
  }
  if (!ok) {
    if (lane == 0) { *cs.fail = 1; if (cs.fail_dev) *cs.fail_dev = 1; __threadfence_system(); }
    return false;
  }
  total = wave_allsum(acc);
  return true;
}

// PCGSchurSolver::solve (solver/pcg_schur.hpp:79-168) + BlockJacobiSchurPreconditioner (block_jacobi_schur.hpp:114-178), all of
// it in ONE launch: grid = Nc workgroups of one wave, workgroup i owns block row i of S.  Lanes 0..8 hold the row's x, r, z, p,
// x_backup entries in registers for the whole loop; p and the dot-product partials cross workgroups through device memory.
// The arithmetic per entry is that of k_schur_pcg_prepare / k_schur_matvec / k_pcgs_update / k_pcgs_direction; the dot products
// are summed per row first, then over the rows in row order.
template <typename T>
__global__ void __launch_bounds__(64)
k_schur_pcg_coop(int Nc, const int *__restrict__ row_ptr, const int *__restrict__ row_blk, const int *__restrict__ row_col,
                 const T *__restrict__ S, const int *__restrict__ diag_blk, const int *__restrict__ cam_chunk_ptr,
                 const T *__restrict__ partial9, const T *__restrict__ bc, const T *__restrict__ scales, T *__restrict__ b_schur,
                 T *__restrict__ p_glob, T *__restrict__ x_out, int max_iter, double tol, double rejection_ratio, CoopState cs,
                 const LmDev *__restrict__ lm) {
  if (lm && lm->stop) return;
  const int i = blockIdx.x, lane = threadIdx.x;
  const int g = lane / 9, r = lane % 9;
  const bool own = lane < 9;
  unsigned phase = 0;
  if (i == 0 && lane == 0 && cs.ts) cs.ts[0] = wall_clock64();
  // ---- b_S row, block-Jacobi inverse of the diagonal block (Gauss-Jordan across the 9 lanes, as k_finalize_bj), start of the loop
  T b = T(0);
  if (own) {
    for (int ch = cam_chunk_ptr[i]; ch < cam_chunk_ptr[i + 1]; ++ch) b += partial9[9 * (size_t)ch + r];
    b = scales[9 * (size_t)i + r] * (bc[9 * (size_t)i + r] - b);
    b_schur[9 * (size_t)i + r] = b;
  }
  // The row's blocks do not change over the inner iterations, and the matvec used to be a chain per round — block index -> block ->
  // its 9 entries, 7 rounds of it per iteration on a 49-block row (Ladybug-49: ~7 us of an 18 us launch that runs ONE inner iteration on
  // average).  The first CR rounds of every lane group are fetched ONCE, here, into registers (entry (r, c) of the block as the product
  // uses it); the loads fly under the 9 x 9 inversion below.  An iteration then only fetches p.  Same products in the same order.
  constexpr int CR = 7;
  T cS[CR][9];
  int ccol[CR];
  const int row_beg = row_ptr[i], row_end = row_ptr[i + 1];
#pragma unroll
  for (int q = 0; q < CR; ++q) {
    const int e = row_beg + g + 7 * q;
    ccol[q] = -1;
#pragma unroll
    for (int c = 0; c < 9; ++c) cS[q][c] = T(0);
    if (g < 7 && e < row_end) {
      const int blk = row_blk[e];
      ccol[q] = row_col[e];
      const bool transposed = blk < 0;
      const T *Ab = S + 81 * (size_t)(transposed ? ~blk : blk);
#pragma unroll
      for (int c = 0; c < 9; ++c) cS[q][c] = transposed ? Ab[c + 9 * r] : Ab[r + 9 * c];
    }
  }
  double A[9]; // column r of the diagonal block in lane r
  {
    const T *B = S + 81 * (size_t)diag_blk[i];
#pragma unroll
    for (int q = 0; q < 9; ++q) A[q] = own ? (double)B[q + 9 * r] : (q == r % 9 ? 1.0 : 0.0);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      double f[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) f[q] = __shfl(A[q], k, 64);
      const double piv = 1.0 / f[k];
      if (own) {
        A[k] = (r == k) ? piv : A[k] * piv;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          if (q == k) continue;
          A[q] = (r == k) ? -f[q] * piv : A[q] - f[q] * A[k];
        }
      }
    }
  }
  // z = Minv r needs ROW r of the inverse in lane r (the inverse of a symmetric block is symmetric only up to rounding; the row
  // keeps the bits of spd_inverse<9> / k_schur_pcg_prepare): transposed across the lanes once, Mrow[q] = Minv[r][q] = entry r of lane q's column
  T Mrow[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    double mrq = 0.0;
#pragma unroll
    for (int e = 0; e < 9; ++e) { const double t = __shfl(A[e], q, 64); mrq = (e == r) ? t : mrq; }
    Mrow[q] = (T)mrq;
  }
  auto apply_minv = [&](T v) -> T {
    T s = T(0);
#pragma unroll
    for (int q = 0; q < 9; ++q) s += Mrow[q] * __shfl(v, q, 64);
    return s;
  };
  T xr = T(0), rr = b, zr = apply_minv(b), pr = zr, xbr = T(0);
  double rz = 0.0;
  {
    const double d = wave_allsum(own ? (double)(rr * zr) : 0.0);
    if (own) __hip_atomic_store(&p_glob[9 * (size_t)i + r], pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!coop_exchange<true>(cs, ++phase, i, Nc, d, rz)) return;
  }
  double rz0 = __builtin_inf();
  int k = 0, iters = 0;
  for (; k < max_iter; ++k) {
    if (rz == 0.0) break;
    // y_i = (S p)_i : 7 groups of 9 lanes stride over the row's block list (k_schur_matvec)
    T acc = T(0);
    if (g < 7) {
      T xq[CR][9];
#pragma unroll
      for (int q = 0; q < CR; ++q)
#pragma unroll
        for (int c = 0; c < 9; ++c) xq[q][c] = ccol[q] >= 0 ? __hip_atomic_load(&p_glob[9 * (size_t)ccol[q] + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T(0);
#pragma unroll
      for (int q = 0; q < CR; ++q)
        if (ccol[q] >= 0) {
#pragma unroll
          for (int c = 0; c < 9; ++c) acc += cS[q][c] * xq[q][c];
        }
      for (int e = row_beg + g + 7 * CR; e < row_end; e += 7) { // rows of more than 49 blocks: the rest as before
        const int blk = row_blk[e], j = row_col[e];
        const bool transposed = blk < 0;
        const T *Ab = S + 81 * (size_t)(transposed ? ~blk : blk);
        T xj[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) xj[c] = __hip_atomic_load(&p_glob[9 * (size_t)j + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!transposed) {
#pragma unroll
          for (int c = 0; c < 9; ++c) acc += Ab[r + 9 * c] * xj[c];
        } else {
#pragma unroll
          for (int c = 0; c < 9; ++c) acc += Ab[c + 9 * r] * xj[c];
        }
      }
    }
    T y = acc;
#pragma unroll
    for (int gg = 1; gg < 7; ++gg) y += __shfl(acc, gg * 9 + r, 64);
    double den = 0.0;
    {
      const double d = wave_allsum(own ? (double)(y * pr) : 0.0);
      if (!coop_exchange<false>(cs, ++phase, i, Nc, d, den)) return;
    }
    if (den == 0.0 || den != den) break;
    ++iters;
    const T alpha = (T)rz / (T)den;
    xbr = xr;
    xr = alpha * pr + xr;
    rr = -alpha * y + rr;
    zr = apply_minv(rr);
    double rz_sum = 0.0;
    {
      const double d = wave_allsum(own ? (double)(rr * zr) : 0.0);
      if (!coop_exchange<false>(cs, ++phase, i, Nc, d, rz_sum)) return;
    }
    const T rz_new = (T)rz_sum;
    const bool reject = (fabs((double)rz_new) > rejection_ratio * rz0) || (rz_new != rz_new);
    if (reject) { xr = xbr; break; }
    const T beta = rz_new / (T)rz;
    pr = beta * pr + zr;
    rz0 = fmin(rz0, fabs((double)rz_new));
    rz = (double)rz_new; // pcg_schur.hpp keeps rz in T
    if (fabs((double)rz_new) < tol) { ++k; break; }
    if (own) __hip_atomic_store(&p_glob[9 * (size_t)i + r], pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    { double unused; if (!coop_exchange<true>(cs, ++phase, i, Nc, 0.0, unused)) return; } // every row's p is published before any row reads it
  }
  if (own) x_out[9 * (size_t)i + r] = xr;
  if (i == 0 && lane == 0) {
    *cs.iters = iters;
    if (cs.hiters) *cs.hiters = iters;
    if (cs.ts) cs.ts[1] = wall_clock64();
    __threadfence_system();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// SchurComplement::compute_landmark_update (schur.hpp:279-302) + the trial step (graph.hpp:292-309, ops/update.hpp:11-31) +
// compute_rho's denominator partials (levenberg_marquardt.hpp:34-41) + the camera packs: workgroups [0, nct) take 28 cameras each,
// the others points, FIN_PL lanes per point over its observations (point-major order: a point's Hcp blocks are contiguous).
// APPLY = false (user-traits problems): the landmark part of the step only — the trial step is the user's Traits::update
// (gr_model_ops.step, the launch behind this one)
template <typename T, bool APPLY = true>
__global__ void __launch_bounds__(TPB)
k_backsub_apply(int Nc, int Np, int nct, const int *__restrict__ pt_ptr, const int *__restrict__ cam_pm, const T *__restrict__ Hcp,
                const T *__restrict__ Hll_inv, const T *__restrict__ bu, const T *__restrict__ scales, T *__restrict__ x /* [9 Nc + 3 Np]: xp in, xl out */,
                T *__restrict__ cams, T *__restrict__ pts, T *__restrict__ cams_bak, T *__restrict__ pts_bak, T *__restrict__ pack,
                double mu, double *__restrict__ rho_partial, const LmDev *__restrict__ lm, const T *__restrict__ Hcp_alt = nullptr, const LmDev *__restrict__ hs = nullptr) {
  if (lm) { if (lm->stop) return; mu = lm->mu; }
  if (hs && (hs->hsel & 1)) Hcp = Hcp_alt;
  __shared__ double red[4];
  __shared__ T cs[252];
  const unsigned pose_dim = 9u * (unsigned)Nc;
  double rho = 0;
  if ((int)blockIdx.x < nct) {
    if (!APPLY) return;
    const unsigned i = blockIdx.x * 252u + threadIdx.x;
    if (threadIdx.x < 252 && i < pose_dim) {
      const T d = x[i], s = scales[i], xo = cams[i];
      cams_bak[i] = xo;
      const T xn = xo + d * s;
      cams[i] = xn;
      cs[threadIdx.x] = xn;
      rho = (double)(d * ((T)mu * d + s * bu[i]));
    }
    __syncthreads();
    const unsigned c = blockIdx.x * 28u + threadIdx.x;
    if (threadIdx.x < 28 && 9u * c < pose_dim) {
      T cam[9], pk[PACK];
#pragma unroll
      for (int k = 0; k < 9; ++k) cam[k] = cs[9 * threadIdx.x + k];
      make_campack(cam, pk);
#pragma unroll
      for (int k = 0; k < PACK; ++k) pack[PACK * (size_t)c + k] = pk[k];
    }
  } else {
    constexpr int PPB = TPB / FIN_PL;
    const int ntile = (Np + PPB - 1) / PPB;
    const unsigned jl = threadIdx.x % FIN_PL;
    for (int tile = (int)blockIdx.x - nct; tile < ntile; tile += (int)gridDim.x - nct) {
      const int l = tile * PPB + (int)(threadIdx.x / FIN_PL);
      const bool on = l < Np;
      T v[3] = {T(0), T(0), T(0)};
      if (on) {
        for (int a = pt_ptr[l] + (int)jl; a < pt_ptr[l + 1]; a += FIN_PL) {
          const int c = cam_pm[a];
          const T *h = Hcp + 27 * (size_t)a;
#pragma unroll
          for (int r = 0; r < 9; ++r) {
            const T xs = scales[9 * (size_t)c + r] * x[9 * (size_t)c + r];
            v[0] += h[r] * xs; v[1] += h[r + 9] * xs; v[2] += h[r + 18] * xs;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) { v[q] += lane_xor<1>(v[q]); v[q] += lane_xor<2>(v[q]); }
      if (!on || jl != 0) continue;
      const size_t t0 = (size_t)pose_dim + 3 * (size_t)l;
      const T s0 = scales[t0], s1 = scales[t0 + 1], s2 = scales[t0 + 2];
      const T r0 = s0 * (bu[t0] - v[0]), r1 = s1 * (bu[t0 + 1] - v[1]), r2 = s2 * (bu[t0 + 2] - v[2]);
      const T *inv = Hll_inv + 9 * (size_t)l;
      const T sv[3] = {s0, s1, s2};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const T d = inv[q] * r0 + inv[q + 3] * r1 + inv[q + 6] * r2;
        x[t0 + q] = d;
        if (!APPLY) continue;
        const T xo = pts[3 * (size_t)l + q];
        pts_bak[3 * (size_t)l + q] = xo;
        pts[3 * (size_t)l + q] = xo + d * sv[q];
        rho += (double)(d * ((T)mu * d + sv[q] * bu[t0 + q]));
      }
    }
  }
  if (!APPLY) return;
  rho = block_sum_256(rho, red);
  if (threadIdx.x == 0) rho_partial[blockIdx.x] = rho;
}

} // namespace gr
