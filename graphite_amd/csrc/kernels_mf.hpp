// Linearisation, chi2 and the matrix-free PCG (PCGSolver) kernels — gfx950.
//
// All per-observation work runs in CAMERA-major ("cm") order, ONE observation per
// lane, flat indexing: the four per-observation streams (camera id, point id,
// pm position, observation) are coalesced and independent, so a lane's
// dependent-load chain is two deep (indices -> pack/point gather -> math).
// Neighbouring lanes share the camera, so the 192-byte camera pack costs one
// cache line per wave instead of 64 (in point-major order it was the
// bottleneck), and camera-side sums are wave reductions.  A wave that straddles
// a camera boundary simply reduces once per distinct camera.
//   segment = (wave, camera) pair; seg id = cam_seg_ptr[c] + (wave - (cam_ptr[c] >> 6))
// Point-side sums: each lane stores its contribution at the observation's pm
// position; a point's contributions are then consecutive and are summed in
// fixed order by the consumer kernel (no atomics, bitwise reproducible).
//
// Dot products: every block publishes its partial sums, the LAST block to finish
// (ticket) adds them in fixed order and writes the PCG control record of the
// iteration, so consumers read one small struct instead of re-reducing, and the
// host can stop enqueueing iterations as soon as the loop has left (pinned flag).
#pragma once
#include "kernels.hpp"
#include "comm.hpp"

namespace gr {
#ifndef LIN_WAVES
#define LIN_WAVES 3
#endif
#ifndef OP_WAVES
#define OP_WAVES 4
#endif
#ifndef FIN_PL
#ifndef UPD_VAR
#define UPD_VAR 0 // diagnostic builds only: ablations of the update kernel's point part
#endif
#define FIN_PL 4 // lanes per point in the finalize kernel (power of two; TPB and 90 Nc keep groups inside a wave)
#endif

constexpr int TICKET_GROUPS = 64;

// PCG scalars never visit the host.  Iteration k owns record k of `acc` ([cap][5][NS] doubles):
//   RZP r.z' and RR r.r at the START of iteration k (z' = Minv r; the reference's r.z with
//       z = Minv (r/||r||) is rzp / sqrt(rr)),
//   PDZ p.D.z' and ZDZ z'.D.z' (D = clamped diagonal or I) of the same moment, from which the
//       direction kernel gets p.D.p of the NEW direction by recurrence — no reduction of its own,
//   DEN sum_obs rho'|J ps|^2 of iteration k
//       — NS partial sums each, filled with fire-and-forget atomics and re-summed by every
//       consumer wave (a last-block/ticket reduction added ~10 us of tail to every kernel);
//   pdp[k] p.D.p of the direction used by iteration k, rz0[k] running min of |rz_new| before
//   iteration k, done[k] loop left before iteration k — written by thread 0 of the direction
//   kernel of iteration k-1.
// Multi-GPU: every record is summed over ranks (RCCL) between the producing and the consuming
// kernel; camera-space vectors are replicated, so their dot-product share is added by rank 0 only
// (cam_weight).
enum { RZP = 0, RR = 1, PDZ = 2, ZDZ = 3, DEN = 4, NSLOT = 5 };
#ifndef GR_BJ_WAVES
#define GR_BJ_WAVES 1
#endif
struct PcgState {
  double *acc;          // [cap][NSLOT][NSW]
  double *pdp, *rz0;    // [cap]
  int *done;            // [cap]
  int *iters;           // [1]
  volatile int *hflag;  // pinned host memory [cap]: 1 = iteration finished, 2 = loop left
  volatile int *hiters; // pinned host mirror of iters
  int *left;            // [1] 1 once the loop has left (set by the direction kernel, cleared by k_block_jacobi / the init kernel):
                        // the gate of the trial-step kernels that the host enqueues BEFORE it has seen the exit flag
  // LAZY direction (lazy != 0): the direction kernel only takes the loop decision and publishes beta[k], scale[k]; the
  // direction p_k = beta p_{k-1} + scale z'_k is formed where it is used — by the operator from the pre-scaled vectors
  // ps = s.*p_{k-1} and zs = s.*z'_k as it gathers them, and by the update kernel, which also stores p_k, ps, zs.  One
  // pass over five n-vectors per inner iteration less.
  double *beta, *scale; // [cap]
  void *ps, *zs;        // T[n]
  void *zrec = nullptr; // single-reduction form with point records: [X Y Z | zs_x zs_y zs_z . .] per point, the zs half written by whoever writes zs
  int lazy;             // 0 direction kernel, 1 lazy direction, 2 single-reduction recurrence (below)
  // SINGLE-REDUCTION form (lazy == 2; Chronopoulos-Gear, the variant oracle/bal_pipeline.hpp::solve_pcg_cg documents): the
  // operator is applied to the preconditioned residual u = z'/|r| instead of the direction, s = A p follows the recurrence
  // s_k = A u_k + beta_k s_{k-1}, and alpha_k = gamma_k / (delta_k - beta_k gamma_k / alpha_{k-1}).  All dot products of an
  // iteration (r.z', r.r, z'.D.z' of record k and the operator's DEN of record k) are needed at ONE point, the start of
  // update k, so landmark shards all-reduce them in one message together with the operator's camera rows.
  // beta[k] then holds alpha_k and scale[k] holds gamma_k.
  void *sv;             // T[n]: s = A p
  // the loop decision of the lazy form is taken in the prologue of the NEXT operator launch (pcg_decide)
  void *x, *xb;         // T[n]: solution and its backup (a rejected step restores x)
  long long *ts = nullptr; // pinned (LM loop, fused form): [0] device wall clock when the PCG loop starts, [1] when it has ended
  int ts_op = 0;           // the loop starts with the operator of iteration 0 (no first direction launch): it takes stamp [0]
  // ... and in that form k_finalize_bj leaves the three dots of the PCG start (r.r, r.z', z'.D.z') as one partial per workgroup
  // here ([3][n_part0]) instead of adding them to the slots with atomics at its very end (5 us of tail: 3 x 1 086 atomics that
  // all workgroups issue at the same moment); the LAST workgroup of the operator launch of iteration 0 — which does not need
  // them — adds them in workgroup order into slot 0 while the others already work: off the critical path, and reproducible
  double *part0 = nullptr;
  int n_part0 = 0;
  // resident PCG (kernels_rp.hpp): the words of its grid barrier, cleared by the launch in front of it (k_finalize_bj's first workgroup)
  unsigned *bar = nullptr;
  int bar_words = 0;
  unsigned n;
  double tol, rej;
  __device__ __forceinline__ double *slots(int k, int which) const { return acc + ((size_t)k * NSLOT + which) * NSW; }
};

// gate != nullptr (user-traits problems, the trial step's "loop state is spent" reset): enqueued ahead of the PCG exit flag — nothing
// to do while the loop has not left; the gate word itself (st.left) is then left alone, the kernels behind this one test it too
__global__ void k_pcg_state_init(PcgState st, int cap, const int *__restrict__ gate = nullptr) {
  if (gate && !*gate) return;
  for (int i = threadIdx.x; i < cap * NSLOT * NS; i += blockDim.x) st.acc[slot_word(i)] = 0.0;
  for (int i = threadIdx.x; i < cap; i += blockDim.x) { st.done[i] = 0; st.pdp[i] = 0.0; st.rz0[i] = (i == 0) ? __builtin_inf() : 0.0; }
  if (threadIdx.x == 0) { st.iters[0] = 0; if (st.left && !gate) *st.left = 0; }
}

// BlockJacobiPreconditioner::set_damping_factor (block_jacobi.hpp:120-172) for cameras AND points in one
// launch of 64-thread blocks ([0, nbc) camera blocks, then point blocks); with st.acc != nullptr the last
// block also resets the PCG loop state, so a solve starts without its own init launch.
template <typename T>
__global__ void __launch_bounds__(64, GR_BJ_WAVES)
k_block_jacobi(int Nc, int Np, int nbc, int nbp, const T *__restrict__ Hcc, const T *__restrict__ Hll,
               const T *__restrict__ scales, double mu, int use_identity, T *__restrict__ MinvC,
               T *__restrict__ MinvP, T *__restrict__ diag_clamped, PcgState st, int cap,
               const LmDev *__restrict__ lm = nullptr,
               // start of the PCG loop fused in (the loop state must already be clean): r = s .* b^u, x = 0,
               // z' = Minv r (identity_precond: z' = r), record 0 of the dots  (k_pcg_update MODE 0)
               const T *__restrict__ bu = nullptr, T *__restrict__ x = nullptr, T *__restrict__ r = nullptr,
               T *__restrict__ zt = nullptr, int identity_precond = 0,
               int cam_weight = 1 /* landmark shards: the replicated camera part of the dots counts on rank 0 only */) {
  if (lm) { if (lm->stop) return; mu = lm->mu; }
  const int b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0 && st.left) *st.left = 0; // a new loop starts
  if (b == 0 && st.bar) for (int i = threadIdx.x; i < st.bar_words; i += TPB) st.bar[i] = 0u;
  double prr = 0, prz = 0, pzz = 0;
  if (b < nbc) {
    const int c = b * 64 + threadIdx.x;
    if (c < Nc) {
      double A[81];
      T dcl[9];
      const T *B = Hcc + 81 * (size_t)c;
      const T *s = scales + 9 * (size_t)c;
      double sh[9]; // the scaled, damped block is formed in double from the stored sums (scale_hat): its inverse sees no fp32 rounding of its own
#pragma unroll
      for (int i = 0; i < 9; ++i) sh[i] = scale_hat(s[i], B[10 * i]);
#pragma unroll
      for (int col = 0; col < 9; ++col)
#pragma unroll
        for (int rw = 0; rw < 9; ++rw) {
          const double v = sh[rw] * (double)B[rw + 9 * col] * sh[col];
          if (rw == col) {
            A[rw + 9 * col] = use_identity ? v + mu : v + mu * clampd(v, 1.0e-6, 1.0e32);
            dcl[rw] = (T)clampd(v, 1.0e-6, 1.0e32);
            diag_clamped[9 * (size_t)c + rw] = dcl[rw];
          } else A[rw + 9 * col] = v;
        }
      spd_inverse<9>(A);
#pragma unroll
      for (int i = 0; i < 81; ++i) MinvC[81 * (size_t)c + i] = (T)A[i];
      if (x) {
        T rv[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) rv[i] = s[i] * bu[9 * (size_t)c + i];
#pragma unroll
        for (int row = 0; row < 9; ++row) {
          T z = 0;
          if (identity_precond) z = rv[row];
          else {
#pragma unroll
            for (int q = 0; q < 9; ++q) z += (T)A[row + 9 * q] * rv[q];
          }
          const size_t t = 9 * (size_t)c + row;
          x[t] = T(0); r[t] = rv[row]; zt[t] = z;
          if (st.lazy) static_cast<T *>(st.zs)[t] = s[row] * z;
          const T d = use_identity ? T(1) : dcl[row];
          if (cam_weight) { prr += (double)(rv[row] * rv[row]); prz += (double)(rv[row] * z); pzz += (double)(d * z * z); }
        }
      }
    }
  } else if (b < nbc + nbp) {
    const int l = (b - nbc) * 64 + threadIdx.x;
    if (l < Np) {
      const size_t t0 = 9 * (size_t)Nc + 3 * (size_t)l;
      const T *s = scales + t0;
      const T *H = Hll + 9 * (size_t)l;
      double A[9];
      T dcl[3];
      const double sh[3] = {scale_hat(s[0], H[0]), scale_hat(s[1], H[4]), scale_hat(s[2], H[8])};
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rw = 0; rw < 3; ++rw) {
          const double v = sh[rw] * (double)H[rw + 3 * c] * sh[c];
          if (rw == c) {
            A[rw + 3 * c] = use_identity ? v + mu : v + mu * clampd(v, 1.0e-6, 1.0e32);
            dcl[rw] = (T)clampd(v, 1.0e-6, 1.0e32);
            diag_clamped[t0 + rw] = dcl[rw];
          } else A[rw + 3 * c] = v;
        }
      spd_inverse<3>(A);
#pragma unroll
      for (int i = 0; i < 9; ++i) MinvP[9 * (size_t)l + i] = (T)A[i];
      if (x) {
        T rv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = s[i] * bu[t0 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const T z = identity_precond ? rv[i] : (T)A[i] * rv[0] + (T)A[i + 3] * rv[1] + (T)A[i + 6] * rv[2];
          x[t0 + i] = T(0); r[t0 + i] = rv[i]; zt[t0 + i] = z;
          if (st.lazy) static_cast<T *>(st.zs)[t0 + i] = s[i] * z;
          if (st.zrec) static_cast<T *>(st.zrec)[8 * (size_t)l + 3 + i] = s[i] * z;
          const T d = use_identity ? T(1) : dcl[i];
          prr += (double)(rv[i] * rv[i]); prz += (double)(rv[i] * z); pzz += (double)(d * z * z);
        }
      }
    }
  } else if (st.acc) {
    for (int i = threadIdx.x; i < cap * NSLOT * NS; i += 64) st.acc[slot_word(i)] = 0.0;
    for (int i = threadIdx.x; i < cap; i += 64) { st.done[i] = 0; st.pdp[i] = 0.0; st.rz0[i] = (i == 0) ? __builtin_inf() : 0.0; }
    if (threadIdx.x == 0) st.iters[0] = 0;
    return;
  }
  if (x) { // one wave per block
    prr = wave_sum(prr); prz = wave_sum(prz); pzz = wave_sum(pzz);
    if (threadIdx.x == 0) { slot_add(st.slots(0, RR), 0, prr); slot_add(st.slots(0, RZP), 0, prz); slot_add(st.slots(0, ZDZ), 0, pzz); }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LM loop, matrix-free PCG: ONE launch between the linearisation kernel and the first PCG iteration.
//   (1) optional: the accept decision of the trial step this linearisation evaluated (levenberg_marquardt.hpp:184-197) —
//       every workgroup sums the chi2 / rho-denominator block partials in the same fixed order, derives rho and the new
//       damping in T like the host loop, and returns at once when the step is not accepted; workgroup 0 publishes
//       (chi2, denominator, new damping, accepted) to pinned host memory and to LmDev for the kernels enqueued behind it;
//   (2) k_linearize_finalize: fixed-order sums of the (wave, camera) segment partials -> Hcc^u, bc^u, camera scales; per
//       point the sums of its observations' records -> Hll^u, bl^u, point scales (graph.hpp:254-270);
//   (3) k_block_jacobi: damped, scaled diagonal blocks inverted (block_jacobi.hpp:120-172), clamped diagonal;
//   (4) the PCG start r = s .* b^u, x = 0, z' = Minv r and record 0 of the dots (pcg.hpp:108-127).
// The blocks (2) writes and (3) re-reads never leave the registers; (2) + (3) were 16 + 18 us as two launches on
// Ladybug-1723.  Cameras: one lane per COLUMN of a camera's 9 x 9 block (7 cameras per wave, 28 per workgroup): the
// Gauss-Jordan inverse runs across the 9 lanes with row broadcasts instead of serially in one thread (same operations per
// entry, same order: same bits as spd_inverse<9>).  Points: FIN_PL lanes per point as in k_linearize_finalize.
struct LmDecide {
  int seq = 0;                                  // != 0: decide; the value published to hres_seq
  int report_only = 0;                          // with seq: only publish the chi2 sum (first linearisation of an LM call), no decision
  double chi2_cur = 0, mu_cur = 0;              // chi2 and damping of the iteration whose trial step is being judged
  const double *chi2_partial = nullptr; int n_chi2 = 0; // block partials of the trial linearisation's chi2 (k_linearize) ...
  const double *rho_partial = nullptr; int n_rho = 0;   // ... and of the rho denominator (ApplyOnExit)
  volatile double *hres = nullptr;              // pinned: [0] trial chi2, [1] rho denominator, [2] new damping, [3] accepted
  volatile int *hres_seq = nullptr;
  LmDev *lm = nullptr;
  double *dscal = nullptr;                      // device copy of [0], [1]
};
#ifndef FBJ_WAVES
#define FBJ_WAVES 4 // waves per SIMD the kernel is compiled for (130 VGPRs uncapped: 3 workgroups per CU of a grid sized for 4)
#endif
template <typename T, int VAR = 0> // VAR (diagnostic timing only): 1 few output stores, 2 no inverse / scale math, 4 no record loads, 8 no dot-product atomics
__global__ void __launch_bounds__(TPB, FBJ_WAVES)
k_finalize_bj(int Nc, int Np, int nbc, int scale_system, const int *__restrict__ cam_seg_ptr, const T *__restrict__ cam_partial,
              const int *__restrict__ pt_ptr, const T *__restrict__ g9, T *__restrict__ Hcc, T *__restrict__ bu,
              T *__restrict__ Hll, T *__restrict__ scales, double mu, int use_identity, T *__restrict__ MinvC,
              T *__restrict__ MinvP, T *__restrict__ diag_clamped, PcgState st, T *__restrict__ x, T *__restrict__ r,
              T *__restrict__ zt, T *__restrict__ zs /* != nullptr: s .* z' instead of x = 0 (lazy first PCG iteration) */,
              int identity_precond, int cam_weight, LmDecide dec,
              const unsigned char *__restrict__ cam_fixed, const unsigned char *__restrict__ pt_fixed) {
  __shared__ double sA[4][7 * 81];
  __shared__ double red3[4][3];
  if (dec.seq) {
    // wave 0 of every workgroup: lane L adds partials L, L + 64, ... in order (eight loads in flight at a time), then a
    // butterfly; the two sums reach the other waves through LDS.  Same fixed order in every workgroup: same decision.
    // (Measured alternatives: a 256-thread block reduction of each array 3-4 us per workgroup; every wave summing for itself
    // 10-15 us; 64 group partials folded by a ticket in the producers 2.7 us of tail in k_linearize + 1.5 us in the direction launch.)
    __shared__ double s_sum[2];
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
    } else {
    const T chi2v = (T)dec.chi2_cur, new_chi2 = (T)cs;
    const T denom = (T)rs + (T)1.0e-3;
    const T rho = (chi2v - new_chi2) / denom;
    const bool ok = isfinite((double)new_chi2) && rho > T(0);
    double alpha = 1.0 - pow(2.0 * (double)rho - 1.0, 3.0);
    alpha = fmax(fmin(alpha, 2.0 / 3.0), 1.0 / 3.0);
    const T mun = (T)dec.mu_cur * (T)alpha;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      dec.lm->mu = (double)mun; dec.lm->stop = ok ? 0 : 2;
      if (dec.dscal) { dec.dscal[0] = cs; dec.dscal[1] = rs; }
      dec.hres[0] = cs; dec.hres[1] = rs; dec.hres[2] = (double)mun; dec.hres[3] = ok ? 1.0 : 0.0;
      __threadfence_system();
      *dec.hres_seq = dec.seq;
    }
    if (!ok) return;
    mu = (double)mun;
    }
  }
  const int b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0 && st.left) *st.left = 0; // a new loop starts
  if (b == 0 && st.bar) for (int i = threadIdx.x; i < st.bar_words; i += TPB) st.bar[i] = 0u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double prr = 0, prz = 0, pzz = 0;
  if (b < nbc) {
    const int g = lane / 9, j = lane - 9 * g;
    const int c = (b * 4 + wave) * 7 + g;
    const bool on = g < 7 && c < Nc;
    const int base = g < 7 ? 9 * g : 54; // lane 63 idles on group 6's lanes
    T a[9], bj = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = T(0);
    bool fixed = false;
    if (on) {
      int idx[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) { const int lo = i < j ? i : j, hi = i < j ? j : i; idx[i] = hi * (hi + 1) / 2 + lo; }
      // four segments of loads in flight, added in segment order (the sums are what k_linearize_finalize forms)
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
      if (sg < sg1) { // 1-3 segments left: one batch with clamped indices instead of a dependent round trip per segment
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
      fixed = cam_fixed && cam_fixed[c];
      if (fixed) {
        bj = T(0);
#pragma unroll
        for (int i = 0; i < 9; ++i) a[i] = T(0);
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) Hcc[81 * (size_t)c + i + 9 * j] = a[i];
      bu[9 * (size_t)c + j] = bj;
    }
    T ajj = a[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) ajj = (i == j) ? a[i] : ajj;
    const T sj = (on && scale_system && !fixed) ? (T)(1.0 / (DBL_EPSILON + sqrt((double)ajj))) : T(1);
    if (on) scales[9 * (size_t)c + j] = sj;
    // damped, scaled block: column j of A in this lane
    double A[9];
    T dclj = T(1);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const T si = __shfl(sj, base + i, 64);
      const T v = si * a[i] * sj;
      if (i == j) { A[i] = (double)damp_diag(v, mu, use_identity); dclj = (T)clampd((double)v, 1.0e-6, 1.0e32); }
      else A[i] = (double)v;
    }
    if (!on) {
#pragma unroll
      for (int i = 0; i < 9; ++i) A[i] = (i == j) ? 1.0 : 0.0; // idle lanes invert an identity
    }
    if (on) diag_clamped[9 * (size_t)c + j] = dclj;
    // in-place Gauss-Jordan without pivoting (spd_inverse), entry (r, c) in lane c: per step the pivot and the pivot
    // column come from lane k of the group
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      double f[9];
#pragma unroll
      for (int rr_ = 0; rr_ < 9; ++rr_) f[rr_] = __shfl(A[rr_], base + k, 64);
      const double piv = 1.0 / f[k];
      A[k] = (j == k) ? piv : A[k] * piv;
#pragma unroll
      for (int rr_ = 0; rr_ < 9; ++rr_) {
        if (rr_ == k) continue;
        A[rr_] = (j == k) ? -f[rr_] * piv : A[rr_] - f[rr_] * A[k];
      }
    }
    if (on) {
#pragma unroll
      for (int i = 0; i < 9; ++i) MinvC[81 * (size_t)c + i + 9 * j] = (T)A[i];
    }
    // z' = Minv r needs ROW j of the inverse in lane j: through LDS
    if (g < 7) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sA[wave][81 * g + i + 9 * j] = A[i];
    }
    __syncthreads();
    const T rvj = sj * bj;
    T zj = T(0);
    if (identity_precond) zj = rvj;
    else {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const T rq = __shfl(rvj, base + q, 64);
        if (g < 7) zj += (T)sA[wave][81 * g + j + 9 * q] * rq;
      }
    }
    if (on) {
      const size_t t = 9 * (size_t)c + j;
      if (zs) zs[t] = sj * zj; else x[t] = T(0);
      r[t] = rvj; zt[t] = zj;
      const T d = use_identity ? T(1) : dclj;
      if (cam_weight) { prr = (double)(rvj * rvj); prz = (double)(rvj * zj); pzz = (double)(d * zj * zj); }
    }
  } else {
    // point tiles of TPB / FIN_PL points, dealt round-robin to the point workgroups
    using V2 = typename Vec2T<T>::type;
    constexpr int PPB = TPB / FIN_PL;
    const int ntile = (Np + PPB - 1) / PPB;
    const unsigned jl = threadIdx.x % FIN_PL;
    // (round 6) the run bounds of the NEXT tile's points are requested while this tile is worked on: a tile's dependent chain is
    // bounds -> records -> sums -> inverse -> stores, and the first link is now off it
    int pa_n = 0, pe_n = 0;
    { const int l0 = (b - nbc) * PPB + (int)(threadIdx.x / FIN_PL); if (b - nbc < ntile && l0 < Np) { pa_n = pt_ptr[l0]; pe_n = pt_ptr[l0 + 1]; } }
    for (int tile = b - nbc; tile < ntile; tile += gridDim.x - nbc) {
      const int l = tile * PPB + (int)(threadIdx.x / FIN_PL);
      const bool on = l < Np;
      const int pa = pa_n, pe = pe_n;
      { const int tn = tile + (int)gridDim.x - nbc, ln = tn * PPB + (int)(threadIdx.x / FIN_PL); pa_n = pe_n = 0; if (tn < ntile && ln < Np) { pa_n = pt_ptr[ln]; pe_n = pt_ptr[ln + 1]; } }
      // The sums of a point's records, its scales, the scaled damped block, the inverse and z' = Minv r are taken in DOUBLE whatever
      // T is (for T = double nothing changes): the 3 x 3 block of a weakly observed point has a condition number of 1e3-1e4, and in
      // fp32 every rounding of the block's entries came back that much larger in the inverse (fp32 step 2.8 x further from the fp64
      // step than a plain fp32 restatement's, all of it in the point part; tools/fp32_first_iteration_probe.py).  Only outputs are rounded.
      double v[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] = 0.0;
      if (on && (VAR & 4)) { v[0] = v[3] = v[5] = (double)(2 + l % 3); v[6] = 1.0; }
      if (on && !(VAR & 4)) {
        for (int a = pa + (int)jl; a < pe; a += FIN_PL) {
          const V2 *gq = reinterpret_cast<const V2 *>(g9 + 8 * (size_t)a);
          const V2 q0 = gq[0], q1 = gq[1], q2 = gq[2], qe = gq[3]; // sqrt(w) Jp columns, sqrt(w) e
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
      for (int i = 0; i < 9; ++i) { v[i] += lane_xor<1>(v[i]); v[i] += lane_xor<2>(v[i]); } // DPP quad_perm, not ds_bpermute
      if (!on) continue;
      const bool pfixed = pt_fixed && pt_fixed[l];
      if (pfixed) {
#pragma unroll
        for (int i = 0; i < 9; ++i) v[i] = 0.0;
      }
      const bool sc_on = scale_system && !pfixed && !(VAR & 2);
      const double sd[3] = {sc_on ? 1.0 / (DBL_EPSILON + sqrt(v[0])) : 1.0, sc_on ? 1.0 / (DBL_EPSILON + sqrt(v[3])) : 1.0, sc_on ? 1.0 / (DBL_EPSILON + sqrt(v[5])) : 1.0};
      const double Hd[9] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5]};
      const T H[9] = {(T)v[0], (T)v[1], (T)v[2], (T)v[1], (T)v[3], (T)v[4], (T)v[2], (T)v[4], (T)v[5]};
      const T sv[3] = {(T)sd[0], (T)sd[1], (T)sd[2]};
      double A[9];
      T dcl[3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rw = 0; rw < 3; ++rw) {
          const double q = sd[rw] * Hd[rw + 3 * c] * sd[c];
          if (rw == c) { A[rw + 3 * c] = use_identity ? q + mu : q + mu * clampd(q, 1.0e-6, 1.0e32); dcl[rw] = (T)clampd(q, 1.0e-6, 1.0e32); }
          else A[rw + 3 * c] = q;
        }
      if (!(VAR & 2)) spd_inverse<3>(A);
      const double rvd[3] = {sd[0] * v[6], sd[1] * v[7], sd[2] * v[8]};
      const T rv[3] = {(T)rvd[0], (T)rvd[1], (T)rvd[2]};
      T z[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) z[i] = identity_precond ? rv[i] : (T)(A[i] * rvd[0] + A[i + 3] * rvd[1] + A[i + 6] * rvd[2]);
      // 36 outputs per point, written 4 lanes wide: lane jl stores outputs jl, jl + 4, ... (9 store instructions with every lane
      // active; one predicated store per output was 39 instructions)
      static_assert(FIN_PL == 4, "k_finalize_bj writes its outputs four lanes wide");
      const size_t t0 = 9 * (size_t)Nc + 3 * (size_t)l;
      const T out[36] = {H[0], H[1], H[2], H[3], H[4], H[5], H[6], H[7], H[8],
                         (T)A[0], (T)A[1], (T)A[2], (T)A[3], (T)A[4], (T)A[5], (T)A[6], (T)A[7], (T)A[8],
                         (T)v[6], (T)v[7], (T)v[8], sv[0], sv[1], sv[2], dcl[0], dcl[1], dcl[2],
                         zs ? sv[0] * z[0] : T(0), zs ? sv[1] * z[1] : T(0), zs ? sv[2] * z[2] : T(0), rv[0], rv[1], rv[2], z[0], z[1], z[2]};
      T *const dst[8] = {Hll + 9 * (size_t)l, MinvP + 9 * (size_t)l, bu + t0, scales + t0, diag_clamped + t0, (zs ? zs : x) + t0, r + t0, zt + t0};
#pragma unroll
      for (int m = 0; m < ((VAR & 1) ? 3 : 9); ++m) {
        // output e = 4 m + jl lives in array (e < 18 ? e / 9 : 2 + (e - 18) / 3) at offset (e < 18 ? e % 9 : (e - 18) % 3)
#define GR_FBJ_PTR(e) (dst[(e) < 18 ? (e) / 9 : 2 + ((e) - 18) / 3] + ((e) < 18 ? (e) % 9 : ((e) - 18) % 3))
        T *q = jl == 0 ? GR_FBJ_PTR(4 * m) : jl == 1 ? GR_FBJ_PTR(4 * m + 1) : jl == 2 ? GR_FBJ_PTR(4 * m + 2) : GR_FBJ_PTR(4 * m + 3);
#undef GR_FBJ_PTR
        const T val = jl == 0 ? out[4 * m] : jl == 1 ? out[4 * m + 1] : jl == 2 ? out[4 * m + 2] : out[4 * m + 3];
        *q = val;
      }
      if (jl == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const T d = use_identity ? T(1) : dcl[i];
          prr += (double)(rv[i] * rv[i]); prz += (double)(rv[i] * z[i]); pzz += (double)(d * z[i] * z[i]);
        }
      }
    }
  }
  // the three dots: one atomic each per WORKGROUP (one set per wave was 7 of the 30 us of the point part)
  prr = wave_sum(prr); prz = wave_sum(prz); pzz = wave_sum(pzz);
  if (lane == 0) { red3[wave][0] = prr; red3[wave][1] = prz; red3[wave][2] = pzz; }
  __syncthreads();
  if (threadIdx.x < 3 && !(VAR & 8)) {
    const double v = red3[0][threadIdx.x] + red3[1][threadIdx.x] + red3[2][threadIdx.x] + red3[3][threadIdx.x];
    if (zs && st.part0) st.part0[(size_t)threadIdx.x * st.n_part0 + blockIdx.x] = v; // threadIdx 0: r.r, 1: r.z', 2: z'.D.z'
    else slot_add(st.slots(0, threadIdx.x == 0 ? RR : threadIdx.x == 1 ? RZP : ZDZ), 0, v);
  }
}

// Graph::backup_parameters + Graph::apply_update (graph.hpp:292-309, ops/update.hpp:11-31) over cameras and
// points in one pass, plus this block's share of the compute_rho denominator sum dx (mu dx + b)
// (levenberg_marquardt.hpp:34-41) while dx and the scales are in registers anyway, plus the camera packs of
// the moved cameras.  Blocks [0, nbc): 28 cameras each (252 scalars, then the 28 packs);
// blocks [nbc, ...): one thread per point scalar.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_apply_update_rho(unsigned n, unsigned pose_dim, int nbc, int cam_weight, T *__restrict__ cams, T *__restrict__ pts,
                   T *__restrict__ cams_bak, T *__restrict__ pts_bak, const T *__restrict__ dx,
                   const T *__restrict__ scales, const T *__restrict__ bu, double mu, double *__restrict__ rho_partial,
                   T *__restrict__ pack, T *__restrict__ xp = nullptr, const LmDev *__restrict__ lm = nullptr,
                   PcgState rst = PcgState{}, int rst_cap = 0, const int *__restrict__ gate = nullptr) {
  if (lm) { if (lm->stop) return; mu = lm->mu; }
  if (gate && !*gate) return; // enqueued ahead of the PCG exit flag and the loop has not left: nothing to do
  if (blockIdx.x == gridDim.x - 1 && rst_cap > 0) {
    // one extra block: the PCG loop state of the solve that produced dx is spent; clearing it here lets the next
    // k_block_jacobi start the next loop itself (no separate init / first-update launches)
    for (int i = threadIdx.x; i < rst_cap * NSLOT * NS; i += TPB) rst.acc[slot_word(i)] = 0.0;
    for (int i = threadIdx.x; i < rst_cap; i += TPB) { rst.done[i] = 0; rst.pdp[i] = 0.0; rst.rz0[i] = (i == 0) ? __builtin_inf() : 0.0; }
    if (threadIdx.x == 0) rst.iters[0] = 0;
    return;
  }
  __shared__ double red[4];
  double rho = 0;
  if ((int)blockIdx.x < nbc) {
    // 28 cameras = 252 scalars per block, one thread per scalar (coalesced), the new values go through LDS to
    // the 28 threads that rebuild the packs
    __shared__ T cs[252];
    const unsigned i = blockIdx.x * 252u + threadIdx.x;
    if (threadIdx.x < 252 && i < pose_dim) {
      const T d = dx[i], s = scales[i], xo = cams[i];
      cams_bak[i] = xo;
      const T xn = xo + d * s;
      cams[i] = xn;
      cs[threadIdx.x] = xn;
      if (cam_weight) rho = (double)(d * ((T)mu * d + s * bu[i]));
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
    const unsigned q = (blockIdx.x - nbc) * TPB + threadIdx.x, i = pose_dim + q;
    if (i < n) {
      const T d = dx[i], s = scales[i];
      const T xo = pts[q];
      pts_bak[q] = xo;
      const T xn = xo + d * s;
      pts[q] = xn;
      if (xp) xp[8 * (size_t)(q / 3u) + q % 3u] = xn; // operator's point records
      rho = (double)(d * ((T)mu * d + s * bu[i]));
    }
  }
  rho = block_sum_256(rho, red);
  if (threadIdx.x == 0) rho_partial[blockIdx.x] = rho;
}

// Graph::revert_parameters (graph.hpp:311-318) + the camera packs of the restored cameras, one launch: workgroups [0, nbc) restore
// 28 cameras each (252 scalars, then their packs), the others one point scalar per thread
template <typename T>
__global__ void __launch_bounds__(TPB)
k_revert_pack(unsigned pose_dim, unsigned npt, int nbc, T *__restrict__ cams, T *__restrict__ pts, const T *__restrict__ cams_bak,
              const T *__restrict__ pts_bak, T *__restrict__ pack) {
  if ((int)blockIdx.x < nbc) {
    __shared__ T cs[252];
    const unsigned i = blockIdx.x * 252u + threadIdx.x;
    if (threadIdx.x < 252 && i < pose_dim) { const T v = cams_bak[i]; cams[i] = v; cs[threadIdx.x] = v; }
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
    const unsigned q = (blockIdx.x - nbc) * TPB + threadIdx.x;
    if (q < npt) pts[q] = pts_bak[q];
  }
}

// Point records of the matrix-free operator: [X Y Z | ps_x ps_y ps_z | pad pad] per point, one aligned
// 64-byte (fp64) / 32-byte (fp32) sector.  The operator's two per-observation gathers (point, scaled
// direction) become ONE sector; on Venice/Final-shaped graphs, where every such gather is a cache-line
// miss, that halves the line traffic that bounds the kernel.
template <typename T> __global__ void k_points_to_records(int Np, const T *__restrict__ pts, T *__restrict__ xp) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 3u * (unsigned)Np) xp[8 * (size_t)(t / 3u) + t % 3u] = pts[t];
}

// XCD-aware tile ranges for the persistent per-observation kernels.  Workgroups are dealt
// round-robin over the 8 XCDs (blockIdx % 8 share an XCD, MI355X_MICROARCH.md), each with its own
// 4 MiB L2.  With gridDim.x a multiple of 8, XCD x walks the x-th eighth of the tiles, i.e. a
// contiguous range of cameras, whose point gathers / scatters (points are ordered by first
// camera) then stay inside that XCD's L2 instead of every L2 seeing every point.  Speed only.
// ntiles < 0 selects the SWEEP form used with point-tiled observation orders (Engine::build_tiled_order): XCD x still
// owns the x-th eighth of the tiles (= its point tiles, one after the other), but its workgroups take the tiles of that
// eighth round-robin, so that at any moment the whole XCD works inside one L2-sized point tile.
__device__ __forceinline__ void xcd_tile_range(int ntiles, int &t0, int &t1, int &tstep) {
  const int nb = gridDim.x >> 3, x = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int nt = ntiles < 0 ? -ntiles : ntiles;
  const int x0 = (int)((long long)x * nt / 8), x1 = (int)((long long)(x + 1) * nt / 8);
  if (ntiles < 0) { t0 = x0 + bi; t1 = x1; tstep = nb; return; }
  t0 = x0 + (int)((long long)bi * (x1 - x0) / nb);
  t1 = x0 + (int)((long long)(bi + 1) * (x1 - x0) / nb);
  tstep = 1;
}

// scalars of iteration k as every wave derives them (all lanes must call)
struct PcgIter { double rzp, rscale, rz; };
__device__ __forceinline__ PcgIter pcg_iter(const PcgState &st, int k) {
  PcgIter it;
  it.rzp = slot_sum(st.slots(k, RZP), 0);
  it.rscale = 1.0 / sqrt(slot_sum(st.slots(k, RR), 0));
  it.rz = it.rzp * it.rscale;
  return it;
}

// Every block calls this once (all threads).  v0/v1: thread 0's block sums.  Returns true
// in ALL threads of the last block, after which tot0/tot1 (thread 0 only) hold the grid totals
// summed in fixed block order.
// Visibility: write-through (sc1) stores, drained vmcnt, then a RELAXED ticket; the last block
// reads with sc1 loads (cdna_hip_programming.md G16, form R1) — an agent release fence per
// block would write back the whole L2 thousands of times.
// Contention: returning atomics on ONE address serialise at 15-30 ns each, i.e. 40-90 us for
// a 2,600-block grid (measured), so the ticket is two-level: 64 group counters (blockIdx % 64),
// and only the block that completes its group touches the top counter.
// ticket[0] = top counter, ticket[1 + g] = group counters; all return to 0.
__device__ __forceinline__ bool grid_sum2(double v0, double v1, double *partial, unsigned *ticket,
                                          double *red, double &tot0, double &tot1) {
  __shared__ bool s_last;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partial[2 * blockIdx.x], v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partial[2 * blockIdx.x + 1], v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned g = blockIdx.x % TICKET_GROUPS;
    const unsigned ngroups = gridDim.x < TICKET_GROUPS ? gridDim.x : TICKET_GROUPS;
    const unsigned in_group = (gridDim.x - g + TICKET_GROUPS - 1) / TICKET_GROUPS;
    bool last = false;
    const unsigned tk = __hip_atomic_fetch_add(&ticket[1 + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk == in_group - 1) {
      __hip_atomic_store(&ticket[1 + g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned top = __hip_atomic_fetch_add(&ticket[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (top == ngroups - 1);
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return false;
  double s0 = 0, s1 = 0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += TPB) {
    s0 += __hip_atomic_load(&partial[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s1 += __hip_atomic_load(&partial[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  s0 = block_sum_256(s0, red);
  s1 = block_sum_256(s1, red);
  if (threadIdx.x == 0) { tot0 = s0; tot1 = s1; __hip_atomic_store(&ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  return true;
}

// ---------------------------------------------------------------------------
// Graph::linearize + Hessian::update_values (A4-A10 of SURVEY §8a), one launch.
//   camera side : 45 + 9 sums per (wave, camera) segment -> cam_partial[seg][54]
//   point side  : per-observation sqrt(w) [Jp (6), e (2)]        -> g9[pm position][8]
//   Hcp^u       : per-observation 9x3 block -> Hcp[pm position][27]  (Schur solvers only)
//   chi2        : block partial (summed in fixed order by k_linearize_finalize)
// Persistent form: gridDim.x = min(ntiles, CUs x 4) blocks, each walks a CONTIGUOUS range of
// 256-observation tiles and prefetches the next tile's index streams before it computes the
// current one, so the per-block fixed costs (launch, tail, chi2 partial) are amortised and the
// index -> gather dependency is off the critical path.
// LV (diagnostic builds only, GR_DIAG): 1 no point-record write, 2 no camera reduction, 4 no point gather, 8 no Jacobian math
// (Round 4, measured and removed: the camera pack through a wave-uniform index (scalar loads, as in k_pcg_operator) with the Jacobian
// evaluated inside the per-camera loop: Ladybug-1723 27.8 -> 27.5 us, Final-13682 1 608 -> 1 565 us, Venice-1778 fp32 154 -> 163 us —
// a vector load whose lanes share one address is cheap, and the fp32 kernel took 13 more VGPRs.)
template <typename T, bool WRITE_HCP, typename JT = T, int LV = 0>
__global__ void __launch_bounds__(TPB, LIN_WAVES)
k_linearize(int No, int ntiles, const int *__restrict__ cam_cm, const int *__restrict__ pt_cm,
            const int *__restrict__ pos_cm, const T *__restrict__ obs_cm, const int *__restrict__ blk_seg,
            const int *__restrict__ seg_slot, const T *__restrict__ pts, const T *__restrict__ pack,
            int loss_kind, T loss_delta, T *__restrict__ g9, T *__restrict__ Hcp, T *__restrict__ cam_partial,
            double *__restrict__ chi2_partial, const LmDev *__restrict__ lm = nullptr,
            const int *__restrict__ gate = nullptr,
            // fixed vertices (WRITE_HCP only): the camera-point block of an observation whose camera or point is fixed is zero
            // (the reference computes no Jacobian block for a fixed vertex, ops/linearize.hpp:24)
            const unsigned char *__restrict__ cam_fixed = nullptr, const unsigned char *__restrict__ pt_fixed = nullptr,
            // LM loop: the PCG loop state of the solve whose step this linearisation evaluates is spent (every kernel that
            // reads it precedes this launch); the last workgroup clears it, so the next k_finalize_bj / k_block_jacobi
            // starts the next loop itself
            PcgState rst = PcgState{}, int rst_cap = 0,
            // Schur solvers whose rejected steps do not stop the head (kernels_sf.hpp): the blocks go to the buffer that does NOT hold the
            // current point's (hs->hsel: 0 = Hcp, 1 = Hcp_alt), which a rejected step still needs
            T *__restrict__ Hcp_alt = nullptr, const LmDev *__restrict__ hs = nullptr) {
  if (lm && lm->stop) return;
  if (gate && !*gate) return;
  if (WRITE_HCP && hs && (hs->hsel & 1) == 0) Hcp = Hcp_alt;
  if (rst_cap > 0 && blockIdx.x == gridDim.x - 1) {
    for (int i = threadIdx.x; i < rst_cap * NSLOT * NS; i += TPB) rst.acc[slot_word(i)] = 0.0;
    for (int i = threadIdx.x; i < rst_cap; i += TPB) { rst.done[i] = 0; rst.pdp[i] = 0.0; rst.rz0[i] = (i == 0) ? __builtin_inf() : 0.0; }
    if (threadIdx.x == 0) rst.iters[0] = 0;
  }
  __shared__ double red[4];
  using V2 = typename Vec2T<T>::type;
  const int lane = threadIdx.x & 63;
  int j0, jstride, niter, jlim;
  xcd_obs_range(ntiles, No, j0, jstride, niter, jlim);
  double chi2 = 0.0;
  int j = j0 + threadIdx.x;
  bool valid = niter > 0 && j < jlim;
  int c_n = -1, l_n = 0, a_n = 0;
  V2 o_n{};
  if (valid) { c_n = cam_cm[j]; l_n = pt_cm[j]; a_n = pos_cm[j]; o_n = reinterpret_cast<const V2 *>(obs_cm)[j]; }
  for (int it = 0; it < niter; ++it) {
    const int c = c_n, l = l_n;
    const size_t a = (size_t)a_n;
    const V2 o = o_n;
    const int jn = j + jstride;
    const bool validn = (it + 1 < niter) && jn < jlim;
    if (validn) { c_n = cam_cm[jn]; l_n = pt_cm[jn]; a_n = pos_cm[jn]; o_n = reinterpret_cast<const V2 *>(obs_cm)[jn]; }
    T Jc[18], e0 = 0, e1 = 0, w = 0;
#pragma unroll
    for (int i = 0; i < 18; ++i) Jc[i] = T(0);
    if (valid) {
      T pk[PACK], Jp[6];
      load_pack(pack, c, pk);
      const size_t lp = (LV & 4) ? (size_t)(j & 1023) : (size_t)l;
      if (LV & 8) {
        e0 = o.x; e1 = o.y;
#pragma unroll
        for (int i = 0; i < 18; ++i) Jc[i] = pts[3 * lp + (i % 3)] + pk[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) Jp[i] = pts[3 * lp + (i % 3)] - pk[i];
      } else
        bal_linearize_j<T, JT>(pk, pts[3 * lp], pts[3 * lp + 1], pts[3 * lp + 2], o.x, o.y, e0, e1, Jc, Jp);
      const T raw = e0 * e0 + e1 * e1;
      w = loss_drho(loss_kind, loss_delta, raw);
      chi2 += (double)loss_rho(loss_kind, loss_delta, raw);
      const T wp0x = w * Jp[0], wp0y = w * Jp[1], wp1x = w * Jp[2], wp1y = w * Jp[3], wp2x = w * Jp[4], wp2y = w * Jp[5];
      // point-side record: sqrt(w) [Jp (6), e (2)] = 8 scalars = one aligned 64-byte (fp64) / 32-byte (fp32)
      // sector per observation, written with vector stores; the per-point kernel forms Jp^T Jp and Jp^T e
      // from it.  (The 9-scalar [w Jp^T Jp, -w Jp^T e] record it replaces straddled sectors: 72 MB of HBM
      // writes for 49 MB of payload on Ladybug-1723.)
      if (!(LV & 1)) {
        const T sw = t_sqrt(w);
        V2 *g = reinterpret_cast<V2 *>(g9 + 8 * a);
        V2 q0, q1, q2, q3;
        q0.x = sw * Jp[0]; q0.y = sw * Jp[1]; q1.x = sw * Jp[2]; q1.y = sw * Jp[3];
        q2.x = sw * Jp[4]; q2.y = sw * Jp[5]; q3.x = sw * e0; q3.y = sw * e1;
        g[0] = q0; g[1] = q1; g[2] = q2; g[3] = q3;
      }
      if (WRITE_HCP) {
        T *h = Hcp + 27 * a;
        const T keep = ((cam_fixed && cam_fixed[c]) || (pt_fixed && pt_fixed[l])) ? T(0) : T(1);
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          h[r] = keep * (Jc[2 * r] * wp0x + Jc[2 * r + 1] * wp0y);
          h[r + 9] = keep * (Jc[2 * r] * wp1x + Jc[2 * r + 1] * wp1y);
          h[r + 18] = keep * (Jc[2 * r] * wp2x + Jc[2 * r + 1] * wp2y);
        }
      }
    }
    // camera-side reduction, once per distinct camera in the wave (usually one)
    unsigned long long remaining = (LV & 2) ? 0ull : __ballot(valid);
    if (LV & 2) chi2 += (double)(Jc[0] + Jc[17] + w);
    int segf = blk_seg[__builtin_amdgcn_readfirstlane(j >> 6)]; // flat id of the block's first (run, block) segment; the next runs follow
    while (remaining) {
      const int leader = __builtin_ctzll(remaining);
      const int cl = __builtin_amdgcn_readlane(c, leader); // leader is wave-uniform (from a ballot): v_readlane, not ds_bpermute
      const int segl = seg_slot[segf++];
      const bool mine = valid && c == cl;
      const T wm = mine ? w : T(0);
      T acc[64];
      int kk = 0;
#pragma unroll
      for (int col = 0; col < 9; ++col) {
        const T wx = wm * Jc[2 * col], wy = wm * Jc[2 * col + 1];
#pragma unroll
        for (int row = 0; row <= col; ++row) acc[kk++] = Jc[2 * row] * wx + Jc[2 * row + 1] * wy;
        acc[45 + col] = -(wx * e0 + wy * e1);
      }
#pragma unroll
      for (int i = 54; i < 64; ++i) acc[i] = T(0);
      const T tot = wave_transpose_sum<T, 64>(acc, lane);
      if (lane < 54) cam_partial[54 * (size_t)segl + lane] = tot;
      remaining &= ~__ballot(mine);
    }
    valid = validn;
    j = jn;
  }
  chi2 = block_sum_256(chi2, red);
  if (threadIdx.x == 0) chi2_partial[blockIdx.x] = chi2;
}

// Finalise a linearisation:
//   threads [0, 90 Nc)        fixed-order sum of the segment partials -> Hcc^u, bc^u, camera scales
//   threads [90 Nc, +Np)      one per point: sum over its observations' records of Jp^T Jp, -Jp^T e -> Hll^u, bl^u, point scales
//   block 0                   chi2 total
// (column scales: graph.hpp:254-270)
template <typename T>
__global__ void __launch_bounds__(TPB)
k_linearize_finalize(int Nc, int Np, int scale_system, int cam_scales, const int *__restrict__ cam_seg_ptr,
                     const T *__restrict__ cam_partial, const int *__restrict__ pt_ptr,
                     const T *__restrict__ g9, T *__restrict__ Hcc, T *__restrict__ bc, T *__restrict__ Hll,
                     T *__restrict__ bl, T *__restrict__ scales, int n_partials,
                     const double *__restrict__ chi2_partial, double *__restrict__ chi2_out,
                     const double *__restrict__ rho_partial = nullptr, int n_rho = 0,
                     volatile double *hres = nullptr, volatile int *hres_seq = nullptr, int seq = 0,
                     const int *__restrict__ gate = nullptr,
                     // fixed vertices: no Jacobian block (ops/linearize.hpp:24) -> zero Hessian block, zero gradient, scale 1
                     const unsigned char *__restrict__ cam_fixed = nullptr, const unsigned char *__restrict__ pt_fixed = nullptr,
                     // landmark shards, fused message (comm.hpp IpcFused; fz.boxes == nullptr: off): every camera thread also stores its
                     // LOCAL sum (entry 90 c + e of the message) into every peer's mailbox, the launch's last workgroup adds the
                     // scalars (chi2, rho denominator) at scal_off, raises the flags, waits for the peers and leaves the scalars
                     // summed over the ranks in chi2_out / hres; k_shard_cam_sums then sums the camera entries over the ranks
                     IpcFused fz = IpcFused{}, unsigned long long scal_off = 0) {
  if (gate && !*gate) return;
  const unsigned t = blockIdx.x * TPB + threadIdx.x;
  const unsigned ncam = 90u * (unsigned)Nc, ncam_pad = (ncam + TPB - 1) / TPB * TPB;
  __shared__ unsigned long long s_seq;
  int fz_set = 0;
  if (fz.boxes) {
    if (threadIdx.x == 0) s_seq = __hip_atomic_load(fz.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
    __syncthreads();
    fz_set = (int)(s_seq & 1ull);
  }
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
    { // four loads in flight (the last batch clamped), added in segment order
      int sg = cam_seg_ptr[c];
      const int sg1 = cam_seg_ptr[c + 1], last = sg1 - 1;
      for (; sg < sg1; sg += 4) {
        const int n = sg1 - sg;
        const T q0 = cam_partial[54 * (size_t)sg + idx], q1 = cam_partial[54 * (size_t)(sg + 1 < last ? sg + 1 : last) + idx],
                q2 = cam_partial[54 * (size_t)(sg + 2 < last ? sg + 2 : last) + idx], q3 = cam_partial[54 * (size_t)(sg + 3 < last ? sg + 3 : last) + idx];
        s += q0;
        if (n > 1) s += q1;
        if (n > 2) s += q2;
        if (n > 3) s += q3;
      }
    }
    const bool fixed = cam_fixed && cam_fixed[c];
    if (fixed) s = T(0);
    if (e < 81u) {
      Hcc[81 * (size_t)c + e] = s;
      if (row == col && cam_scales) scales[9 * c + row] = (scale_system && !fixed) ? (T)(1.0 / (DBL_EPSILON + sqrt((double)s))) : T(1);
    } else bc[9 * c + (e - 81u)] = s;
    if (fz.boxes && (!fz.contrib || cam_seg_ptr[c + 1] > cam_seg_ptr[c])) // with contributor masks: only the cameras this rank holds
      for (int r = 0; r < fz.size; ++r) ipc_store(reinterpret_cast<T *>(fz.slot(fz.push_box(r), fz_set, fz.push_slot(r))) + t, fz.push_value(r, s));
  } else if (t >= ncam_pad && t < ncam_pad + FIN_PL * (unsigned)Np) { // point part starts on a block boundary
    // FIN_PL lanes share a point (records j, j + FIN_PL, ...): the serial chain of dependent record loads is
    // ~deg / FIN_PL long; then a butterfly over the FIN_PL lanes (fixed order)
    const unsigned l = (t - ncam_pad) / FIN_PL, jl = (t - ncam_pad) % FIN_PL;
    T v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = T(0);
    using V2 = typename Vec2T<T>::type;
    for (int a = pt_ptr[l] + (int)jl; a < pt_ptr[l + 1]; a += FIN_PL) {
      const V2 *g = reinterpret_cast<const V2 *>(g9 + 8 * (size_t)a);
      const V2 c0 = g[0], c1 = g[1], c2 = g[2], e = g[3]; // sqrt(w) Jp columns, sqrt(w) e
      v[0] += c0.x * c0.x + c0.y * c0.y;
      v[1] += c0.x * c1.x + c0.y * c1.y;
      v[2] += c0.x * c2.x + c0.y * c2.y;
      v[3] += c1.x * c1.x + c1.y * c1.y;
      v[4] += c1.x * c2.x + c1.y * c2.y;
      v[5] += c2.x * c2.x + c2.y * c2.y;
      v[6] -= c0.x * e.x + c0.y * e.y;
      v[7] -= c1.x * e.x + c1.y * e.y;
      v[8] -= c2.x * e.x + c2.y * e.y;
    }
    static_assert(FIN_PL == 4, "quad butterfly");
#pragma unroll
    for (int i = 0; i < 9; ++i) { v[i] += lane_xor<1>(v[i]); v[i] += lane_xor<2>(v[i]); } // DPP quad_perm, not ds_bpermute
    const bool pfixed = pt_fixed && pt_fixed[l];
    if (pfixed) {
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] = T(0);
    }
    if (pfixed) scale_system = 0; // this thread's point only: scale 1
#if FIN_PL == 4
    { // every lane of the group holds the sums: the 15 outputs are written 4 lanes wide (lane jl: outputs jl, jl+4, ...)
      const T out[16] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5], v[6], v[7], v[8],
                         scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)v[0]))) : T(1),
                         scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)v[3]))) : T(1),
                         scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)v[5]))) : T(1), T(0)};
      T *H = Hll + 9 * (size_t)l, *b3 = bl + 3 * (size_t)l, *s3 = scales + 9 * (size_t)Nc + 3 * (size_t)l;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const T val = jl == 0 ? out[4 * m] : jl == 1 ? out[4 * m + 1] : jl == 2 ? out[4 * m + 2] : out[4 * m + 3];
        const unsigned idx = 4u * m + jl;
        if (idx < 9u) H[idx] = val;
        else if (idx < 12u) b3[idx - 9u] = val;
        else if (idx < 15u) s3[idx - 12u] = val;
      }
    }
#else
    if (jl == 0) {
      T *H = Hll + 9 * (size_t)l;
      H[0] = v[0]; H[1] = v[1]; H[2] = v[2]; H[3] = v[1]; H[4] = v[3]; H[5] = v[4]; H[6] = v[2]; H[7] = v[4]; H[8] = v[5];
      bl[3 * (size_t)l] = v[6]; bl[3 * (size_t)l + 1] = v[7]; bl[3 * (size_t)l + 2] = v[8];
      T *s = scales + 9 * (size_t)Nc + 3 * (size_t)l;
      s[0] = scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)v[0]))) : T(1);
      s[1] = scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)v[3]))) : T(1);
      s[2] = scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)v[5]))) : T(1);
    }
#endif
  }
  if (blockIdx.x == 0 && chi2_out) {
    __shared__ double red[4];
    double s = 0;
    for (int k = threadIdx.x; k < n_partials; k += TPB) s += chi2_partial[k];
    s = block_sum_256(s, red);
    // speculative LM step: the trial chi2 is this linearisation's chi2; the rho denominator was left
    // as block partials by k_apply_update_rho; both are mirrored to pinned host memory, then `seq`
    double r = 0;
    for (int k = threadIdx.x; k < n_rho; k += TPB) r += rho_partial[k];
    r = block_sum_256(r, red);
    if (threadIdx.x == 0) {
      if (fz.boxes) { // read by the launch's last workgroup, possibly on another XCD: written through
        __hip_atomic_store(&chi2_out[0], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&chi2_out[1], rho_partial ? r : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
      chi2_out[0] = s;
      if (rho_partial) chi2_out[1] = r;
      if (hres) {
        hres[0] = s; hres[1] = r;
        __threadfence_system();
        *hres_seq = seq;
      }
      }
    }
  }
  if (!fz.boxes) return;
  // ---- fused message: completion count over the workgroups, then ONE workgroup finishes the message (as shard_push_tail does)
  __shared__ unsigned s_last;
  __shared__ int s_bad;
  __shared__ double s_g[2 * 64];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(fz.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old + 1u == gridDim.x ? 1u : 0u;
    if (s_last) __hip_atomic_store(fz.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_bad = 0;
  }
  __syncthreads();
  if (!s_last) return;
  const unsigned long long mseq = s_seq;
  if ((int)threadIdx.x < 2 * fz.size) {
    const int q = threadIdx.x / fz.size, r = threadIdx.x % fz.size;
    const double v = __hip_atomic_load(&chi2_out[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ipc_store(reinterpret_cast<double *>(fz.slot(fz.push_box(r), fz_set, fz.push_slot(r)) + scal_off) + q, fz.push_value(r, v));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(fz.seq, mseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if ((int)threadIdx.x < fz.size) ipc_store(fz.flag(fz.push_box(threadIdx.x), fz_set, fz.push_slot(threadIdx.x)), mseq);
  if (threadIdx.x == 0 && ipc_load(reinterpret_cast<const unsigned long long *>(fz.boxes[fz.rank]) + 500) != 0ull) s_bad = 1; // an earlier message timed out: no second wait
  __syncthreads();
  if ((int)threadIdx.x < fz.size && !s_bad) {
    const unsigned long long *flag = fz.flag(fz.rank, fz_set, threadIdx.x);
    const long long t0 = wall_clock64();
    while (ipc_load(flag) < mseq) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > fz.timeout_ticks) {
        ipc_store(reinterpret_cast<unsigned long long *>(fz.boxes[fz.rank]) + 500, 1ull);
        if (fz.h_err) { *fz.h_err = 1; __threadfence_system(); }
        s_bad = 1;
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if ((int)threadIdx.x < 2 * fz.size)
    s_g[threadIdx.x] = ipc_load(reinterpret_cast<const double *>(fz.slot(fz.rank, fz_set, threadIdx.x % fz.size) + scal_off) + threadIdx.x / fz.size);
  __syncthreads();
  if (threadIdx.x == 0) {
    double c2 = 0, rh = 0;
    for (int r = 0; r < fz.size; ++r) { c2 += s_g[r]; rh += s_g[fz.size + r]; }
    if (s_bad) c2 = rh = __builtin_nan(""); // a peer never arrived: the host finds the error word before it uses these
    chi2_out[0] = c2; chi2_out[1] = rh;
    if (hres) {
      hres[0] = c2; hres[1] = rh;
      __threadfence_system();
      *hres_seq = seq;
    }
  }
}

// Landmark shards, fused linearisation message: Hcc, bc and the camera column scales from the ranks' entries in the mailbox,
// summed in rank order (the same bits on every rank); entry 90 c + e as k_linearize_finalize's camera threads index them
template <typename T>
__global__ void __launch_bounds__(TPB) k_shard_cam_sums(int Nc, int scale_system, IpcFused fz, T *__restrict__ Hcc, T *__restrict__ bc, T *__restrict__ scales,
                                                        const unsigned char *__restrict__ cam_fixed) {
  __shared__ int s_set;
  if (threadIdx.x == 0) s_set = (int)(__hip_atomic_load(fz.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1ull);
  __syncthreads();
  const unsigned t = blockIdx.x * TPB + threadIdx.x;
  if (t >= 90u * (unsigned)Nc) return;
  T s = T(0);
  const unsigned who = fz.contributors(t / 90u);
  for (int r0 = 0; r0 < fz.size; r0 += 8) { // eight uncached loads in flight, added in rank order
    T q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = (r0 + u < fz.size && IpcFused::rank_in(who, r0 + u)) ? ipc_load(reinterpret_cast<const T *>(fz.slot(fz.rank, s_set, r0 + u)) + t) : T(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) if (r0 + u < fz.size && IpcFused::rank_in(who, r0 + u)) s += q[u];
  }
  const unsigned c = t / 90u, e = t % 90u;
  if (e < 81u) {
    Hcc[81 * (size_t)c + e] = s;
    if (e % 10u == 0u) scales[9 * c + e / 10u] = (scale_system && !(cam_fixed && cam_fixed[c])) ? (T)(1.0 / (DBL_EPSILON + sqrt((double)s))) : T(1);
  } else bc[9 * c + (e - 81u)] = s;
}


// Multi-GPU: camera column scales from the all-reduced Hcc diagonal (graph.hpp:262-270)
template <typename T>
__global__ void k_camera_scales(int Nc, int scale_system, const T *__restrict__ Hcc, T *__restrict__ scales) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 9u * (unsigned)Nc) return;
  const T d = Hcc[81 * (size_t)(t / 9u) + 10 * (t % 9u)];
  scales[t] = scale_system ? (T)(1.0 / (DBL_EPSILON + sqrt((double)d))) : T(1);
}

// chi2 of a trial step (Graph::compute_error + Graph::chi2) fused with compute_rho's
// denominator sum dx (mu dx + b) (levenberg_marquardt.hpp:34-41).  One observation per
// thread (cm order); the first ceil(n/256) blocks also take one vector element each.
// The last block writes dscal[0] = chi2, dscal[1] = rho denominator (fixed-order sums)
// and mirrors them to pinned host memory (hres[0..1], then hres_seq = seq).
template <typename T>
__global__ void __launch_bounds__(TPB)
k_chi2(int No, unsigned n, unsigned pose_dim, int cam_weight, const int *__restrict__ cam_cm, const int *__restrict__ pt_cm,
       const int *__restrict__ pos_cm, const T *__restrict__ obs_cm, const T *__restrict__ pts,
       const T *__restrict__ pack, int loss_kind, T loss_delta, const T *__restrict__ dx,
       const T *__restrict__ bu, const T *__restrict__ scales, double mu, double *__restrict__ partial,
       unsigned *__restrict__ ticket, double *__restrict__ dscal, volatile double *hres, volatile int *hres_seq,
       int seq, T *__restrict__ res_out) {
  __shared__ double red[4];
  using V2 = typename Vec2T<T>::type;
  double chi2 = 0, rho = 0;
  const int stride = gridDim.x * TPB;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < No; j += stride) {
    const int c = cam_cm[j], l = pt_cm[j];
    const T *pk = pack + PACK * (size_t)c;
    const V2 o = reinterpret_cast<const V2 *>(obs_cm)[j];
    T e0, e1;
    bal_residual(pk, pts[3 * (size_t)l], pts[3 * (size_t)l + 1], pts[3 * (size_t)l + 2], o.x, o.y, e0, e1);
    chi2 += (double)loss_rho(loss_kind, loss_delta, e0 * e0 + e1 * e1);
    if (res_out) { const size_t a = (size_t)pos_cm[j]; res_out[2 * a] = e0; res_out[2 * a + 1] = e1; }
  }
  if (dx) {
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < n; i += (unsigned)stride) {
      const T x = dx[i];
      if (i >= pose_dim || cam_weight) rho += (double)(x * ((T)mu * x + scales[i] * bu[i]));
    }
  }
  chi2 = block_sum_256(chi2, red);
  rho = block_sum_256(rho, red);
  double t0, t1;
  if (grid_sum2(chi2, rho, partial, ticket, red, t0, t1) && threadIdx.x == 0) {
    dscal[0] = t0; dscal[1] = t1;
    if (hres) {
      hres[0] = t0; hres[1] = t1;
      __threadfence_system();
      *hres_seq = seq;
    }
  }
}

// Loop decision before iteration k of the LAZY form = the scalar half of k_pcg_direction(k - 1) (pcg.hpp:108-127 for
// k = 0, :184-217 otherwise), evaluated by EVERY wave of the operator launch of iteration k from the dot-product slots
// (same inputs, same arithmetic: all waves agree); workgroup 0 publishes it (done[k], rz0[k], pdp[k], beta[k], scale[k],
// iteration count, the host flag of iteration k - 1 — which the host therefore sees while this launch is still running).
// Returns false when the launch has nothing to do (loop left; a rejected step has then been reverted by all workgroups).
template <typename T> struct PcgStep { T beta, scale; };
template <typename T>
__device__ __forceinline__ bool pcg_decide(const PcgState &st, int k, PcgStep<T> &stp) {
  const bool first = (blockIdx.x == 0 && threadIdx.x == 0);
  stp.beta = T(0); stp.scale = T(0);
  if (k == 0) {
    stp.scale = (T)(1.0 / (double)(T)sqrt((double)(T)slot_sum(st.slots(0, RR), 0)));
    const double zdz = slot_sum(st.slots(0, ZDZ), 0);
    if (first) { st.pdp[0] = (double)stp.scale * (double)stp.scale * zdz; st.beta[0] = 0.0; st.scale[0] = (double)stp.scale; }
    return true;
  }
  const int j = k - 1;
  const double rz0 = st.rz0[j];
  bool leave = st.done[j] != 0;
  PcgIter it{};
  if (!leave) { it = pcg_iter(st, j); leave = (it.rzp == 0.0); }
  if (leave) {
    if (first) { st.done[k] = 1; st.rz0[k] = rz0; if (st.left) *st.left = 1; st.hflag[j] = 2; __threadfence_system(); }
    return false;
  }
  const PcgIter nx = pcg_iter(st, k);
  const double pdz = slot_sum(st.slots(k, PDZ), 0), zdz = slot_sum(st.slots(k, ZDZ), 0);
  const T rz = (T)it.rzp * (T)(double)(T)it.rscale, rz_new = (T)nx.rzp * (T)(double)(T)nx.rscale;
  const bool reject = (fabs((double)rz_new) > st.rej * rz0) || (rz_new != rz_new);
  const bool done_next = reject || fabs((double)rz_new) < st.tol;
  stp.beta = rz_new / rz;
  stp.scale = (T)(double)(T)nx.rscale;
  if (first) {
    st.rz0[k] = reject ? rz0 : fmin(rz0, fabs((double)rz_new));
    st.done[k] = done_next ? 1 : 0;
    st.pdp[k] = (double)stp.beta * (double)stp.beta * st.pdp[j] + 2.0 * (double)stp.beta * (double)stp.scale * pdz + (double)stp.scale * (double)stp.scale * zdz;
    st.beta[k] = (double)stp.beta; st.scale[k] = (double)stp.scale;
    st.iters[0] = k;
    *st.hiters = k;
    if (done_next && st.left) *st.left = 1;
    st.hflag[j] = done_next ? 2 : 1;
    __threadfence_system();
  }
  if (reject) {
    T *x = static_cast<T *>(st.x);
    const T *xb = static_cast<const T *>(st.xb);
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < st.n; t += gridDim.x * blockDim.x) x[t] = xb[t];
  }
  return !done_next;
}
// Single-reduction form: loop decision and step scalars at the start of update k, from record k of the dots (complete,
// i.e. all-reduced, only now).  Every wave evaluates it; workgroup 0 publishes.  The decision concerns iteration k - 1
// (solver/pcg.hpp:166-229: rejection ratio, tolerance, rz == 0) and is what the host flag of iteration k - 1 reports.
template <typename T> struct PcgCgStep { T alpha, beta, sigma; };
// dots[RZP], dots[RR], dots[ZDZ], dots[DEN]: record k of the dot products, complete (summed over the ranks)
template <typename T>
__device__ __forceinline__ bool pcg_cg_decide_v(const PcgState &st, int k, double mu, PcgCgStep<T> &stp, const double (&dots)[NSLOT]) {
  const bool first = (blockIdx.x == 0 && threadIdx.x == 0);
  const double rzp = dots[RZP], rr = dots[RR];
  const T sigma = (T)(1.0 / (double)(T)sqrt((double)(T)rr));
  const T gamma = (T)rzp * sigma;
  bool stop = false, reject = false;
  double rz0 = __builtin_inf();
  if (k > 0) {
    rz0 = st.rz0[k - 1];
    reject = (fabs((double)gamma) > st.rej * rz0) || (gamma != gamma);
    stop = reject || fabs((double)gamma) < st.tol;
  }
  if (rzp == 0.0) stop = true; // pcg.hpp:133: rz == 0 ends the loop before the iteration starts
  if (first) {
    st.rz0[k] = (k == 0 || reject) ? rz0 : fmin(rz0, fabs((double)gamma));
    st.done[k] = stop ? 1 : 0;
    if (k > 0) { st.iters[0] = k; *st.hiters = k; }
    if (stop && st.left) *st.left = 1;
    if (k > 0) st.hflag[k - 1] = stop ? 2 : 1;
    else if (stop) st.hflag[0] = 2;
    __threadfence_system();
  }
  if (reject) {
    T *x = static_cast<T *>(st.x);
    const T *xb = static_cast<const T *>(st.xb);
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < st.n; t += gridDim.x * blockDim.x) x[t] = xb[t];
  }
  if (stop) return false;
  // delta = u.A.u = sigma^2 (sum rho' |J s.*z'|^2 + mu z'.D.z')
  const double den = dots[DEN], zdz = dots[ZDZ];
  const T delta = (T)((double)sigma * (double)sigma * (den + mu * zdz));
  stp.sigma = sigma;
  if (k == 0) { stp.beta = T(0); stp.alpha = gamma / delta; }
  else {
    const T gamma_prev = (T)st.scale[k - 1], alpha_prev = (T)st.beta[k - 1];
    stp.beta = gamma / gamma_prev;
    stp.alpha = gamma / (delta - stp.beta * gamma / alpha_prev);
  }
  if (first) { st.beta[k] = (double)stp.alpha; st.scale[k] = (double)gamma; }
  return true;
}
// the loop had already ended before iteration k: nothing to decide (every rank sees the same done[k - 1])
__device__ __forceinline__ bool pcg_cg_already_done(const PcgState &st, int k) {
  if (!(k > 0 && st.done[k - 1])) return false;
  if (blockIdx.x == 0 && threadIdx.x == 0) { st.done[k] = 1; st.rz0[k] = st.rz0[k - 1]; }
  return true;
}
template <typename T>
__device__ __forceinline__ bool pcg_cg_decide(const PcgState &st, int k, double mu, PcgCgStep<T> &stp) {
  stp.alpha = stp.beta = stp.sigma = T(0);
  if (pcg_cg_already_done(st, k)) return false;
  double dots[NSLOT];
  dots[RZP] = slot_sum(st.slots(k, RZP), 0); dots[RR] = slot_sum(st.slots(k, RR), 0); dots[PDZ] = 0.0;
  dots[ZDZ] = slot_sum(st.slots(k, ZDZ), 0); dots[DEN] = slot_sum(st.slots(k, DEN), 0);
  return pcg_cg_decide_v<T>(st, k, mu, stp, dots);
}
template <typename T> __global__ void __launch_bounds__(TPB) k_pcg_close_cg(PcgState st, int k, double mu) {
  PcgCgStep<T> stp;
  (void)pcg_cg_decide<T>(st, k, mu, stp);
}
// closes the loop after the last update of a solve that ran into its iteration cap (the decision an operator launch of
// iteration `k` would have taken)
template <typename T> __global__ void __launch_bounds__(TPB) k_pcg_close(PcgState st, int k) {
  PcgStep<T> stp;
  (void)pcg_decide<T>(st, k, stp);
}

// ===========================================================================
// Matrix-free PCG (PCGSolver, solver/pcg.hpp:61-232)
// ===========================================================================
// Operator (J^T rho' J) applied to ps = s .* p with J RECOMPUTED from the camera
// pack (the reference streams the stored J twice per iteration, pcg.hpp:143-163):
//   u = J ps, w = rho' u;   den = sum rho' |u|^2   (p.A.p = den + mu p.D.p: no pass over v2)
//   camera rows: Jc^T w reduced per (wave, camera) segment -> op_partial[seg][9]
//   point rows : per-observation Jp^T w                    -> g3[pm position][3]
// VAR (diagnostic builds only, GR_DIAG): 1 no g3 scatter, 2 no point gather, 4 no ps_l gather,
// 8 no Jacobian math, 16 no wave reduction.  VAR = 0 is the product kernel.
// Landmark shards, fused message (comm.hpp IpcFused): what the operator launch needs to finish the camera rows of this rank and
// push them, with the dot-product records of the iteration, into every peer's mailbox.
struct ShardPush {
  IpcFused fz;
  const int *cam_seg_ptr = nullptr; // [Nc + 1] segments of camera c
  const int *cam_wg = nullptr;      // [Nc] workgroups of THIS grid whose tile range holds observations of camera c (0: none on this shard)
  unsigned *cam_cnt = nullptr;      // [Nc] workgroups that have finished camera c so far (returns to 0)
  const int *empty = nullptr;       // cameras without observations on this shard: workgroup 0 pushes their zero rows
  int n_empty = 0;
  unsigned long long dots_off = 0;  // byte offset of the NSLOT dot sums inside a slot (behind the 9 Nc camera rows)
};
// sum of the NS partials of one dot-product record, read past this XCD's L2 (the partials were added by atomics of other XCDs)
__device__ __forceinline__ double slot_sum_agent(const double *base) { // whole wave must call
  return wave_allsum(__hip_atomic_load(&base[(size_t)(threadIdx.x & 63) * SS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// The tail of a FUSE operator launch, called by every workgroup (all threads) once its own tiles are done and its DEN partial
// is in the slots.  [c_lo, c_hi]: cameras of the workgroup's observation range (c_lo > c_hi: none).
//  1. per camera of the range an arrival counter; the workgroup that arrives LAST at a camera sums that camera's segment
//     partials in segment order (fixed: reproducible) and stores the 9 sums into slot `rank` of EVERY mailbox (system scope);
//  2. a launch-wide counter over cameras finished + workgroups finished; whoever completes it sums the rank's NSLOT dot-product
//     records, pushes them too, and raises this rank's flag in every mailbox with the new sequence number.
// Visibility inside the GPU: the segment partials are written through (agent-scope stores) and drained before the arrival
// counters are touched (relaxed tickets, cdna_hip_programming.md G16 form R1); across GPUs: write-through system-scope stores,
// drained by every thread, relaxed launch-wide counter, and ONE system fence in the workgroup that raises the flags.
template <typename T>
__device__ __forceinline__ void shard_push_tail(const ShardPush &sp, const PcgState &st, int k, int Nc, int c_lo, int c_hi, const T *__restrict__ op_partial, unsigned long long seq_before) {
  const IpcFused &fz = sp.fz;
  __shared__ int s_list[TPB];
  __shared__ int s_n;
  __shared__ unsigned s_last;
  __shared__ unsigned long long s_seq;
  if (threadIdx.x == 0) s_seq = seq_before + 1ull; // this launch's message (the count was loaded by thread 0 at the START of the launch: an uncached round trip off the tail)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads(); // every segment partial (and the DEN atomic) of this workgroup has left the CU
  const unsigned long long seq = s_seq;
  const int set = (int)(seq & 1ull);
  int pushed = 0;
  for (int base = c_lo; base <= c_hi; base += TPB) {
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int c = base + (int)threadIdx.x;
    if (c <= c_hi) {
      const int want = sp.cam_wg[c];
      if (want > 0) {
        const unsigned old = __hip_atomic_fetch_add(&sp.cam_cnt[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)old == want - 1) {
          __hip_atomic_store(&sp.cam_cnt[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_list[atomicAdd(&s_n, 1)] = c;
        }
      }
    }
    __syncthreads();
    const int nfin = s_n;
    for (int e = threadIdx.x; e < 9 * nfin; e += TPB) {
      const int cc = s_list[e / 9], i = e % 9;
      T row = T(0);
      { // uncached loads (another XCD's workgroup wrote them): four in flight, the last batch clamped, added in segment order
        int sg = sp.cam_seg_ptr[cc];
        const int sg1 = sp.cam_seg_ptr[cc + 1], last = sg1 - 1;
        for (; sg < sg1; sg += 4) {
          const int n = sg1 - sg;
          const T q0 = __hip_atomic_load(&op_partial[9 * (size_t)sg + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                  q1 = __hip_atomic_load(&op_partial[9 * (size_t)(sg + 1 < last ? sg + 1 : last) + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                  q2 = __hip_atomic_load(&op_partial[9 * (size_t)(sg + 2 < last ? sg + 2 : last) + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                  q3 = __hip_atomic_load(&op_partial[9 * (size_t)(sg + 3 < last ? sg + 3 : last) + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          row += q0;
          if (n > 1) row += q1;
          if (n > 2) row += q2;
          if (n > 3) row += q3;
        }
      }
      for (int r = 0; r < fz.size; ++r) ipc_store(reinterpret_cast<T *>(fz.slot(fz.push_box(r), set, fz.push_slot(r))) + 9 * (size_t)cc + i, fz.push_value(r, row));
    }
    pushed += nfin;
    __syncthreads();
  }
  { // the cameras this shard never sees: zero rows (the slot is reused every other message), an equal share per workgroup — with
    // points cut by camera locality (dist.py) 7/8 of the cameras of a shard are such, and one workgroup pushing all of them was
    // the longest part of the launch (Final-13682: 12 000 cameras x 9 rows x 8 peers)
    const int e0 = (int)((long long)sp.n_empty * blockIdx.x / gridDim.x), e1 = (int)((long long)sp.n_empty * (blockIdx.x + 1) / gridDim.x);
    if (!fz.contrib) // with contributor masks nobody reads this rank's slot for a camera it does not hold
    for (int e = 9 * e0 + (int)threadIdx.x; e < 9 * e1; e += TPB)
      for (int r = 0; r < fz.size; ++r) ipc_store(reinterpret_cast<T *>(fz.slot(fz.push_box(r), set, fz.push_slot(r))) + 9 * (size_t)sp.empty[e / 9] + e % 9, T(0));
    pushed += e1 - e0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    // no fence here: the rows went out as write-through system-scope stores and every thread drained its own (vmcnt) before the
    // barrier above; a release fence per WORKGROUP would write this kernel's dirty g3 lines back 768 times (measured: 200 us)
    const unsigned add = (unsigned)pushed + 1u, target = (unsigned)Nc + gridDim.x;
    const unsigned old = __hip_atomic_fetch_add(fz.counter, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (old + add == target) ? 1u : 0u;
    if (s_last) __hip_atomic_store(fz.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  // every camera row of this rank is in every mailbox and every workgroup's DEN partial is in the slots.  From here on ONE
  // workgroup runs alone: everything below is spread over its threads so that no lane walks a chain of uncached round trips
  __shared__ double s_d[NSLOT];
  __shared__ double s_g[NSLOT * 64];
  for (int q = (int)(threadIdx.x >> 6); q < NSLOT; q += TPB / 64) { // one wave per record
    const double v = slot_sum_agent(st.slots(k, q));
    if ((threadIdx.x & 63) == 0) s_d[q] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NSLOT * fz.size; e += TPB) {
    const int q = e / fz.size, r = e % fz.size;
    ipc_store(reinterpret_cast<double *>(fz.slot(fz.push_box(r), set, fz.push_slot(r)) + sp.dots_off) + q, fz.push_value(r, s_d[q]));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(fz.seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if ((int)threadIdx.x < fz.size) ipc_store(fz.flag(fz.push_box(threadIdx.x), set, fz.push_slot(threadIdx.x)), seq);
  // 3. the same workgroup — ONE, at the very end of the launch, while nothing else of this rank runs — waits (bounded) for every
  //    rank's message and leaves the dot records summed over the ranks (rank order) in record k of the slots: slot 0 the sum, the
  //    others zero, so that the update launch reads them exactly as it reads all-reduced slots.  (Waiting in the update launch
  //    instead cost 42 us: its 1 500 workgroups each paid three uncached round trips to the mailbox.)  The camera rows are summed
  //    over the ranks by the update launch's camera workgroups, straight from the mailbox.
  __shared__ int s_bad;
  // a communicator on which an earlier message timed out (the mailbox's error word, comm.hpp k_ipc_allreduce): no second wait —
  // only the FIRST failure pays the bound, every fused launch enqueued behind it finishes at once with NaN dots
  if (threadIdx.x == 0) s_bad = ipc_load(reinterpret_cast<const unsigned long long *>(fz.boxes[fz.rank]) + 500) != 0ull ? 1 : 0;
  __syncthreads();
  if ((int)threadIdx.x < fz.size && !s_bad) {
    const unsigned long long *flag = fz.flag(fz.rank, set, threadIdx.x);
    const long long t0 = wall_clock64();
    while (ipc_load(flag) < seq) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > fz.timeout_ticks) {
        ipc_store(reinterpret_cast<unsigned long long *>(fz.boxes[fz.rank]) + 500, 1ull); // the mailbox's error word (comm.hpp)
        if (fz.h_err) { *fz.h_err = 1; __threadfence_system(); }
        s_bad = 1;
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  for (int e = threadIdx.x; e < NSLOT * fz.size; e += TPB) // one load per thread, summed below in rank order
    s_g[e] = ipc_load(reinterpret_cast<const double *>(fz.slot(fz.rank, set, e % fz.size) + sp.dots_off) + e / fz.size);
  __syncthreads();
  for (int e = threadIdx.x; e < NSLOT * NS; e += TPB) {
    const int q = e / NS, sl = e % NS;
    double v = 0.0;
    if (sl == 0) {
      // a peer that never arrived: NaN dots end this rank's PCG loop by the rejection test; the host finds the error word
      if (s_bad) v = __builtin_nan("");
      else for (int r = 0; r < fz.size; ++r) v += s_g[q * fz.size + r];
    }
    st.slots(k, q)[(size_t)sl * SS] = v;
  }
}

// REC: xp records (compile time: three 16-byte loads per lane; a run-time test makes hipcc merge both layouts into five).
// (Round 4, measured and removed — a MULTI form for orders whose waves mostly hold two cameras (point-tiled Final-13682: 66-observation
// runs; landmark shards): Jacobian and products once per lane from per-lane camera data, only the 9-value reduction per distinct
// camera.  Final-13682 774 -> 932 us, Venice-1778 83.5 -> 90.4 us: the 21 extra vector loads per lane cost more than the second
// evaluation from SGPRs.)
template <typename T, int VAR = 0, typename JT = T, int LAZY = 0, bool FUSE = false, bool REC = false>
__global__ void __launch_bounds__(TPB, OP_WAVES)
k_pcg_operator(int No, int Nc, int ntiles, const int *__restrict__ cam_cm, const int *__restrict__ pt_cm,
               const int *__restrict__ pos_cm, const T *__restrict__ obs_cm, const int *__restrict__ blk_seg,
               const int *__restrict__ seg_slot, const T *__restrict__ pts, const T *__restrict__ pack,
               int loss_kind, T loss_delta, const T *__restrict__ ps, T *__restrict__ g3,
               T *__restrict__ op_partial, double mu, PcgState st, int k, const T *__restrict__ xp = nullptr,
               const LmDev *__restrict__ lm = nullptr, ShardPush sp = ShardPush{}) {
  if (lm && lm->stop) return;
  PcgStep<T> stp{T(0), T(0)};
  if (LAZY == 2) { // single-reduction form: A applied to the un-normalised z' (the update kernel scales); decisions are the update kernel's
    if (k > 0 && st.done[k - 1]) return;
    stp.scale = T(1);
  } else if (LAZY) {
    if (k == 0 && st.ts && blockIdx.x == 0 && threadIdx.x == 0) st.ts[0] = wall_clock64();
    if (!pcg_decide<T>(st, k, stp)) return;
    if (slot_sum(st.slots(k, RZP), 0) == 0.0) return; // rz == 0: the next launch's decision closes the loop
  } else if (!(VAR & 32)) {
    if (k == 0 && st.ts_op && st.ts && blockIdx.x == 0 && threadIdx.x == 0) st.ts[0] = wall_clock64();
    if (st.done[k]) return;                      // direction(k-1) already told the host
    // rz == 0: the direction kernel of this iteration closes the loop.  (Iteration 0 of the first-lazy form: the dots are
    // still per-workgroup partials — summed below — and a zero r.z' only makes the update kernel return.)
    if (!(k == 0 && st.ts_op && st.part0) && slot_sum(st.slots(k, RZP), 0) == 0.0) return;
  }
  __shared__ double red[4];
  unsigned long long fuse_seq = 0; // FUSE: messages pushed so far (only this launch's last workgroup changes it, after every workgroup has read it)
  if (FUSE && threadIdx.x == 0) fuse_seq = __hip_atomic_load(sp.fz.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (LAZY == 0 && k == 0 && st.ts_op && st.part0 && blockIdx.x == gridDim.x - 1) {
    // PcgState::part0: the PCG start's dots, one partial per workgroup of k_finalize_bj, summed here in workgroup order
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      double v = 0;
      for (int i = threadIdx.x; i < st.n_part0; i += TPB) v += st.part0[(size_t)q * st.n_part0 + i];
      v = block_sum_256(v, red);
      if (threadIdx.x == 0) st.slots(0, q == 0 ? RR : q == 1 ? RZP : ZDZ)[0] = v;
    }
    __syncthreads();
  }
  using V2 = typename Vec2T<T>::type;
  const int lane = threadIdx.x & 63;
  const size_t pose_dim = 9 * (size_t)Nc;
  // lazy direction: ps holds s.*p_{k-1}; the direction of this iteration is beta ps + scale zs (k == 0: scale zs)
  constexpr bool lazy = LAZY != 0;
  const bool lazy_old = LAZY == 1 && k > 0;
  const T *zs = static_cast<const T *>(st.zs);
  const T lz_beta = stp.beta, lz_scale = stp.scale;
  int j0, jstride, niter, jlim;
  xcd_obs_range(ntiles, No, j0, jstride, niter, jlim);
  double den = 0;
  // One 64-observation tile of a wave, given its index streams and its gathered point record: wave-uniform camera data (24-scalar
  // pack, 9 direction scalars, segment id) are fetched per DISTINCT camera of the wave through a uniform index, i.e. with scalar
  // loads into SGPRs: the kernel keeps its VGPRs for the per-observation state.  Almost every wave has one camera.
  auto tile_body = [&](const int j, const bool valid, const int c, const size_t a, const V2 o,
                       const T X, const T Y, const T Z, const T pl0, const T pl1, const T pl2) __attribute__((always_inline)) {
    unsigned long long remaining = __ballot(valid);
    int segf = blk_seg[__builtin_amdgcn_readfirstlane(j >> 6)];
    while (remaining) {
      const int leader = __builtin_ctzll(remaining);
      const int cl = __builtin_amdgcn_readlane(c, leader); // leader is wave-uniform (from a ballot): v_readlane, not ds_bpermute
      const bool mine = valid && c == cl;
      const int segl = (VAR & 64) ? (j >> 6) : seg_slot[segf++];
      T pk[PACK], pc[9];
      if (VAR & 64) {
#pragma unroll
        for (int i = 0; i < PACK; ++i) pk[i] = (T)(cl + i) * T(1e-3);
#pragma unroll
        for (int i = 0; i < 9; ++i) pc[i] = (T)(cl - i) * T(1e-3);
      } else {
        load_pack(pack, cl, pk);
        if (!lazy) {
#pragma unroll
          for (int i = 0; i < 9; ++i) pc[i] = ps[9 * (size_t)cl + i];
        } else {
#pragma unroll
          for (int i = 0; i < 9; ++i) pc[i] = lz_scale * zs[9 * (size_t)cl + i];
          if (lazy_old) {
#pragma unroll
            for (int i = 0; i < 9; ++i) pc[i] += lz_beta * ps[9 * (size_t)cl + i];
          }
        }
      }
      T e0, e1, Jc[18], Jp[6];
      if (VAR & 8) {
        e0 = o.x; e1 = o.y;
#pragma unroll
        for (int i = 0; i < 18; ++i) Jc[i] = X + pk[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) Jp[i] = Y - pk[i];
      } else
      bal_linearize_j<T, JT>(pk, X, Y, Z, o.x, o.y, e0, e1, Jc, Jp);
      const T w = mine ? loss_drho(loss_kind, loss_delta, e0 * e0 + e1 * e1) : T(0);
      T u0 = Jp[0] * pl0 + Jp[2] * pl1 + Jp[4] * pl2;
      T u1 = Jp[1] * pl0 + Jp[3] * pl1 + Jp[5] * pl2;
#pragma unroll
      for (int i = 0; i < 9; ++i) { u0 += Jc[2 * i] * pc[i]; u1 += Jc[2 * i + 1] * pc[i]; }
      if (mine) den += (double)(w * (u0 * u0 + u1 * u1));
      u0 *= w; u1 *= w;
      T m[16];
#pragma unroll
      for (int i = 0; i < 9; ++i) m[i] = mine ? Jc[2 * i] * u0 + Jc[2 * i + 1] * u1 : T(0);
#pragma unroll
      for (int i = 9; i < 16; ++i) m[i] = T(0);
      if (mine) {
        T *g = g3 + 3 * ((VAR & 1) ? (size_t)j : a);
        g[0] = Jp[0] * u0 + Jp[1] * u1;
        g[1] = Jp[2] * u0 + Jp[3] * u1;
        g[2] = Jp[4] * u0 + Jp[5] * u1;
      }
      if (VAR & 16) { den += (double)(m[0] + m[8]); }
      else {
      const T tot = wave_transpose_sum<T, 16>(m, lane);
      if ((lane & 3) == 0 && (lane >> 2) < 9) {
        if (FUSE) __hip_atomic_store(&op_partial[9 * (size_t)segl + (lane >> 2)], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // read by another XCD's workgroup (shard_push_tail)
        else op_partial[9 * (size_t)segl + (lane >> 2)] = tot;
      }
      }
      remaining &= ~__ballot(mine);
    }
  };
  // (Round 4, measured and removed: index streams two tiles and the point gather one tile ahead of the arithmetic — +19 VGPRs, exact
  // vmcnt waits, no spill: Ladybug-1723 19.55 -> 19.47 us with records, 19.85 -> 20.4 us without; Final-13682 723 -> 715 / 868 -> 919 us.
  // The kernel is bound by the NUMBER of divergent memory instructions (64 lines each), not by the latency of the chain:
  // the record layout as a compile-time parameter — three 16-byte loads instead of the five hipcc emits for the run-time test —
  // gave 782 -> 723 us on Final-13682.)
  struct Gat { T X, Y, Z, p0, p1, p2; };
  auto gather_plain = [&](const size_t lp) __attribute__((always_inline)) {
    Gat g;
    if constexpr (REC) { // one aligned record: [X Y Z ps_x ps_y ps_z . .]
      const V2 *rec = reinterpret_cast<const V2 *>(xp + 8 * lp);
      const V2 r0 = rec[0], r1 = rec[1], r2 = rec[2];
      g.X = r0.x; g.Y = r0.y; g.Z = r1.x; g.p0 = r1.y; g.p1 = r2.x; g.p2 = r2.y;
    } else {
      g.X = pts[3 * lp]; g.Y = pts[3 * lp + 1]; g.Z = pts[3 * lp + 2];
      const T *pl = ps + pose_dim + 3 * lp;
      g.p0 = pl[0]; g.p1 = pl[1]; g.p2 = pl[2];
    }
    return g;
  };
  int j = j0 + threadIdx.x;
  bool valid = niter > 0 && j < jlim;
  int c_n = -1, l_n = 0, a_n = 0;
  V2 o_n{};
  if (valid) { c_n = cam_cm[j]; l_n = pt_cm[j]; a_n = pos_cm ? pos_cm[j] : j; o_n = reinterpret_cast<const V2 *>(obs_cm)[j]; }
  for (int it = 0; it < niter; ++it) {
    const int c = c_n, l = l_n;
    const size_t a = (size_t)a_n;
    const V2 o = o_n;
    const int jn = j + jstride;
    const bool validn = (it + 1 < niter) && jn < jlim;
    if (validn) { c_n = cam_cm[jn]; l_n = pt_cm[jn]; a_n = pos_cm ? pos_cm[jn] : jn; o_n = reinterpret_cast<const V2 *>(obs_cm)[jn]; }
    const size_t lp = (VAR & 2) ? (size_t)(j & 1023) : (size_t)(valid ? l : 0);
    if constexpr (!lazy) {
      const Gat g = gather_plain(lp);
      tile_body(j, valid, c, a, o, g.X, g.Y, g.Z, g.p0, g.p1, g.p2);
    } else if constexpr (LAZY == 2 && REC) { // single-reduction form on records: [X Y Z | zs] (the un-normalised s .* z'; lz_scale = 1)
      const Gat g = gather_plain(lp);
      tile_body(j, valid, c, a, o, g.X, g.Y, g.Z, lz_scale * g.p0, lz_scale * g.p1, lz_scale * g.p2);
    } else { // lazy direction: formed from zs (and the previous ps) while it is gathered
      const T X = pts[3 * lp], Y = pts[3 * lp + 1], Z = pts[3 * lp + 2];
      const T *pl = ps + pose_dim + 3 * lp;
      const T *zl = zs + pose_dim + 3 * lp;
      T pl0 = lz_scale * zl[0], pl1 = lz_scale * zl[1], pl2 = lz_scale * zl[2];
      if (lazy_old) { pl0 += lz_beta * pl[0]; pl1 += lz_beta * pl[1]; pl2 += lz_beta * pl[2]; }
      tile_body(j, valid, c, a, o, X, Y, Z, pl0, pl1, pl2);
    }
    valid = validn;
    j = jn;
  }
  if (VAR & 128) { if (den == 1.2345) st.pdp[0] = den; return; }
  den = block_sum_256(den, red);
  if (threadIdx.x == 0) slot_add(st.slots(k, DEN), 0, den);
  if (FUSE) {
    // plain camera-major order (the fused form is not used with point tiles): the workgroup's observations are one contiguous
    // range, its cameras the range between the first and the last one's
    int c_lo = 0, c_hi = -1;
    if (niter > 0 && j0 < jlim) { c_lo = cam_cm[j0]; c_hi = cam_cm[jlim - 1]; }
    shard_push_tail<T>(sp, st, k, Nc, c_lo, c_hi, op_partial, fuse_seq);
  }
}

// g3 kept in OBSERVATION order (the operator's stores are then whole lines: a wave writes 64 consecutive 3-vectors) and
// gathered per point by the update kernel through gidx (pm position -> observation-order position).  The update kernel
// then walks the points as the operator walked their observations: point tile q (points [ptile_ptr[q], ptile_ptr[q+1]))
// belongs to XCD q % 8 and that XCD's workgroups sweep it together, so the lines one XCD wrote are read back by the
// same XCD while they are still in its L2.  gidx == nullptr: g3 in pm order, points in contiguous ranges per workgroup.
struct G3Gather { const int *gidx = nullptr; const int *ptile_ptr = nullptr; int n_ptiles = 0;
                  // FIXED vertices (vertex.hpp:262 set_fixed; the reference's kernels skip their Jacobian blocks, ops/linearize.hpp:24,
                  // ops/hessian.hpp:95): their rows of every operator product are dropped here.  nullptr: none fixed.
                  const unsigned char *cam_fixed = nullptr, *pt_fixed = nullptr; };

// x / r / z' update of the matrix-free PCG.
// Persistent blocks walk contiguous ranges of
//   camera tiles: 252 camera scalars (28 cameras) — fixed-order sum of the segment partials,
//                 9x9 block-Jacobi through LDS
//   point tiles : 255 point scalars (85 points) — fixed-order sum of the observations' g3 per scalar,
//                 3x3 block-Jacobi through LDS
// MODE 0 (init): r = s .* b^u, x = 0.      MODE 1: v2 = s .* sums + mu d .* p,
//   x_backup = x; x += alpha p; r -= alpha v2.
// z' = Minv r; accumulates rr = r.r and rzp = r.z' (the reference applies the preconditioner to
// r/||r||; Minv is linear so z = z'/||r|| and r.z = rzp/||r||).
template <typename T, int MODE, bool IDENTITY, int LAZY = 0>
__global__ void __launch_bounds__(TPB)
k_pcg_update(int Nc, int Np, const T *__restrict__ bu, const T *__restrict__ scales, T *__restrict__ x,
             T *__restrict__ xb, T *__restrict__ r, T *__restrict__ zt, const T *__restrict__ p,
             const T *__restrict__ g3, const int *__restrict__ pt_ptr, const T *__restrict__ op_partial,
             const int *__restrict__ cam_seg_ptr, const T *__restrict__ raw_c, int cam_weight,
             const T *__restrict__ diag, double mu, int use_identity,
             const T *__restrict__ MinvC, const T *__restrict__ MinvP, PcgState st, int k,
             const LmDev *__restrict__ lm = nullptr, G3Gather gg = G3Gather{}, IpcFused fz = IpcFused{}, unsigned long long dots_off = 0) {
  if (lm) { if (lm->stop) return; mu = lm->mu; }
  T alpha = 0, first_sigma = 0;
  constexpr bool CG = (LAZY == 2) && MODE == 1;
  PcgCgStep<T> cg{T(0), T(0), T(0)};
  // landmark shards, fused message: the operator launch has pushed this rank's camera rows, waited for every peer's and left the
  // dot records summed over the ranks in the slots; the camera rows are summed here, by the camera workgroups, from the mailbox
  const bool fused = CG && fz.boxes != nullptr;
  (void)dots_off;
  if (CG) {
    if (!pcg_cg_decide<T>(st, k, mu, cg)) return;
    alpha = cg.alpha;
  } else if (MODE == 1) {
    if (st.done[k]) return;
    // the dot products of this iteration in one go (slot_sums): r.z', r.r, den (and z'.D.z' for the first iteration)
    const double *const sl[4] = {st.slots(k, RZP), st.slots(k, RR), st.slots(k, DEN), st.slots(k, ZDZ)};
    double sv[4];
    slot_sums<4>(sl, sv);
    PcgIter it;
    it.rzp = sv[0]; it.rscale = 1.0 / sqrt(sv[1]); it.rz = it.rzp * it.rscale;
    if (it.rzp == 0.0) return;
    // T-precision scalars, as the reference keeps them in T on the host;  p.A.p = den + mu p.D.p
    if (LAZY == 3) {
      // first iteration without a direction launch: the operator ran on the UN-normalised direction s .* z' (it is linear:
      // A (sigma z') = sigma A z'), sigma = 1 / |r| is applied here: p.A.p = sigma^2 (den~ + mu z'.D.z')
      first_sigma = (T)(1.0 / (double)(T)sqrt((double)(T)sv[1]));
      const double pdp0 = (double)first_sigma * (double)first_sigma * sv[3];
      if (blockIdx.x == 0 && threadIdx.x == 0) { st.pdp[0] = pdp0; st.beta[0] = 0.0; st.scale[0] = (double)first_sigma; }
      alpha = (T)it.rz / (T)((double)first_sigma * (double)first_sigma * sv[2] + mu * pdp0);
    } else
    alpha = (T)it.rz / (T)(sv[2] + mu * st.pdp[k]);
  }
  __shared__ double red[4];
  __shared__ T rs[TPB];
  const unsigned pose_dim = 9u * (unsigned)Nc;
  // lazy direction (PcgState): p_k = beta p_{k-1} + scale z'_k is formed here, stored with ps = s.*p_k; zs = s.*z'_{k+1}
  // LAZY == 3: the FIRST iteration of the direction-kernel form without its first direction launch — the operator of iteration 0
  // ran (in its plain form) on the un-normalised s .* z'_0 that k_finalize_bj left in zs; sigma = 1 / |r_0| is applied here, where
  // p_0 = sigma z'_0 is formed and stored; x is known to be 0 and is not read; from the direction launch of iteration 0 on, the
  // loop is the direction-kernel form
  constexpr bool lazy = LAZY != 0;
  constexpr bool FIRST = LAZY == 3;
  const bool lazy_old = lazy && MODE == 1 && k > 0; // a previous direction exists
  T *pw = const_cast<T *>(p), *psw = static_cast<T *>(st.ps), *zsw = static_cast<T *>(st.zs), *svw = static_cast<T *>(st.sv);
  const T lz_beta = CG ? cg.beta : (lazy_old ? (T)st.beta[k] : T(0)), lz_scale = CG ? cg.sigma : FIRST ? first_sigma : ((lazy && MODE == 1) ? (T)st.scale[k] : T(0));
  const int cam_tiles = (int)((pose_dim + 251u) / 252u), pt_tiles = (Np + 84) / 85;
  double prr = 0, prz = 0, ppz = 0, pzz = 0;
  const double cw = (double)cam_weight;
  // persistent: every block walks a contiguous range of camera tiles, then of point tiles
  const int ct0 = (int)((long long)blockIdx.x * cam_tiles / gridDim.x), ct1 = (int)((long long)(blockIdx.x + 1) * cam_tiles / gridDim.x);
  int fz_set = 0;
  if (fused && ct0 < ct1) { // block-uniform: the message the operator launch of this iteration pushed
    __shared__ int s_set;
    if (threadIdx.x == 0) s_set = (int)(__hip_atomic_load(fz.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1ull);
    __syncthreads();
    fz_set = s_set;
  }
  for (int tile = ct0; tile < ct1; ++tile) {
    const unsigned t = (unsigned)tile * 252u + threadIdx.x;
    const bool on = threadIdx.x < 252 && t < pose_dim;
    T rn = 0, pvk = 0;
    if (on) {
      if (MODE == 0) { rn = scales[t] * bu[t]; x[t] = T(0); }
      else {
        const unsigned c = t / 9u, i = t % 9u;
        T raw = 0;
        if (fused) { // the ranks' camera rows, summed here in rank order (the same bits on every rank)
          const unsigned who = fz.contributors(c);
          for (int r0 = 0; r0 < fz.size; r0 += 8) { // eight uncached loads in flight instead of a chain of them; added in rank order
            T q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = (r0 + u < fz.size && IpcFused::rank_in(who, r0 + u)) ? ipc_load(reinterpret_cast<const T *>(fz.slot(fz.rank, fz_set, r0 + u)) + t) : T(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) if (r0 + u < fz.size && IpcFused::rank_in(who, r0 + u)) raw += q[u];
          }
        } else if (raw_c) raw = raw_c[t]; // multi-GPU: camera rows already summed over segments and ranks
        else
        {
          // the workgroups that hold a camera tile start their point share after it: the segment sums are on the launch's critical
          // path — sixteen loads in flight (round 6; four before: a Venice camera has 44 segments = 11 dependent rounds), lanes past
          // the camera's last segment issue nothing, added in segment order
          int sg = cam_seg_ptr[c];
          const int sg1 = cam_seg_ptr[c + 1];
          for (; sg < sg1; sg += 16) {
            T q[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) q[u] = sg + u < sg1 ? op_partial[9 * (size_t)(sg + u) + i] : T(0);
#pragma unroll
            for (int u = 0; u < 16; ++u) if (sg + u < sg1) raw += q[u];
          }
        }
        if (gg.cam_fixed && gg.cam_fixed[c]) raw = T(0);
        T pv;
        if (!lazy) pv = p[t];
        else {
          pv = lz_scale * zt[t];
          if (lazy_old) pv += lz_beta * p[t];
          pw[t] = pv; if (!CG && !FIRST) psw[t] = scales[t] * pv;
        }
        pvk = pv;
        T v2;
        if (CG) { // raw = rows of A' z' (un-normalised): w = sigma s raw + mu d u, s_k = w + beta s_{k-1}
          const T uo = lz_scale * zt[t];
          v2 = lz_scale * (scales[t] * raw) + (use_identity ? (T)mu * uo : (T)mu * diag[t] * uo);
          if (k > 0) v2 += lz_beta * svw[t];
          svw[t] = v2;
        } else if (FIRST) v2 = lz_scale * (scales[t] * raw) + (use_identity ? (T)mu * pv : (T)mu * diag[t] * pv);
        else
        v2 = scales[t] * raw + (use_identity ? (T)mu * pv : (T)mu * diag[t] * pv);
        const T xo = FIRST ? T(0) : x[t];
        xb[t] = xo;
        x[t] = alpha * pv + xo;
        rn = -alpha * v2 + r[t];
      }
      r[t] = rn;
    }
    __syncthreads();
    rs[threadIdx.x] = rn;
    __syncthreads();
    if (on) {
      T s = 0;
      if (IDENTITY) s = rn;
      else {
        const T *M = MinvC + 81 * (size_t)(t / 9u);
        const int row = (int)(t % 9u);
        const T *rc = rs + (threadIdx.x / 9) * 9;
#pragma unroll
        for (int q = 0; q < 9; ++q) s += M[row + 9 * q] * rc[q];
      }
      zt[t] = s;
      if (lazy && !FIRST) zsw[t] = scales[t] * s;
      const T d = use_identity ? T(1) : diag[t];
      const T pv = pvk;
      prr += cw * (double)(rn * rn);
      prz += cw * (double)(rn * s);
      ppz += cw * (double)(d * pv * s);
      pzz += cw * (double)(d * s * s);
    }
  }
  // point tiles: 85 points = 255 scalars per tile, one thread per SCALAR so that every load and store is a
  // contiguous run over the wave (a thread per point reads 3-scalar groups 24 bytes apart: three times the
  // address-processing work for the same bytes); the 3 x 3 block-Jacobi goes through LDS like the camera part
  const unsigned lt = threadIdx.x / 3u, li = threadIdx.x % 3u;
  const bool sweep = gg.n_ptiles > 0; // gridDim.x is then a multiple of 8
  const int pt0 = sweep ? 0 : (int)((long long)blockIdx.x * pt_tiles / gridDim.x), pt1 = sweep ? 0 : (int)((long long)(blockIdx.x + 1) * pt_tiles / gridDim.x);
  int q = (int)(blockIdx.x & 7), tile = sweep ? (int)(blockIdx.x >> 3) : pt0;
  unsigned P0 = 0, P1 = (unsigned)Np;
  if (sweep && q < gg.n_ptiles) { P0 = (unsigned)gg.ptile_ptr[q]; P1 = (unsigned)gg.ptile_ptr[q + 1]; }
  // the tile walk, one step: first point of the workgroup's next tile and the end of the range it lies in; false: no tile left
  auto next_tile = [&](unsigned &lbase, unsigned &lend) __attribute__((always_inline)) -> bool {
    if (sweep) {
      while (q < gg.n_ptiles && P0 + 85u * (unsigned)tile >= P1) { // next point tile of this XCD
        q += 8; tile = (int)(blockIdx.x >> 3);
        if (q < gg.n_ptiles) { P0 = (unsigned)gg.ptile_ptr[q]; P1 = (unsigned)gg.ptile_ptr[q + 1]; }
      }
      if (q >= gg.n_ptiles) return false;
    } else if (tile >= pt1) return false;
    lbase = P0 + (unsigned)tile * 85u; lend = P1;
    tile += sweep ? (int)(gridDim.x >> 3) : 1;
    return true;
  };
  // (round 6) the run bounds of the NEXT tile's points are requested while this tile is worked on: they are what the dependent chain
  // of a tile starts with (bounds -> [indices ->] rows); with them at hand the rows are requested together with the tile's vectors
  unsigned lb_n = 0, le_n = 0;
  bool have_n = next_tile(lb_n, le_n);
  int a_n = 0, ae_n = 0;
  if (have_n && MODE == 1) { const unsigned ln = lb_n + lt; if (threadIdx.x < 255 && ln < le_n) { a_n = pt_ptr[ln]; ae_n = pt_ptr[ln + 1]; } }
  while (have_n) {
    const unsigned l = lb_n + lt;
    const bool on = threadIdx.x < 255 && l < le_n;
    const int a_cur = a_n, ae_cur = ae_n;
    have_n = next_tile(lb_n, le_n);
    a_n = ae_n = 0;
    if (have_n && MODE == 1) { const unsigned ln = lb_n + lt; if (threadIdx.x < 255 && ln < le_n) { a_n = pt_ptr[ln]; ae_n = pt_ptr[ln + 1]; } }
    const size_t t = (size_t)pose_dim + 3 * (size_t)l + li;
    T rn = 0, pv = 0, m0 = 0, m1 = 0, m2 = 0, dg = T(1);
    if (on) {
      // everything that does not depend on the observation range is requested first, so the kernel pays
      // two memory round trips (indices + vectors, then the g3 run) instead of one per stage
      const T sc = scales[t];
      if (!IDENTITY) { const T *M = MinvP + 9 * (size_t)l; m0 = M[li]; m1 = M[li + 3]; m2 = M[li + 6]; }
      if (!use_identity) dg = diag[t];
      if (MODE == 0) { rn = sc * bu[t]; x[t] = T(0); }
      else {
        if (!lazy) pv = p[t];
        else {
          pv = lz_scale * zt[t];
          if (lazy_old) pv += lz_beta * p[t];
          pw[t] = pv; if (!CG && !FIRST) psw[t] = sc * pv;
        }
        const T xo = FIRST ? T(0) : x[t], ro = r[t];
        double raw = 0; // summed in double whatever T is (a point's rows cancel; T = double: unchanged)
        int a = a_cur;
        const int a_end = ae_cur;
#if UPD_VAR != 1
        // A point's run of observation rows, summed in observation order, EIGHT requests in flight per round (round 6): the sum is a
        // chain of dependent round trips — with four per round a point of five to eight observations (most waves hold one) cost
        // two rounds of values (and, gathered, two of indices before them); the launch is latency-bound (Venice-1778 fp32: 77 % of
        // wave-cycles parked on memory at 1.04 x algorithmic traffic), so the rounds are what counts.  Lanes past their run's end
        // issue nothing (predicated), the additions keep their order.
        if (gg.gidx) { // the slots sit where the operator's wave wrote them
          for (; a < a_end; a += 8) {
            int jx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) jx[u] = a + u < a_end ? gg.gidx[a + u] : -1;
            T gq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) gq[u] = jx[u] >= 0 ? g3[3 * (size_t)jx[u] + li] : T(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) if (jx[u] >= 0) raw += gq[u];
          }
        } else {
          for (; a < a_end; a += 8) {
            const T *gp = g3 + 3 * (size_t)a + li;
            T gq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) gq[u] = a + u < a_end ? gp[3 * u] : T(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) if (a + u < a_end) raw += gq[u];
          }
        }
#else
        raw = (double)(a_end - a);
#endif
        if (gg.pt_fixed && gg.pt_fixed[l]) raw = 0.0;
        const T rawt = (T)raw;
        T v2;
        if (CG) {
          const T uo = lz_scale * zt[t];
          v2 = lz_scale * (sc * rawt) + (use_identity ? (T)mu * uo : (T)mu * dg * uo);
          if (k > 0) v2 += lz_beta * svw[t];
          svw[t] = v2;
        } else if (FIRST) v2 = lz_scale * (sc * rawt) + (use_identity ? (T)mu * pv : (T)mu * dg * pv);
        else
        v2 = sc * rawt + (use_identity ? (T)mu * pv : (T)mu * dg * pv);
        xb[t] = xo;
        x[t] = alpha * pv + xo;
        rn = -alpha * v2 + ro;
      }
      r[t] = rn;
    }
#if UPD_VAR != 3
    __syncthreads();
    rs[threadIdx.x] = rn;
    __syncthreads();
#endif
    if (on) {
      T s;
      if (IDENTITY) s = rn;
      else {
#if UPD_VAR != 3
        const T *rc = rs + 3 * lt;
        s = (T)((double)m0 * (double)rc[0] + (double)m1 * (double)rc[1] + (double)m2 * (double)rc[2]);
#else
        s = m0 * rn + m1 * rn + m2 * rn;
#endif
      }
      zt[t] = s;
      if (lazy && !FIRST) {
        const T zv = scales[t] * s;
        zsw[t] = zv;
        if (LAZY == 2 && st.zrec) static_cast<T *>(st.zrec)[8 * (size_t)l + 3 + li] = zv; // the operator's [X Y Z | zs] record
      }
      prr += (double)(rn * rn);
      prz += (double)(rn * s);
      ppz += (double)(dg * pv * s);
      pzz += (double)(dg * s * s);
    }
  }
  const int slot = (MODE == 0) ? 0 : k + 1;
#if UPD_VAR == 2
  if (MODE == 1) { if (prr + prz + ppz + pzz == 1.2345) st.pdp[0] = prr; return; }
#endif
  // the four dots in one block reduction (one barrier pair instead of four)
  __shared__ double red4[4][4];
  prr = wave_sum(prr); prz = wave_sum(prz); ppz = wave_sum(ppz); pzz = wave_sum(pzz);
  if ((threadIdx.x & 63) == 0) { double *q = red4[threadIdx.x >> 6]; q[0] = prr; q[1] = prz; q[2] = ppz; q[3] = pzz; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const double v = red4[0][threadIdx.x] + red4[1][threadIdx.x] + red4[2][threadIdx.x] + red4[3][threadIdx.x];
    const int which = threadIdx.x == 0 ? RR : threadIdx.x == 1 ? RZP : threadIdx.x == 2 ? PDZ : ZDZ;
    slot_add(st.slots(slot, which), 0, v);
  }
}

// Trial step fused into the direction kernel that ENDS the PCG loop (LM loop, direction-kernel form): once every workgroup
// has found that the loop is over (tolerance, rejection, r.z == 0, or the iteration cap: at_cap), x is final and the launch
// turns into Graph::backup_parameters + Graph::apply_update (graph.hpp:292-309, ops/update.hpp:11-31) + this block's share of
// the compute_rho denominator sum dx (mu dx + b) (levenberg_marquardt.hpp:34-41) + the camera packs of the moved cameras —
// the work of k_apply_update_rho without its launch.  Workgroups [0, nct) take 28 cameras each, the others the point scalars.
template <typename T> struct ApplyOnExit {
  T *cams = nullptr, *pts = nullptr, *cams_bak = nullptr, *pts_bak = nullptr;
  const T *bu = nullptr;
  double *rho_partial = nullptr; // [gridDim.x]
  T *pack = nullptr, *xp = nullptr;
  int cam_weight = 1, at_cap = 0;
  long long *ts = nullptr;       // PcgState::ts
};
template <typename T>
__device__ __forceinline__ void apply_on_exit(const ApplyOnExit<T> &ap, unsigned n, unsigned pose_dim, const T *__restrict__ dx,
                                              const T *__restrict__ scales, double mu, T *__restrict__ x_restore) {
  __shared__ double red[4];
  __shared__ T cs[252];
  if (ap.ts && blockIdx.x == 0 && threadIdx.x == 0) ap.ts[1] = wall_clock64();
  double rho = 0;
  const unsigned nct = (pose_dim + 251u) / 252u;
  if (blockIdx.x < nct) {
    const unsigned i = blockIdx.x * 252u + threadIdx.x;
    if (threadIdx.x < 252 && i < pose_dim) {
      const T d = dx[i], s = scales[i], xo = ap.cams[i];
      if (x_restore) x_restore[i] = d;
      ap.cams_bak[i] = xo;
      const T xn = xo + d * s;
      ap.cams[i] = xn;
      cs[threadIdx.x] = xn;
      if (ap.cam_weight) rho = (double)(d * ((T)mu * d + s * ap.bu[i]));
    }
    __syncthreads();
    const unsigned c = blockIdx.x * 28u + threadIdx.x;
    if (threadIdx.x < 28 && 9u * c < pose_dim) {
      T cam[9], pk[PACK];
#pragma unroll
      for (int k = 0; k < 9; ++k) cam[k] = cs[9 * threadIdx.x + k];
      make_campack(cam, pk);
#pragma unroll
      for (int k = 0; k < PACK; ++k) ap.pack[PACK * (size_t)c + k] = pk[k];
    }
  } else {
    const unsigned npt = n - pose_dim, stride = (gridDim.x - nct) * TPB;
    for (unsigned q = (blockIdx.x - nct) * TPB + threadIdx.x; q < npt; q += stride) {
      const unsigned i = pose_dim + q;
      const T d = dx[i], s = scales[i], xo = ap.pts[q];
      if (x_restore) x_restore[i] = d;
      ap.pts_bak[q] = xo;
      const T xn = xo + d * s;
      ap.pts[q] = xn;
      if (ap.xp) ap.xp[8 * (size_t)(q / 3u) + q % 3u] = xn; // operator's point records
      rho += (double)(d * ((T)mu * d + s * ap.bu[i]));
    }
  }
  rho = block_sum_256(rho, red);
  if (threadIdx.x == 0) ap.rho_partial[blockIdx.x] = rho;
}

// Direction kernel (pcg.hpp:108-127 for k = -1, :184-217 otherwise): rejection test, restore x
// on a rejected step, else p = beta p + z'/||r||; ps = s .* p.  Thread 0 publishes the loop state
// of iteration k+1 (device) and the host flag of iteration k;  p.D.p of the new direction follows
// from the dots the update kernel took:  pdp' = beta^2 pdp + 2 beta sigma p.D.z' + sigma^2 z'.D.z'.
// ap.cams != nullptr (LM loop): the launch that ends the loop applies the step (ApplyOnExit); gridDim.x must then exceed
// the number of camera tiles.
template <typename T>
__global__ void __launch_bounds__(TPB)
k_pcg_direction(unsigned n, T *__restrict__ x, const T *__restrict__ xb, T *__restrict__ p,
                T *__restrict__ ps, const T *__restrict__ zt, const T *__restrict__ scales, PcgState st,
                int k, double tol, double rejection_ratio, unsigned pose_dim = 0, T *__restrict__ xp = nullptr,
                const LmDev *__restrict__ lm = nullptr, double mu = 0.0, ApplyOnExit<T> ap = ApplyOnExit<T>{}) {
  if (lm) { if (lm->stop) return; mu = lm->mu; }
  ap.ts = st.ts;
  const bool first = (blockIdx.x == 0 && threadIdx.x == 0);
  T beta = 0, scale = 0;
  if (k < 0) {
    if (first && st.ts) st.ts[0] = wall_clock64();
    scale = (T)(1.0 / (double)(T)sqrt((double)(T)slot_sum(st.slots(0, RR), 0)));
    const double zdz = slot_sum(st.slots(0, ZDZ), 0);
    if (first) st.pdp[0] = (double)scale * (double)scale * zdz;
  } else {
    const double rz0 = st.rz0[k];
    const bool was_done = st.done[k] != 0; // an earlier direction launch ended the loop (and applied the step)
    bool leave = was_done;
    // every dot product this launch may need, in one go (slot_sums): records k and k + 1
    const double *const sl[6] = {st.slots(k, RZP), st.slots(k, RR), st.slots(k + 1, RZP), st.slots(k + 1, RR), st.slots(k + 1, PDZ), st.slots(k + 1, ZDZ)};
    double sv[6];
    slot_sums<6>(sl, sv);
    PcgIter it{};
    if (!leave) { it.rzp = sv[0]; it.rscale = 1.0 / sqrt(sv[1]); it.rz = it.rzp * it.rscale; leave = (it.rzp == 0.0); }
    if (leave) {
      if (first) { st.done[k + 1] = 1; st.rz0[k + 1] = rz0; if (st.left) *st.left = 1; if (!ap.cams && st.ts && !was_done) st.ts[1] = wall_clock64(); st.hflag[k] = 2; __threadfence_system(); }
      if (ap.cams && !was_done) apply_on_exit<T>(ap, n, pose_dim, x, scales, mu, nullptr); // r.z == 0: x is final
      return;
    }
    PcgIter nx;
    nx.rzp = sv[2]; nx.rscale = 1.0 / sqrt(sv[3]); nx.rz = nx.rzp * nx.rscale;
    const double pdz = sv[4], zdz = sv[5];
    const T rz = (T)it.rzp * (T)(double)(T)it.rscale, rz_new = (T)nx.rzp * (T)(double)(T)nx.rscale;
    const bool reject = (fabs((double)rz_new) > rejection_ratio * rz0) || (rz_new != rz_new);
    const bool done_next = reject || fabs((double)rz_new) < tol;
    beta = rz_new / rz;
    scale = (T)(double)(T)nx.rscale;
    if (first) {
      st.rz0[k + 1] = reject ? rz0 : fmin(rz0, fabs((double)rz_new));
      st.done[k + 1] = (done_next || ap.at_cap) ? 1 : 0;
      st.pdp[k + 1] = (double)beta * (double)beta * st.pdp[k] + 2.0 * (double)beta * (double)scale * pdz + (double)scale * (double)scale * zdz;
      st.iters[0] = k + 1;
      *st.hiters = k + 1;
      if ((done_next || ap.at_cap) && st.left) *st.left = 1;
      if ((done_next || ap.at_cap) && !ap.cams && st.ts) st.ts[1] = wall_clock64(); // (with ApplyOnExit the stamp is apply_on_exit's)
      st.hflag[k] = done_next ? 2 : 1;
      __threadfence_system();
    }
    if (ap.cams && (done_next || ap.at_cap)) { // the loop ends here: the step is x (the backup on a rejected iteration)
      apply_on_exit<T>(ap, n, pose_dim, reject ? xb : x, scales, mu, reject ? x : nullptr);
      return;
    }
    if (reject) {
      for (unsigned t = blockIdx.x * TPB + threadIdx.x; t < n; t += gridDim.x * TPB) x[t] = xb[t];
      return;
    }
    // the loop has ended: nobody reads the next direction (Ladybug-1723 bench line 5 090 -> 5 250 LM it/s: with the
    // reference's tolerance most solves end after 1-2 iterations, so this was every second direction launch)
    if (done_next) return;
  }
  for (unsigned t = blockIdx.x * TPB + threadIdx.x; t < n; t += gridDim.x * TPB) {
    const T pn = (k < 0) ? scale * zt[t] : beta * p[t] + scale * zt[t];
    p[t] = pn;
    if (xp && t >= pose_dim) { const unsigned q = t - pose_dim; xp[8 * (size_t)(q / 3u) + 3 + q % 3u] = scales[t] * pn; }
    else ps[t] = scales[t] * pn;
  }
}

// Multi-GPU helper: camera rows of the operator summed over this rank's segments (the
// all-reduce over ranks follows on the host side of the stream).
template <typename T>
__global__ void k_cam_rows(int Nc, const int *__restrict__ cam_seg_ptr, const T *__restrict__ op_partial,
                           T *__restrict__ raw_c, const int *__restrict__ done, int k) {
  if (done && done[k]) return;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 9u * (unsigned)Nc) return;
  const unsigned c = t / 9u, i = t % 9u;
  T raw = 0;
  int sg = cam_seg_ptr[c];
  const int sg1 = cam_seg_ptr[c + 1], last = sg1 - 1;
  for (; sg < sg1; sg += 4) { // four loads in flight (the last batch clamped), added in segment order
    const int n = sg1 - sg;
    const T q0 = op_partial[9 * (size_t)sg + i], q1 = op_partial[9 * (size_t)(sg + 1 < last ? sg + 1 : last) + i],
            q2 = op_partial[9 * (size_t)(sg + 2 < last ? sg + 2 : last) + i], q3 = op_partial[9 * (size_t)(sg + 3 < last ? sg + 3 : last) + i];
    raw += q0;
    if (n > 1) raw += q1;
    if (n > 2) raw += q2;
    if (n > 3) raw += q3;
  }
  raw_c[t] = raw;
}

} // namespace gr
