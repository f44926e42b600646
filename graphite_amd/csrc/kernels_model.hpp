// Model-free kernels of the USER-TRAITS engine (include/graphite_mi355x_model.h) — gfx950.
//
// The user-side lineariser (include/graphite/engine_model.hpp) leaves the WEIGHTED Jacobian Jt = L J (W = rho' P = L^T L)
// of every observation in GR_MODEL_JSTREAMS structure-of-arrays streams, camera-major observation order: stream 2 col + row
// of the 2 x 9 pose block, then the 2 x 3 landmark block.  With it the matrix-free operator of PCGSolver
// (solver/pcg.hpp:143-163, ops/product.hpp:51-103,228-292: v1 = J p, then J^T P v1 with the stored blocks) needs neither the
// user's functions nor the loss nor the precision matrices:  u = Jt ps,  den = |u|^2,  rows = Jt^T u.
#pragma once
#include "kernels_mf.hpp"

namespace gr {

// One observation per lane, the 2 DC + 6 streams read as whole lines (64 consecutive scalars per wave instruction: the
// kernel's traffic is the stored Jacobian, once), the landmark's direction gathered, the pose's direction fetched through a
// wave-uniform index (scalar loads), pose rows by wave transpose-reduction into (wave, camera) segments, landmark rows at the
// observation's point-major slot — the layout k_pcg_update consumes.  DC: pose streams read (columns >= DC are zero padding).
template <typename T, typename SJ, int DC>
__global__ void __launch_bounds__(TPB, OP_WAVES)
k_pcg_operator_stored(int No, int Nc, int ntiles, const int *__restrict__ cam_cm, const int *__restrict__ pt_cm,
                      const int *__restrict__ pos_cm, const int *__restrict__ blk_seg, const int *__restrict__ seg_slot,
                      const SJ *__restrict__ jst, long long stride, const T *__restrict__ ps, T *__restrict__ g3,
                      T *__restrict__ op_partial, PcgState st, int k, const LmDev *__restrict__ lm) {
  if (lm && lm->stop) return;
  if (st.done[k]) return;
  if (slot_sum(st.slots(k, RZP), 0) == 0.0) return; // rz == 0: the direction kernel of this iteration closes the loop
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  const size_t pose_dim = 9 * (size_t)Nc;
  int j0, jstride, niter, jlim;
  xcd_obs_range(ntiles, No, j0, jstride, niter, jlim);
  double den = 0;
  int j = j0 + threadIdx.x;
  bool valid = niter > 0 && j < jlim;
  int c_n = -1, l_n = 0, a_n = 0;
  if (valid) { c_n = cam_cm[j]; l_n = pt_cm[j]; a_n = pos_cm ? pos_cm[j] : j; }
  for (int it = 0; it < niter; ++it) {
    const int c = c_n, l = l_n;
    const size_t a = (size_t)a_n;
    const int jn = j + jstride;
    const bool validn = (it + 1 < niter) && jn < jlim;
    if (validn) { c_n = cam_cm[jn]; l_n = pt_cm[jn]; a_n = pos_cm ? pos_cm[jn] : jn; }
    T Jc[2 * DC], Jp[6], pl0 = 0, pl1 = 0, pl2 = 0;
#pragma unroll
    for (int i = 0; i < 2 * DC; ++i) Jc[i] = T(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) Jp[i] = T(0);
    if (valid) {
      const SJ *q = jst + j;
#pragma unroll
      for (int i = 0; i < 2 * DC; ++i) Jc[i] = (T)q[(long long)i * stride];
#pragma unroll
      for (int i = 0; i < 6; ++i) Jp[i] = (T)q[(long long)(18 + i) * stride];
      const T *pl = ps + pose_dim + 3 * (size_t)l;
      pl0 = pl[0]; pl1 = pl[1]; pl2 = pl[2];
    }
    const T up0 = Jp[0] * pl0 + Jp[2] * pl1 + Jp[4] * pl2;
    const T up1 = Jp[1] * pl0 + Jp[3] * pl1 + Jp[5] * pl2;
    unsigned long long remaining = __ballot(valid);
    int segf = blk_seg[__builtin_amdgcn_readfirstlane(j >> 6)];
    while (remaining) {
      const int leader = __builtin_ctzll(remaining);
      const int cl = __builtin_amdgcn_readlane(c, leader);
      const bool mine = valid && c == cl;
      const int segl = seg_slot[segf++];
      T u0 = up0, u1 = up1;
#pragma unroll
      for (int i = 0; i < DC; ++i) { const T pc = ps[9 * (size_t)cl + i]; u0 += Jc[2 * i] * pc; u1 += Jc[2 * i + 1] * pc; }
      if (mine) den += (double)(u0 * u0 + u1 * u1);
      T m[16];
#pragma unroll
      for (int i = 0; i < DC; ++i) m[i] = mine ? Jc[2 * i] * u0 + Jc[2 * i + 1] * u1 : T(0);
#pragma unroll
      for (int i = DC; i < 16; ++i) m[i] = T(0);
      if (mine) {
        T *g = g3 + 3 * a;
        g[0] = Jp[0] * u0 + Jp[1] * u1;
        g[1] = Jp[2] * u0 + Jp[3] * u1;
        g[2] = Jp[4] * u0 + Jp[5] * u1;
      }
      const T tot = wave_transpose_sum<T, 16>(m, lane);
      if ((lane & 3) == 0 && (lane >> 2) < 9) op_partial[9 * (size_t)segl + (lane >> 2)] = tot;
      remaining &= ~__ballot(mine);
    }
    valid = validn;
    j = jn;
  }
  den = block_sum_256(den, red);
  if (threadIdx.x == 0) slot_add(st.slots(k, DEN), 0, den);
}

// Second half of Graph::chi2 + compute_rho for user-traits problems (the first is the user-side chi2 launcher, which leaves
// one partial per workgroup): chi2 = fixed-order sum of the partials, rho denominator = sum dx (mu dx + s b)
// (levenberg_marquardt.hpp:34-41); the last workgroup publishes both (dscal, pinned hres, then hres_seq = seq).
template <typename T>
__global__ void __launch_bounds__(TPB)
k_chi2_finish(unsigned n, unsigned pose_dim, int cam_weight, const double *__restrict__ chi2_partial, int n_partials,
              const T *__restrict__ dx, const T *__restrict__ bu, const T *__restrict__ scales, double mu,
              double *__restrict__ partial, unsigned *__restrict__ ticket, double *__restrict__ dscal, volatile double *hres,
              volatile int *hres_seq, int seq) {
  __shared__ double red[4];
  double chi2 = 0, rho = 0;
  if (blockIdx.x == 0) for (int i = threadIdx.x; i < n_partials; i += TPB) chi2 += chi2_partial[i];
  if (dx) {
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
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

} // namespace gr
