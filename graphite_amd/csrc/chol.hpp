// Dense, tile-sparse Cholesky solve of the reduced camera system S x = b_S on MFMA.
//
// Role in the reference: the direct inner solvers EigenSchurLDLTSolver::solve
// (solver/eigen_schur.hpp:71-108: CPU SimplicialLDLT on the upper CSC of S) and
// cudssSchurSolver::solve (solver/cudss_schur.hpp:190-234).  Here S is expanded into a dense
// row-major lower triangle padded to 128x128 tiles and factorised right-looking, S = L L^T:
//
//   per panel k:  k_chol_potrf   L_kk = chol(A_kk), Linv_k = L_kk^-1           (one workgroup, LDS)
//                 k_chol_gemm<0> L_ik = A_ik Linv_k^T          for tiles i in rows(k)   (MFMA)
//                 k_chol_gemm<1> A_ij -= L_ik L_jk^T           for i >= j in rows(k)    (MFMA)
//
// rows(k) comes from a tile-level symbolic factorisation on the host, so a block-banded S
// (Ladybug-like camera graphs) only touches the tiles inside its filled band, while a
// Venice-like S degenerates to the fully dense factorisation.  The GEMM is the one true
// contraction on the whole path and the only place MFMA is used: v_mfma_f64_16x16x4_f64 /
// v_mfma_f32_16x16x4_f32, 128x128 tile per 256-thread workgroup, 64x64 per wave.
#pragma once
#include "common.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>

namespace gr {

constexpr int CH_NB = 128;         // panel width == tile edge
constexpr int CH_KC = 16;          // K chunk staged through LDS per pipeline step
constexpr int CH_LDP = CH_NB + 4;  // LDS pitch of the [k][row] operand images
constexpr int CH_PT = 256;         // threads of the panel-factor workgroup
constexpr int CH_LP = CH_NB + 1;   // LDS pitch of the 128x128 diagonal block

template <typename T> struct MfmaTile;
template <> struct MfmaTile<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  // v_mfma_f64_16x16x4_f64: D[row][lane & 15], row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct MfmaTile<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  // v_mfma_f32_16x16x4_f32: D[row][lane & 15], row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

template <typename T> __device__ __forceinline__ void load8(const T *p, T (&v)[8]);
template <> __device__ __forceinline__ void load8<double>(const double *p, double (&v)[8]) {
  const double2 *q = reinterpret_cast<const double2 *>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) { const double2 t = q[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
}
template <> __device__ __forceinline__ void load8<float>(const float *p, float (&v)[8]) {
  const float4 *q = reinterpret_cast<const float4 *>(p);
#pragma unroll
  for (int i = 0; i < 2; ++i) { const float4 t = q[i]; v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
}

// value of lane `src` (0..15) of the caller's 16-lane row, in every lane of that row: DPP row_share, a plain VALU move —
// v_readlane goes through an SGPR and pays the VALU->SGPR->VALU wait states 480 times per 16 x 16 diagonal block.
// `src` is a constant after unrolling (the switch folds away).
__device__ __forceinline__ int row_share(int v, int src) {
  switch (src & 15) {
#define GR_RS(N) case N: return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xf, 0xf, false);
    GR_RS(0) GR_RS(1) GR_RS(2) GR_RS(3) GR_RS(4) GR_RS(5) GR_RS(6) GR_RS(7) GR_RS(8) GR_RS(9) GR_RS(10) GR_RS(11) GR_RS(12) GR_RS(13) GR_RS(14)
#undef GR_RS
    default: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 15, 0xf, 0xf, false);
  }
}
// 1/sqrt(d) for the pivots: hardware estimate + Newton steps y <- y (1.5 - 0.5 d y^2) (two in fp64, one in fp32) instead
// of the library routine (a correctly rounded sqrt followed by a division) on the serial critical path of the
// 16 x 16 diagonal step; relative error a few ulp, absorbed by the L L^T product like any other rounding
__device__ __forceinline__ double pivot_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}
__device__ __forceinline__ float pivot_rsqrt(float d) {
  float y = __builtin_amdgcn_rsqf(d);
  y = y * (1.5f - 0.5f * d * y * y);
  return y;
}
template <typename T> __device__ __forceinline__ T lane_bcast(T v, int src);
template <> __device__ __forceinline__ float lane_bcast<float>(float v, int src) {
  return __builtin_bit_cast(float, row_share(__builtin_bit_cast(int, v), src));
}
// fp64: ONE 64-bit DPP move (v_mov_b64_dpp row_newbcast, gfx90a+) instead of two 32-bit ones — 240 broadcasts per 16 x 16 diagonal step
__device__ __forceinline__ long long row_share64(long long v, int src) {
  switch (src & 15) {
#define GR_RS(N) case N: return __builtin_amdgcn_update_dpp(v, v, 0x150 + N, 0xf, 0xf, false);
    GR_RS(0) GR_RS(1) GR_RS(2) GR_RS(3) GR_RS(4) GR_RS(5) GR_RS(6) GR_RS(7) GR_RS(8) GR_RS(9) GR_RS(10) GR_RS(11) GR_RS(12) GR_RS(13) GR_RS(14)
#undef GR_RS
    default: return __builtin_amdgcn_update_dpp(v, v, 0x150 + 15, 0xf, 0xf, false);
  }
}
template <> __device__ __forceinline__ double lane_bcast<double>(double v, int src) {
  return __builtin_bit_cast(double, row_share64(__builtin_bit_cast(long long, v), src));
}

// Diagonal block: L = chol(A_kk) and X = L^-1, entirely in LDS, blocked by 16 so that only the eight
// 16x16 diagonal sub-blocks are sequential (one wave, rows in registers, readlane broadcasts) and
// everything else (sub-panel solve, trailing update, blocked inverse) is 16x16x4 MFMA on LDS data.
// X^T lives in the unused strictly-upper part of the same LDS image; the diagonal of X in xd[].
// Writes the lower triangle of X into `Linv` (a 128x128 row-major matrix whose strictly upper part stays zero from its
// allocation) for the panel GEMM and the solves; L_kk itself is not stored: nothing reads it again.  A pivot that is not > 0 raises *fail and is
// replaced by 1 so that the rest stays finite.
// `L` is the LDS image [128][129] (+128 for the diagonal of X); with `load` it is filled from the global
// tile first, otherwise the caller has already put the lower triangle (zeros above) there.
// NT: threads of the calling workgroup (256, or 512 in the nested-dissection update launch: the sub-panel solves, trailing updates
// and inverse rows spread over twice the waves; the serial diagonal steps stay one wave's)
template <typename T, int NT = CH_PT>
__device__ __forceinline__ void chol_potrf_block(T *__restrict__ L, T *__restrict__ Ag, int ld, T *__restrict__ Linv, int *__restrict__ fail, bool load, int skip) {
  using M = MfmaTile<T>;
  typedef typename M::acc_t acc_t;
  T *xd = L + CH_NB * CH_LP; // diag of X
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, cl = lane & 15, g = lane >> 4;
  constexpr int NW = NT / 64, NB16 = CH_NB / 16;
  if (load) { // eight contiguous scalars per request, every request of a thread in flight before the first LDS store (scalar loads: 14 us of the 57)
    constexpr int NCH = CH_NB * CH_NB / 8 / NT, BATCH = NCH < 4 ? NCH : 4; // (four requests of a thread in flight: 64 fp64 registers)
#pragma unroll 1
    for (int q0 = 0; q0 < NCH; q0 += BATCH) {
      T v[BATCH][8];
#pragma unroll
      for (int q = 0; q < BATCH; ++q) {
        const int e = t + (q0 + q) * NT, r = e >> 4, c0 = (e & 15) * 8;
        if (c0 <= r) load8<T>(Ag + (size_t)r * ld + c0, v[q]);
      }
#pragma unroll
      for (int q = 0; q < BATCH; ++q) {
        const int e = t + (q0 + q) * NT, r = e >> 4, c0 = (e & 15) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) L[r * CH_LP + c0 + i] = (c0 <= r && c0 + i <= r) ? v[q][i] : T(0);
      }
    }
  }
  __syncthreads();
  // blocked inverse, block (i, j), j < i: X_ij = -X_ii sum_{k=j}^{i-1} L_ik X_kj.  Row i needs the diagonal step i (X_ii), the
  // sub-panel solves of the steps before it (L_ik) and the rows of X above it: it is computed by waves 1 .. 3 WHILE wave 0 runs
  // the serial diagonal step i + 1 (they used to idle through the eight 3.25 us steps, and the inverse then cost 12 us of its own)
  auto inverse_block = [&](int i, int j) {
    acc_t tacc = {T(0), T(0), T(0), T(0)};
    for (int k = j; k < i; ++k) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int kr = 4 * kk + g;
        const T av = L[(16 * i + cl) * CH_LP + 16 * k + kr];
        T bv;
        if (k == j) bv = kr > cl ? L[(16 * j + cl) * CH_LP + 16 * j + kr] : (kr == cl ? xd[16 * j + cl] : T(0));
        else bv = L[(16 * j + cl) * CH_LP + 16 * k + kr];
        tacc = M::mma(av, bv, tacc);
      }
    }
    acc_t xacc = {T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int kr = M::row(lane, kk); // the row of T this lane holds in register kk is the k it feeds
      const T av = cl > kr ? L[(16 * i + kr) * CH_LP + 16 * i + cl] : (cl == kr ? xd[16 * i + cl] : T(0));
      xacc = M::mma(av, tacc[kk], xacc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) L[(16 * j + cl) * CH_LP + 16 * i + M::row(lane, r)] = -xacc[r];
  };
  // 16x16 Cholesky + inverse of diagonal block s by ONE wave: lane r (mod 16) owns row r of the block, then column r of X
  auto diag_step = [&](int s) {
    const int o = 16 * s;
    T a[16], x[16], rsv[16];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = k <= cl ? L[(o + cl) * CH_LP + o + k] : T(0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      T d = lane_bcast<T>(a[j], j);
      if (!(d > T(0))) { bad = true; d = T(1); }
      rsv[j] = pivot_rsqrt(d);
      a[j] = cl == j ? d * rsv[j] : a[j] * rsv[j];
#pragma unroll
      for (int k = j + 1; k < 16; ++k) a[k] -= a[j] * lane_bcast<T>(a[j], k);
    }
    // X = L^-1, lane cl holds column cl: x[m] = (delta_m,cl - sum_{k<m} L[m][k] x[k]) / L[m][m], by COLUMNS of L — once x[k] is final its
    // multiples leave all later entries at once (15 - k independent multiply-adds; the row form summed each row in a serial chain).  Same
    // products, same order of additions per entry: same bits.  The broadcasts depend on a[] only; left alone all 120 are issued up front
    // (240 fp64 registers: fine in a 256-thread workgroup, 170 spilled to scratch in a 512-thread one) — there they are tied to the
    // previous column's result, one column of look-ahead.
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = (m == cl) ? T(1) : T(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      x[k] = x[k] * rsv[k];
      T ak = a[k];
      if (NT > 256 && k > 0) asm volatile("" : "+v"(ak) : "v"(x[k - 1]));
#pragma unroll
      for (int m = k + 1; m < 16; ++m) x[m] -= lane_bcast<T>(ak, m) * x[k];
    }
    if (lane < 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k <= cl) L[(o + cl) * CH_LP + o + k] = a[k];
        if (k > cl) L[(o + cl) * CH_LP + o + k] = x[k];
      }
      xd[o + cl] = x[cl];
    }
    if (bad && lane == 0) *fail = 1;
  };
  // trailing update of ONE 16x16 tile (ti >= tk > s) by the sub-panel column of step s: A_ik -= L_is L_ks^T
  auto trailing_tile = [&](int s, int q) {
    const int o = 16 * s;
    int bi = 0, rem = q;
    while (rem > bi) { rem -= bi + 1; ++bi; } // q -> (bi, bk) in the lower triangle; q = 0 is the next diagonal block
    const int ti = s + 1 + bi, tk = s + 1 + rem;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = L[(16 * ti + M::row(lane, r)) * CH_LP + 16 * tk + cl];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 4 * kk + g;
      acc = M::mma(-L[(16 * ti + cl) * CH_LP + o + k], L[(16 * tk + cl) * CH_LP + o + k], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) L[(16 * ti + M::row(lane, r)) * CH_LP + 16 * tk + cl] = acc[r];
  };
  // Write-back of rows [rb, rb + 16) of X = L^-1, lower triangle only, by the waves wv0 .. NW - 1 (8 256 of the 32 768 scalars the
  // block used to store; L_kk itself is read by nobody once the panel is solved with X, and the strictly upper part of X is zero:
  // the buffer is zeroed when it is allocated and never written above the diagonal).  Row block s - 1 goes out in window s, under
  // wave 0's diagonal step: the write-back used to be a 6.6 us tail of the factorisation (tools/potrf_bench.hip).
  auto store_rows = [&](int rb, int wv0) {
    if (wave < wv0) return;
    for (int r = rb + (wave - wv0); r < rb + 16; r += NW - wv0) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int cc = 64 * half + lane;
        if (cc <= r) Linv[r * CH_NB + cc] = cc < r ? L[cc * CH_LP + r] : xd[r];
      }
    }
  };
  // sub-panel solve of ONE 16-row block below diagonal block s: L_bi,s = A_bi,s X_ss^T
  auto subpanel_block = [&](int s, int bi) {
    const int o = 16 * s;
    acc_t acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 4 * kk + g;
      const T av = L[(16 * bi + cl) * CH_LP + o + k];
      const T bv = cl > k ? L[(o + k) * CH_LP + o + cl] : (cl == k ? xd[o + cl] : T(0));
      acc = M::mma(av, bv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) L[(16 * bi + M::row(lane, r)) * CH_LP + o + cl] = acc[r];
  };
  // WAVE 0 OWNS THE CRITICAL CHAIN (round 5).  The next diagonal step needs, of step s, only the sub-panel block right below the
  // diagonal and the trailing update of the next diagonal block — both 16 x 16 MFMA products on data only wave 0 touches in this window.
  // So wave 0 runs  sub-panel (s + 1, s) -> trailing (s + 1, s + 1) -> diagonal step s + 1  without meeting a workgroup barrier, while
  // waves 1 .. solve the rest of the sub-panel, meet each other (and wave 0's sub-panel block) at an LDS counter, update the rest of the
  // trailing matrix, write back inverse rows and take the blocks of row s of the inverse; ONE workgroup barrier per window instead of two.
  // The window used to cost max over the waves of (sub-panel share) + barrier + max(wave 0: update + diagonal step, others: trailing share +
  // inverse) = 4.3 us; wave 0's chain alone is 2.6 us.  Every tile is still computed once, by the same instruction sequence: same bits.
  __shared__ int s_arrive[2], s_deal[2]; // window s uses element s & 1; the other one is cleared for window s + 1 meanwhile
  if (t < 2) { s_arrive[t] = 0; s_deal[t] = 0; }
  if (wave == 0 && !(skip & 1)) diag_step(0);
  __syncthreads();
  for (int s = 0; s < NB16; ++s) {
    const int par = s & 1;
    if (t == 64) { s_arrive[par ^ 1] = 0; s_deal[par ^ 1] = 0; } // next window's counters: their last use ended before the barrier that opened this window
    const int m = NB16 - 1 - s, ntile = m * (m + 1) / 2;
    if (wave == 0) {
      if (s + 1 < NB16) {
        if (!(skip & 2)) subpanel_block(s, s + 1);
        wave_lds_fence();
        if (lane == 0) atomicAdd(&s_arrive[par], 1); // (LDS runs a wave's instructions in order: the block is written when the count is)
        if (!(skip & 2)) trailing_tile(s, 0);
        wave_lds_fence();
        if (!(skip & 1)) diag_step(s + 1); // only this wave touches block (s + 1, s + 1) until the barrier below
      }
    } else {
      for (int bi = s + 2 + (wave - 1); bi < NB16 && !(skip & 2); bi += NW - 1) subpanel_block(s, bi);
      if (s + 1 < NB16) {
        wave_lds_fence();
        if (lane == 0) atomicAdd(&s_arrive[par], 1);
        while (__hip_atomic_load(&s_arrive[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NW) __builtin_amdgcn_s_sleep(1);
        wave_lds_fence();
      }
      for (int q = wave; q < ntile && !(skip & 2); q += NW - 1) trailing_tile(s, q);
      if (s >= 1 && !(skip & 8)) store_rows(16 * (s - 1), 1); // complete since the barrier that ended window s - 1
    }
    // row s of the inverse: X_ss, the sub-panel solves of the steps before s and the rows of X above it are complete.  Its s blocks
    // are dealt from a counter to whichever wave is free — in the late windows the row (O(s^2) MFMA steps) outweighs the 16 x 16
    // diagonal step, and wave 0 joins once that is done
    if (!(skip & 4))
      for (;;) {
        int j = 0;
        if (lane == 0) j = atomicAdd(&s_deal[par], 1);
        j = __builtin_amdgcn_readfirstlane(j);
        if (j >= s) break;
        inverse_block(s, j);
      }
    __syncthreads();
  }
  if (!(skip & 8)) store_rows(16 * (NB16 - 1), 0);
  (void)Ag; (void)ld;
}
template <typename T, int NT = CH_PT>
__global__ __launch_bounds__(NT) void k_chol_potrf(T *__restrict__ A, int ld, int k0, T *__restrict__ Linv, int *__restrict__ fail, int skip = 0) {
  extern __shared__ __align__(16) unsigned char ch_smem[];
  chol_potrf_block<T, NT>(reinterpret_cast<T *>(ch_smem), A + (size_t)k0 * ld + k0, ld, Linv, fail, true, skip);
}

constexpr size_t chol_potrf_lds(size_t w) { return (CH_NB * CH_LP + CH_NB) * w; }

// C = P Q^T  (MODE 0, C overwrites P: the panel solve with Q = Linv_k)
// C -= P Q^T (MODE 1, the trailing update; P, Q = panel tiles of rows ti, tj)
// One 128x128 tile of C per workgroup; K = 16 nch (one or two panels) in chunks of 16, double-buffered
// through LDS.
// For MODE 1 the accumulators start as C and P is negated on its way into LDS, so the epilogue
// is a plain store and the C read overlaps the first operand fetch.
// FUSE (MODE 1 only): workgroup 0's tile is the NEXT diagonal tile; once updated it is factorised on the
// spot (chol_potrf_block, from its accumulators through LDS) while the other workgroups keep updating,
// which takes one of the two sequential panel factorisations per 256 columns off the critical path.
template <typename T, int MODE, int PIN = 0, int KC = CH_KC, bool FUSE = false>
__global__ __launch_bounds__(256) void k_chol_gemm(T *__restrict__ A, int ld, const int *__restrict__ tiles, int k0, const T *__restrict__ Linv, int nch,
                                                   T *__restrict__ Linv_next = nullptr, int *__restrict__ fail = nullptr) {
  extern __shared__ __align__(16) unsigned char ch_smem[];
  T *sm = reinterpret_cast<T *>(ch_smem);
  using M = MfmaTile<T>;
  typedef typename M::acc_t acc_t;
  const int ti = MODE == 0 ? tiles[blockIdx.x] : tiles[2 * blockIdx.x];
  const int tj = MODE == 0 ? 0 : tiles[2 * blockIdx.x + 1];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
  const T *Pg = A + (size_t)ti * CH_NB * ld + k0;
  const T *Qg = MODE == 0 ? Linv : A + (size_t)tj * CH_NB * ld + k0;
  const int ldq = MODE == 0 ? CH_NB : ld;
  T *Cg = MODE == 0 ? A + (size_t)ti * CH_NB * ld + k0 : A + (size_t)ti * CH_NB * ld + (size_t)tj * CH_NB;

  acc_t acc[4][4];
  const int ccol = lane & 15;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (MODE == 1) acc[mi][ni][r] = Cg[(size_t)(wr * 64 + mi * 16 + M::row(lane, r)) * ld + wc * 64 + ni * 16 + ccol];
        else acc[mi][ni][r] = T(0);
      }

  // loader role: KC / 8 threads per tile row, 8 consecutive k each; NPASS row groups per chunk
  constexpr int TPR = KC / 8, RPP = 256 / TPR, NPASS = CH_NB / RPP;
  const int lr = t / TPR, lk = (t % TPR) * 8;
  T pp[NPASS][8], pq[NPASS][8];
#pragma unroll
  for (int u = 0; u < NPASS; ++u) {
    load8<T>(Pg + (size_t)(lr + u * RPP) * ld + lk, pp[u]);
    load8<T>(Qg + (size_t)(lr + u * RPP) * ldq + lk, pq[u]);
  }
  constexpr int BUF = 2 * KC * CH_LDP;
  {
    T *Ps = sm, *Qs = sm + KC * CH_LDP;
#pragma unroll
    for (int u = 0; u < NPASS; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) { Ps[(lk + e) * CH_LDP + lr + u * RPP] = MODE == 1 ? -pp[u][e] : pp[u][e]; Qs[(lk + e) * CH_LDP + lr + u * RPP] = pq[u][e]; }
  }
  __syncthreads();
  if (MODE == 1) {
    // pin the C loads before the loop: otherwise their s_waitcnt lands inside the loop body and, in
    // steady state, makes every iteration wait for its own operand prefetch half-way through
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        if (PIN == 0) asm volatile("" : "+a"(acc[mi][ni]));
        else asm volatile("" : "+v"(acc[mi][ni]));
      }
  }
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) {
#pragma unroll
      for (int u = 0; u < NPASS; ++u) {
        load8<T>(Pg + (size_t)(lr + u * RPP) * ld + (c + 1) * KC + lk, pp[u]);
        load8<T>(Qg + (size_t)(lr + u * RPP) * ldq + (c + 1) * KC + lk, pq[u]);
      }
    }
    const T *Ps = sm + (c & 1) * BUF, *Qs = Ps + KC * CH_LDP;
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const int krow = (kk * 4 + (lane >> 4)) * CH_LDP + ccol;
      T a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = Ps[krow + wr * 64 + i * 16]; b[i] = Qs[krow + wc * 64 + i * 16]; }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = M::mma(a[mi], b[ni], acc[mi][ni]);
    }
    if (c + 1 < nch) {
      T *Pn = sm + ((c + 1) & 1) * BUF, *Qn = Pn + KC * CH_LDP;
#pragma unroll
      for (int u = 0; u < NPASS; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) { Pn[(lk + e) * CH_LDP + lr + u * RPP] = MODE == 1 ? -pp[u][e] : pp[u][e]; Qn[(lk + e) * CH_LDP + lr + u * RPP] = pq[u][e]; }
    }
    __syncthreads();
  }
  // keep the 64 store addresses from being hoisted above the K loop (they would cost 128 VGPRs and
  // with them the second workgroup per CU that overlaps this epilogue with MFMA work)
  if (FUSE && blockIdx.x == 0) {
    // (ti, tj) == the next diagonal tile: lower triangle of the updated tile -> LDS image, then factorise it
    T *L = sm; // the operand buffers are free: every wave passed the barrier that ends the K loop
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wr * 64 + mi * 16 + M::row(lane, r), col = wc * 64 + ni * 16 + ccol;
          L[row * CH_LP + col] = col <= row ? acc[mi][ni][r] : T(0);
        }
    chol_potrf_block<T>(L, Cg, ld, Linv_next, fail, false, 0);
    return;
  }
  int ld2 = ld;
  asm volatile("" : "+v"(ld2));
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cg[(size_t)(wr * 64 + mi * 16 + M::row(lane, r)) * ld2 + wc * 64 + ni * 16 + ccol] = acc[mi][ni][r];
}
constexpr size_t chol_gemm_lds(size_t w, int kc = CH_KC) { return (size_t)2 * 2 * kc * CH_LDP * w; }


// zero the structurally non-zero lower tiles; padded diagonal entries (>= n) become 1
template <typename T>
__global__ __launch_bounds__(256) void k_chol_clear(T *__restrict__ A, int ld, int n, const int *__restrict__ tiles) {
  const int ti = tiles[2 * blockIdx.x], tj = tiles[2 * blockIdx.x + 1];
  T *Cg = A + (size_t)ti * CH_NB * ld + (size_t)tj * CH_NB;
  for (int e = threadIdx.x; e < CH_NB * CH_NB; e += 256) {
    const int r = e >> 7, c = e & 127;
    const int R = ti * CH_NB + r, C = tj * CH_NB + c;
    Cg[(size_t)r * ld + c] = (R == C && R >= n) ? T(1) : T(0);
  }
}

// scatter the upper 9x9 blocks of S (column-major blocks, block (i <= j)) into the dense lower triangle
template <typename T>
__global__ __launch_bounds__(256) void k_chol_scatter(int64_t nnzb, const int *__restrict__ rowi, const int *__restrict__ coli, const T *__restrict__ S, T *__restrict__ A, int ld) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= 81 * nnzb) return;
  const int64_t q = e / 81;
  const int w = (int)(e - 81 * q), c = w / 9, r = w - 9 * c; // S_q(r, c) at global (9 i + r, 9 j + c)
  const int R = 9 * coli[q] + c, C = 9 * rowi[q] + r;          // mirrored into the lower triangle
  if (R >= C) A[(size_t)R * ld + C] = S[e];
}

template <typename T> __global__ void k_chol_rhs(int n, int npad, const T *__restrict__ b, T *__restrict__ vb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) vb[i] = i < n ? b[i] : T(0);
}

// 16 rows x 128 columns times a 128-vector held as (v0 = y[lane], v1 = y[64 + lane]): all 32 loads are
// issued before the cross-lane reduction; lane L returns the dot product of row (L >> 2)
template <typename T> __device__ __forceinline__ T rows16_dot(const T *__restrict__ base, size_t ld, T v0, T v1, int lane) {
  T v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = base[i * ld + lane] * v0 + base[i * ld + 64 + lane] * v1;
  return wave_transpose_sum<T, 16>(v, lane);
}

// forward step k: y_k = Linv_k b_k (every workgroup, into LDS); workgroup 0 stores y_k, workgroup
// w >= 1 applies b_i -= L_ik y_k for its tile row i = rows[w - 1]
template <typename T>
__global__ __launch_bounds__(256) void k_chol_fwd(const T *__restrict__ A, int ld, const T *__restrict__ Linv, int k0, const int *__restrict__ rows, T *__restrict__ b, T *__restrict__ y) {
  __shared__ T yk[CH_NB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const T b0 = b[k0 + lane], b1 = b[k0 + 64 + lane];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r0 = wave * 32 + h * 16;
    const T s = rows16_dot<T>(Linv + r0 * CH_NB, CH_NB, b0, b1, lane);
    if ((lane & 3) == 0) yk[r0 + (lane >> 2)] = s;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (t < CH_NB) y[k0 + t] = yk[t];
    return;
  }
  const int ti = rows[blockIdx.x - 1];
  const T y0 = yk[lane], y1 = yk[64 + lane];
  const T *Lg = A + (size_t)ti * CH_NB * ld + k0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r0 = wave * 32 + h * 16;
    const T s = rows16_dot<T>(Lg + (size_t)r0 * ld, (size_t)ld, y0, y1, lane);
    if ((lane & 3) == 0) b[ti * CH_NB + r0 + (lane >> 2)] -= s;
  }
}

// backward step k, part 1: partial[w][c] = sum_r L_ik[r][c] x_i[r] for tile row i = rows[w]
template <typename T>
__global__ __launch_bounds__(256) void k_chol_bwd_partial(const T *__restrict__ A, int ld, int k0, const int *__restrict__ rows, const T *__restrict__ x, T *__restrict__ partial) {
  __shared__ T half[CH_NB];
  const int t = threadIdx.x, c = t & 127, h = t >> 7;
  const int ti = rows[blockIdx.x];
  const T *Lg = A + (size_t)(ti * CH_NB + h * 64) * ld + k0 + c;
  const T *xg = x + ti * CH_NB + h * 64;
  T s = T(0);
#pragma unroll 8
  for (int r = 0; r < 64; ++r) s += Lg[(size_t)r * ld] * xg[r];
  if (h == 1) half[c] = s;
  __syncthreads();
  if (h == 0) partial[(size_t)blockIdx.x * CH_NB + c] = s + half[c];
}
// part 2: x_k = Linv_k^T (y_k - sum_w partial[w])
template <typename T>
__global__ __launch_bounds__(256) void k_chol_bwd_final(const T *__restrict__ Linv, int k0, int nrows, const T *__restrict__ partial, const T *__restrict__ y, T *__restrict__ x) {
  __shared__ T v[CH_NB];
  __shared__ T half[CH_NB];
  const int t = threadIdx.x, c = t & 127, h = t >> 7;
  if (h == 0) {
    T s = y[k0 + c];
#pragma unroll 8
    for (int w = 0; w < nrows; ++w) s -= partial[(size_t)w * CH_NB + c];
    v[c] = s;
  }
  __syncthreads();
  T s = T(0);
#pragma unroll 8
  for (int r = h * 64; r < h * 64 + 64; ++r) s += Linv[r * CH_NB + c] * v[r];
  if (h == 1) half[c] = s;
  __syncthreads();
  if (h == 0) x[k0 + c] = s + half[c];
}

struct CholProfSink {
  virtual void begin(const char *name, double bytes, double flops) = 0;
  virtual void end() = 0;
  virtual ~CholProfSink() = default;
};

// Host side: tile-level symbolic factorisation + the launch sequence.
//
// Panels are paired into super-panels (a, b = a + 1): both are factored and solved first (the update
// of tile column b by panel a is a narrow K = 128 pass), then the trailing matrix is updated ONCE with
// K = 256, which halves the read-modify-write traffic of the C tiles.  The first workgroup of that
// update owns the next diagonal tile and factorises it in place (k_chol_gemm<..., FUSE>) while the
// other ~7000 tiles are still being updated, so only one of the two panel factorisations per 256
// columns is on the critical path.  (A two-stream look-ahead of the whole panel phase was tried first:
// an update workgroup leaves neither the registers nor the LDS for a panel workgroup on the same CU,
// so nothing overlapped.)
template <typename T> struct DenseChol {
  hipStream_t stream = nullptr;
  int n = 0, npad = 0, nt = 0, nsp = 0;
  DevBuf<T> A, Linv, vb, vy, vx, partial;
  DevBuf<int> d_rows, d_pairs, d_nz, d_fail;
  std::vector<int> row_beg, row_end;                // per panel: rows below the diagonal tile (into d_rows)
  std::vector<int> col_off, next_off, rest_off;     // per super-panel: pair ranges (into d_pairs, in pairs)
  int nz_tiles = 0, max_rows = 0;
  int64_t total_pairs = 0, total_rows = 0, total_pairs128 = 0;
  int *h_fail = nullptr;
  CholProfSink *sink = nullptr;
  bool attrs_set = false;
  bool fuse_potrf = true; // gr_bal_tuning.chol_fuse (the next diagonal tile factorised inside the trailing update: 43.5 -> 39.4 ms on n = 15 507)
  // PIN = 1 keeps the C tile in VGPRs until the loop (one workgroup per CU in fp64): measured 7 % faster in
  // fp64 and on par in fp32 against the AGPR-pinned, two-workgroup variant (A/B in one run, n = 15507)
  int pin_variant = 1;    // gr_bal_tuning.chol_pin
  int potrf_skip = 0;     // timing ablation of chol_potrf_block's phases (tools/potrf_bench.hip sets it; never set by the product)

  DenseChol() = default;
  DenseChol(const DenseChol &) = delete;
  ~DenseChol() {
    if (h_fail) (void)hipHostFree(h_fail);
  }

  static size_t bytes_needed(int64_t n_) {
    const int64_t np = (n_ + CH_NB - 1) / CH_NB * CH_NB;
    return (size_t)np * np * sizeof(T);
  }
  // tile_nz: nt*nt, [i*nt + j] != 0 for structurally non-zero lower tiles (i >= j); empty = dense
  void set_structure(int n_, std::vector<char> tile_nz, hipStream_t s) {
    stream = s; n = n_; nt = (n + CH_NB - 1) / CH_NB; npad = nt * CH_NB; nsp = (nt + 1) / 2;
    if (tile_nz.empty()) tile_nz.assign((size_t)nt * nt, 1);
    auto nz = [&](int i, int j) -> char & { return tile_nz[(size_t)i * nt + j]; };
    for (int i = 0; i < nt; ++i) nz(i, i) = 1;
    std::vector<int> h_rows, h_pairs, h_nz;
    row_beg.assign(nt, 0); row_end.assign(nt, 0);
    col_off.assign(nsp + 1, 0); next_off.assign(nsp + 1, 0); rest_off.assign(nsp + 1, 0);
    std::vector<int> col_end(nsp, 0), next_end(nsp, 0);
    fused_next.assign(nsp, 0);
    max_rows = 0; total_pairs = 0; total_pairs128 = 0;
    std::vector<int> U;
    for (int p = 0; p < nsp; ++p) {
      const int a = 2 * p, b = a + 1, an = a + 2, bn = a + 3;
      U.clear();
      if (b < nt) {
        nz(b, a) = 1;
        for (int i = b + 1; i < nt; ++i) if (nz(i, a) || nz(i, b)) { U.push_back(i); nz(i, a) = nz(i, b) = 1; }
      }
      // rows of panel a: [b] + U; rows of panel b: U
      row_beg[a] = row_end[a] = (int)h_rows.size();
      if (b < nt) {
        h_rows.push_back(b);
        row_beg[b] = (int)h_rows.size();
        for (int i : U) h_rows.push_back(i);
        row_end[a] = row_end[b] = (int)h_rows.size();
      }
      max_rows = std::max(max_rows, (int)U.size() + 1);
      // pairs: column b from panel a (K = 128), then the trailing tiles (K = 256): next two columns first
      col_off[p] = (int)(h_pairs.size() / 2);
      if (b < nt) {
        h_pairs.push_back(b); h_pairs.push_back(b);
        for (int i : U) { h_pairs.push_back(i); h_pairs.push_back(b); }
      }
      col_end[p] = (int)(h_pairs.size() / 2);
      total_pairs128 += col_end[p] - col_off[p];
      next_off[p] = col_end[p];
      // the tile the next panel factorisation needs, (an, an), leads the list: workgroup 0 of the fused update
      fused_next[p] = !U.empty() && U[0] == an;
      if (fused_next[p]) { h_pairs.push_back(an); h_pairs.push_back(an); }
      for (size_t x = 0; x < U.size(); ++x)
        for (size_t y = 0; y <= x; ++y) {
          nz(U[x], U[y]) = 1; // fill-in
          if ((U[y] == an || U[y] == bn) && !(fused_next[p] && U[x] == an && U[y] == an)) { h_pairs.push_back(U[x]); h_pairs.push_back(U[y]); }
        }
      next_end[p] = (int)(h_pairs.size() / 2);
      rest_off[p] = next_end[p];
      for (size_t x = 0; x < U.size(); ++x)
        for (size_t y = 0; y <= x; ++y)
          if (U[y] > bn) { h_pairs.push_back(U[x]); h_pairs.push_back(U[y]); }
      total_pairs += (int64_t)(h_pairs.size() / 2) - next_off[p];
    }
    col_off[nsp] = next_off[nsp] = rest_off[nsp] = (int)(h_pairs.size() / 2);
    col_end_ = col_end; next_end_ = next_end;
    for (int i = 0; i < nt; ++i)
      for (int j = 0; j <= i; ++j) if (nz(i, j)) { h_nz.push_back(i); h_nz.push_back(j); }
    nz_tiles = (int)(h_nz.size() / 2);
    total_rows = (int64_t)h_rows.size();
    if (h_rows.empty()) h_rows.push_back(0);
    if (h_pairs.empty()) { h_pairs.push_back(0); h_pairs.push_back(0); }
    d_rows.upload(h_rows, stream); d_pairs.upload(h_pairs, stream); d_nz.upload(h_nz, stream);
    A.alloc((size_t)npad * npad); Linv.alloc((size_t)nt * CH_NB * CH_NB);
    Linv.zero(stream); // chol_potrf_block writes the lower triangle of every inverse only
    vb.alloc(npad); vy.alloc(npad); vx.alloc(npad); partial.alloc((size_t)std::max(max_rows, 1) * CH_NB);
    d_fail.alloc(1);
    if (!h_fail) GR_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_fail), sizeof(int), hipHostMallocDefault));
    if (!attrs_set) {
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chol_gemm<T, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_gemm_lds(sizeof(T))));
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chol_gemm<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_gemm_lds(sizeof(T))));
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chol_gemm<T, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_gemm_lds(sizeof(T))));
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chol_potrf<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_potrf_lds(sizeof(T))));
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chol_gemm<T, 1, 1, CH_KC, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(chol_gemm_lds(sizeof(T)), chol_potrf_lds(sizeof(T)))));
      attrs_set = true;
    }
    GR_HIP(hipStreamSynchronize(stream));
  }
  std::vector<int> col_end_, next_end_;
  std::vector<char> fused_next; // super-panel p's update also factorises diagonal tile 2(p+1)
  int ld() const { return npad; }
  double factor_flops() const { return (total_rows + total_pairs128 + 2.0 * total_pairs) * 2.0 * CH_NB * CH_NB * CH_NB + nt * (2.0 / 3.0) * CH_NB * CH_NB * CH_NB; }

  void clear() {
    k_chol_clear<T><<<nz_tiles, 256, 0, stream>>>(A.p, npad, n, d_nz.p);
  }
  struct Sc {
    CholProfSink *s;
    Sc(CholProfSink *s_, const char *nm, double by, double fl) : s(s_) { if (s) s->begin(nm, by, fl); }
    ~Sc() { if (s) s->end(); }
  };
  static constexpr double tile_b() { return (double)CH_NB * CH_NB * sizeof(T); }
  static constexpr double tile_f() { return 2.0 * CH_NB * CH_NB * CH_NB; }
  // panel phase of super-panel p on stream q: potrf(a), solve(a), column b update, potrf(b), solve(b)
  // panel phase of super-panel p: [potrf(a) unless the previous update already did it], solve(a), column b
  // update, potrf(b), solve(b)
  void panel_phase(int p, hipStream_t q, CholProfSink *sk, bool a_done) {
    const int a = 2 * p, b = a + 1;
    const size_t lds_g = chol_gemm_lds(sizeof(T)), lds_p = chol_potrf_lds(sizeof(T));
    T *La = Linv.p + (size_t)a * CH_NB * CH_NB, *Lb = La + CH_NB * CH_NB;
    if (!a_done) {
      Sc sc(sk, "chol_potrf", 3 * tile_b(), tile_f() / 3);
      k_chol_potrf<T><<<1, CH_PT, lds_p, q>>>(A.p, npad, a * CH_NB, La, d_fail.p, potrf_skip);
    }
    if (b >= nt) return;
    const int nra = row_end[a] - row_beg[a], nrb = row_end[b] - row_beg[b], ncol = col_end_[p] - col_off[p];
    {
      Sc sc(sk, "chol_trsm", (2.0 * nra + 1) * tile_b(), nra * tile_f());
      k_chol_gemm<T, 0><<<nra, 256, lds_g, q>>>(A.p, npad, d_rows.p + row_beg[a], a * CH_NB, La, CH_NB / CH_KC);
    }
    {
      Sc sc(sk, "chol_syrk_col", (3.0 * ncol) * tile_b(), ncol * tile_f());
      k_chol_gemm<T, 1><<<ncol, 256, lds_g, q>>>(A.p, npad, d_pairs.p + 2 * (size_t)col_off[p], a * CH_NB, nullptr, CH_NB / CH_KC);
    }
    {
      Sc sc(sk, "chol_potrf", 3 * tile_b(), tile_f() / 3);
      k_chol_potrf<T><<<1, CH_PT, lds_p, q>>>(A.p, npad, b * CH_NB, Lb, d_fail.p, potrf_skip);
    }
    if (nrb) {
      Sc sc(sk, "chol_trsm", (2.0 * nrb + 1) * tile_b(), nrb * tile_f());
      k_chol_gemm<T, 0><<<nrb, 256, lds_g, q>>>(A.p, npad, d_rows.p + row_beg[b], b * CH_NB, Lb, CH_NB / CH_KC);
    }
  }
  // trailing update of super-panel p (K = 256); returns true when it also factorised diagonal tile 2(p+1)
  bool update(int p, hipStream_t q, CholProfSink *sk) {
    const int beg = next_off[p], end = col_off[p + 1];
    if (end <= beg) return false;
    const int np_ = end - beg;
    const bool fuse = fuse_potrf && fused_next[p];
    Sc sc(sk, "chol_syrk", (2.0 * np_ + 2.0 * std::sqrt(2.0 * np_)) * tile_b(), 2.0 * np_ * tile_f() + (fuse ? tile_f() / 3 : 0.0));
    const int *pairs = d_pairs.p + 2 * (size_t)beg;
    if (fuse) {
      const size_t lds = std::max(chol_gemm_lds(sizeof(T)), chol_potrf_lds(sizeof(T)));
      k_chol_gemm<T, 1, 1, CH_KC, true><<<np_, 256, lds, q>>>(A.p, npad, pairs, 2 * p * CH_NB, nullptr, 2 * CH_NB / CH_KC, Linv.p + (size_t)(2 * p + 2) * CH_NB * CH_NB, d_fail.p);
    } else if (pin_variant) k_chol_gemm<T, 1, 1><<<np_, 256, chol_gemm_lds(sizeof(T)), q>>>(A.p, npad, pairs, 2 * p * CH_NB, nullptr, 2 * CH_NB / CH_KC);
    else k_chol_gemm<T, 1><<<np_, 256, chol_gemm_lds(sizeof(T)), q>>>(A.p, npad, pairs, 2 * p * CH_NB, nullptr, 2 * CH_NB / CH_KC);
    return fuse;
  }
  void factor() {
    GR_HIP(hipMemsetAsync(d_fail.p, 0, sizeof(int), stream));
    bool a_done = false;
    for (int p = 0; p < nsp; ++p) {
      panel_phase(p, stream, sink, a_done);
      a_done = update(p, stream, sink);
    }
  }
  // b, x: device vectors of length n (x may alias b)
  void solve(const T *b, T *x) {
    Sc sc(sink, "chol_solve", 2.0 * (total_rows + 2.0 * nt) * CH_NB * CH_NB * sizeof(T), 4.0 * (total_rows + nt) * CH_NB * CH_NB);
    k_chol_rhs<T><<<(npad + 255) / 256, 256, 0, stream>>>(n, npad, b, vb.p);
    for (int k = 0; k < nt; ++k) {
      const int nr = row_end[k] - row_beg[k];
      k_chol_fwd<T><<<1 + nr, 256, 0, stream>>>(A.p, npad, Linv.p + (size_t)k * CH_NB * CH_NB, k * CH_NB, d_rows.p + row_beg[k], vb.p, vy.p);
    }
    for (int k = nt - 1; k >= 0; --k) {
      const int nr = row_end[k] - row_beg[k];
      if (nr) k_chol_bwd_partial<T><<<nr, 256, 0, stream>>>(A.p, npad, k * CH_NB, d_rows.p + row_beg[k], vx.p, partial.p);
      k_chol_bwd_final<T><<<1, 256, 0, stream>>>(Linv.p + (size_t)k * CH_NB * CH_NB, k * CH_NB, nr, partial.p, vy.p, vx.p);
    }
    GR_HIP(hipMemcpyAsync(x, vx.p, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
  }
  // true when every pivot was positive (synchronises the stream)
  bool ok() {
    GR_HIP(hipMemcpyAsync(h_fail, d_fail.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
    return *h_fail == 0;
  }
};

} // namespace gr
