// Device-side helpers shared by libgraphite_mi355x.so (graphite_amd/csrc) and the header-only kernels that are
// instantiated on USER traits in the user's translation unit (include/graphite/engine_model.hpp): wave64 cross-lane
// exchanges, fixed-order reductions, dot-product slots, the XCD-aware observation ranges of the persistent per-observation
// kernels.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace gr {

constexpr int TPB = 256;

template <typename T> struct Vec2T;
template <> struct Vec2T<float> { using type = float2; };
template <> struct Vec2T<double> { using type = double2; };

// Device-side hand-over of the LM accept decision.  On an accept streak the host enqueues the FIRST kernels of iteration
// i + 1 (k_finalize_bj, first direction, first PCG iteration) while the trial linearisation of iteration i is still running,
// i.e. before anybody knows whether step i is accepted.  k_finalize_bj takes that decision on the device
// (optimizer/levenberg_marquardt.hpp:184-197), leaves the new damping here and sets `stop` when the step is not accepted;
// the kernels that follow it read the damping from here and return at once when `stop` is set.  The host only OBSERVES the
// decision (pinned memory) and runs the rejection path itself.
struct LmDev {
  double mu;
  int stop; // 0 run | 2 the step was not accepted
  int hsel; // Schur solvers with two camera-point block buffers — bit 0: the one that holds the CURRENT point's blocks (the trial
            // linearisation writes the other one; the finalisation that takes a new linearisation over flips it, a rejected step leaves
            // it); bit 1 (user-traits problems): the vertices still sit at a rejected trial point — the next step restores its backup first
};

// ---- cross-lane exchange without LDS ------------------------------------------------------------------------------------
// lane i <- lane i ^ OFFSET for OFFSET = 1, 2, 4, 8 as DPP moves (plain VALU instructions): quad_perm for 1 and 2, row_ror:8 for
// 8, two bank-masked row rotations for 4.  hipcc lowers __shfl_xor to ds_bpermute_b32 whatever the pattern — an LDS-crossbar
// instruction with ~100 cycles of latency on the dependent chains of the butterflies below (the kernels had 43-294 of them and
// not one DPP move).  OFFSET 16 / 32 keep the bpermute here; the transpose reductions use v_permlane16/32_swap for those.
template <int CTRL, int BANK> __device__ __forceinline__ unsigned dpp_mov(unsigned old, unsigned src) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, BANK, false);
}
template <int OFFSET> __device__ __forceinline__ unsigned lane_xor_u32(unsigned v) {
  static_assert(OFFSET == 1 || OFFSET == 2 || OFFSET == 4 || OFFSET == 8, "DPP forms exist for 1, 2, 4, 8");
  if constexpr (OFFSET == 1) return dpp_mov<0xB1, 0xf>(v, v);      // quad_perm [1, 0, 3, 2]
  else if constexpr (OFFSET == 2) return dpp_mov<0x4E, 0xf>(v, v); // quad_perm [2, 3, 0, 1]
  else if constexpr (OFFSET == 8) return dpp_mov<0x128, 0xf>(v, v); // row_ror:8
  else { // row_ror:n: lane i of a 16-lane row reads lane (i - n) mod 16.  Lanes 4-7, 12-15 (banks 1, 3) take i - 4, the others i + 4 = i - 12
    const unsigned t = dpp_mov<0x124, 0xA>(v, v);
    return dpp_mov<0x12C, 0x5>(t, v);
  }
}
template <int OFFSET> __device__ __forceinline__ float lane_xor(float v) { return __builtin_bit_cast(float, lane_xor_u32<OFFSET>(__builtin_bit_cast(unsigned, v))); }
template <int OFFSET> __device__ __forceinline__ int lane_xor(int v) { return (int)lane_xor_u32<OFFSET>((unsigned)v); }
template <int OFFSET> __device__ __forceinline__ double lane_xor(double v) {
  uint2 u = __builtin_bit_cast(uint2, v);
  u.x = lane_xor_u32<OFFSET>(u.x); u.y = lane_xor_u32<OFFSET>(u.y);
  return __builtin_bit_cast(double, u);
}
// Orders the LDS accesses of ONE wave whose lanes exchange data through LDS without a workgroup barrier: the hardware runs a
// wave's LDS instructions in issue order, but without this the compiler may move a lane's reads above another lane's writes.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// v + (v of lane ^ OFFSET) for the six butterfly steps of a wave
template <typename T> __device__ __forceinline__ T butterfly_low(T v) { // offsets 8, 4, 2, 1: DPP
  v += lane_xor<8>(v); v += lane_xor<4>(v); v += lane_xor<2>(v); v += lane_xor<1>(v);
  return v;
}

// gfx950 half / row exchanges: v_permlane32_swap swaps lanes 32-63 of `a` with lanes 0-31 of `b`, v_permlane16_swap the odd
// 16-lane rows of `a` with the even rows of `b`.  After the swap every lane holds (its own value, its partner's) in (a, b) or
// (b, a); with a == b == v that is the xor-32 / xor-16 butterfly step, again without LDS.
template <int OFFSET> __device__ __forceinline__ void lane_swap(unsigned &a, unsigned &b) {
  if constexpr (OFFSET == 32) { const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; }
  else { const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1]; }
}
template <int OFFSET> __device__ __forceinline__ float swap_add(float a, float b) {
  unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
  lane_swap<OFFSET>(ua, ub);
  return __builtin_bit_cast(float, ua) + __builtin_bit_cast(float, ub);
}
template <int OFFSET> __device__ __forceinline__ double swap_add(double a, double b) {
  uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  lane_swap<OFFSET>(ua.x, ub.x);
  lane_swap<OFFSET>(ua.y, ub.y);
  return __builtin_bit_cast(double, ua) + __builtin_bit_cast(double, ub);
}

// ---- wave64 / block reductions -----------------------------------------------
// butterfly sum: every lane gets the wave total.  All six steps without LDS: permlane swaps for 32 / 16, DPP for 8 / 4 / 2 / 1;
// the partners and the order of the steps are those of the __shfl_xor butterfly, so the bits are too.
template <typename T> __device__ __forceinline__ T wave_allsum(T v) {
  v = swap_add<32>(v, v);
  v = swap_add<16>(v, v);
  return butterfly_low(v);
}
template <typename T> __device__ __forceinline__ T wave_sum(T v) { return wave_allsum(v); } // (valid in every lane)

// Sum over a 256-thread block; result valid in thread 0.  smem: >= 4 T.
template <typename T> __device__ __forceinline__ T block_sum_256(T v, T *smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) v = smem[0] + smem[1] + smem[2] + smem[3];
  return v;
}

// Inclusive segmented sum over runs of equal `key` in a wave (keys sorted, so a
// match at distance o implies the whole span matches).  After the call the LAST
// lane of each run holds the run total.
template <typename T, int NV> __device__ __forceinline__ void seg_scan(T (&v)[NV], int key, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int kk = __shfl_up(key, o, 64);
    const bool ok = (lane >= o) && (kk == key);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const T t = __shfl_up(v[i], o, 64);
      if (ok) v[i] += t;
    }
  }
}

// Sum NV (= 16 or 64) per-lane values over the 64 lanes with NV-1 (+2) shuffles instead
// of 6*NV: at every halving step a lane keeps one half of its values and hands the
// other half to its partner (lane ^ offset).  On return lane L holds the wave total of
// value index (NV == 64 ? L : L >> 2).  Template recursion keeps every array index a
// compile-time constant (the array must stay in VGPRs).
template <typename T, int NV, int HALF, int OFFSET> struct TransposeStep {
  static __device__ __forceinline__ void run(T (&v)[NV], int lane) {
    if constexpr (OFFSET == 32 || OFFSET == 16) {
#pragma unroll
      for (int i = 0; i < HALF; ++i) v[i] = swap_add<OFFSET>(v[i], v[i + HALF]);
    } else {
      const bool hi = (lane & OFFSET) != 0;
#pragma unroll
      for (int i = 0; i < HALF; ++i) {
        const T keep = hi ? v[i + HALF] : v[i];
        const T send = hi ? v[i] : v[i + HALF];
        v[i] = keep + lane_xor<OFFSET>(send);
      }
    }
    TransposeStep<T, NV, HALF / 2, OFFSET / 2>::run(v, lane);
  }
};
template <typename T, int NV, int OFFSET> struct TransposeStep<T, NV, 0, OFFSET> {
  static __device__ __forceinline__ void run(T (&)[NV], int) {}
};
template <typename T, int NV> __device__ __forceinline__ T wave_transpose_sum(T (&v)[NV], int lane) {
  static_assert(NV == 64 || NV == 16, "NV must be 16 or 64");
  TransposeStep<T, NV, NV / 2, 32>::run(v, lane);
  T r = v[0];
  if (NV == 16) { r += lane_xor<2>(r); r += lane_xor<1>(r); }
  return r;
}

// Dot-product accumulators.  Thousands of workgroups adding to ONE address
// serialise at ~12 ns per atomic (MI355X_MICROARCH.md "fanin"), which is longer
// than the kernels themselves; every logical scalar is therefore NS partial
// sums, picked by blockIdx, and re-summed (fixed order) by its readers.
// SS = doubles between two partials.  One 128-byte line per partial (SS = 16) was measured against the packed
// layout (SS = 1) on the 1 900-block update kernel with its 4 sums: no difference (the fire-and-forget atomics
// of a finishing block are not what bounds these kernels), so the partials stay packed.
constexpr int NS = 64;        // partial sums per logical scalar
#ifndef GR_SS
#define GR_SS 1
#endif
constexpr int SS = GR_SS;     // doubles between partials
constexpr int NSW = NS * SS;  // doubles per logical scalar
__device__ __forceinline__ void slot_add(double *base, int k, double v) {
  atomicAdd(&base[(size_t)k * NSW + (size_t)(blockIdx.x & (NS - 1)) * SS], v);
}
__device__ __forceinline__ double slot_sum(const double *base, int k) { // whole wave must call
  return wave_allsum(base[(size_t)k * NSW + (size_t)(threadIdx.x & 63) * SS]);
}
// N logical scalars at once: the N loads are issued together and the butterflies advance in lock step, so a kernel prologue
// that needs several dot products pays ONE memory round trip and one shuffle chain instead of N of each (every wave of every
// PCG kernel starts with these sums; measured on Ladybug-1723: update 20.4 -> 19.9 us on average, direction unchanged).  Same
// butterfly order as slot_sum: same bits.  Whole wave must call.
template <int N> __device__ __forceinline__ void slot_sums(const double *const (&base)[N], double (&out)[N]) {
  double v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = base[i][(size_t)(threadIdx.x & 63) * SS];
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = wave_allsum(v[i]);
}
// index of partial i (0 <= i < count * NS) of an array of logical scalars, for the loops that clear them
__device__ __forceinline__ size_t slot_word(int i) { return (size_t)(i / NS) * NSW + (size_t)(i % NS) * SS; }

__device__ __forceinline__ double clampd(double x, double lo, double hi) {
  return x < lo ? lo : (x > hi ? hi : x);
}

// damped diagonal, hessian.hpp:136-176 (computed in double as the reference does)
template <typename T> __device__ __forceinline__ T damp_diag(T d, double mu, int use_identity) {
  if (use_identity) return (T)((double)d + mu);
  return (T)((double)d + mu * clampd((double)d, 1.0e-6, 1.0e32));
}

// Column scale of a vertex entry in double, for the small per-vertex algebra (block inverses): graph.hpp:262-270 computes
// 1 / (eps + sqrt(diag)) in double even when T = float; the stored scale s is that value rounded to T.  Where s IS that value
// (scale_system on, vertex not fixed) the unrounded one is returned, so that the scaled block has an exactly unit diagonal and the
// inversion — which amplifies entry errors by the block's condition number, 1e3-1e4 for weakly observed points — does not see the
// fp32 rounding of the scale; otherwise (scaling off: 1, fixed vertex: 1) the stored value.
template <typename T> __device__ __forceinline__ double scale_hat(T s, T hii) {
  const double sh = 1.0 / (2.220446049250313e-16 + sqrt((double)hii));
  const double sd = (double)s;
  return fabs(sd - sh) <= 1e-6 * sh ? sh : sd;
}

// In-register inverse of a symmetric positive definite N x N block (column-major
// in/out), Gauss-Jordan without pivoting, fully unrolled so that the block
// stays in VGPRs.  The role of cublas<t>matinvBatched (schur.hpp:1101,
// block_jacobi.hpp:154, block_jacobi_schur.hpp:140) for the 3x3 / 9x9 blocks.
// Arithmetic in double for both dtypes (Nc + Np threads only).
template <int N> __device__ __forceinline__ void spd_inverse(double (&A)[N * N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double piv = 1.0 / A[k + N * k];
#pragma unroll
    for (int c = 0; c < N; ++c) A[k + N * c] = (c == k) ? piv : A[k + N * c] * piv;
#pragma unroll
    for (int r = 0; r < N; ++r) {
      if (r == k) continue;
      const double f = A[r + N * k];
#pragma unroll
      for (int c = 0; c < N; ++c) A[r + N * c] = (c == k) ? -f * piv : A[r + N * c] - f * A[k + N * c];
    }
  }
}


// The XCD-aware partition of the observation list of the persistent per-observation kernels as an observation range
// [j0, jlim) walked in steps of jstride, niter workgroup tiles (see kernels_mf.hpp: xcd_tile_range for the tile form).
// Plain form (ntiles >= 0): the ranges are cut at 64-observation WAVE blocks, not at 256-observation workgroup tiles —
// Ladybug-1723 has 3.45 tiles per workgroup, i.e. 3 or 4, and a CU that drew three 4-tile workgroups carried 48 wave blocks
// against an average of 41.4 (the kernels are bound per CU by their memory instructions); cut by wave blocks every workgroup
// has 13 or 14.  Its last tile is then partly empty (whole waves idle: ranges start and end on 64-observation boundaries, so
// the (wave, camera) segments are unchanged).  ntiles < 0: the SWEEP form of point-tiled observation orders.
__device__ __forceinline__ void xcd_obs_range(int ntiles, int No, int &j0, int &jstride, int &niter, int &jlim) {
  const int nb = gridDim.x >> 3, x = blockIdx.x & 7, bi = blockIdx.x >> 3;
  if (ntiles < 0) {
    const int nt = -ntiles;
    const int x0 = (int)((long long)x * nt / 8), x1 = (int)((long long)(x + 1) * nt / 8), t0 = x0 + bi;
    j0 = t0 * TPB; jstride = nb * TPB;
    niter = t0 < x1 ? (x1 - t0 + nb - 1) / nb : 0;
    const long long lim = (long long)x1 * TPB;
    jlim = lim < (long long)No ? (int)lim : No;
    return;
  }
  const int nblk = (No + 63) >> 6;
  const int x0 = (int)((long long)x * nblk / 8), x1 = (int)((long long)(x + 1) * nblk / 8);
  const int b0 = x0 + (int)((long long)bi * (x1 - x0) / nb), b1 = x0 + (int)((long long)(bi + 1) * (x1 - x0) / nb);
  j0 = b0 << 6;
  const long long lim = (long long)b1 << 6;
  jlim = lim < (long long)No ? (int)lim : No;
  jstride = TPB;
  niter = jlim > j0 ? (jlim - j0 + TPB - 1) / TPB : 0;
}

} // namespace gr
