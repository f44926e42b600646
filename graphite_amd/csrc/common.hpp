// Shared host/device helpers of libgraphite_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace gr {

struct HipError : std::runtime_error {
  explicit HipError(const std::string &m) : std::runtime_error(m) {}
};

#define GR_HIP(expr)                                                                               \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      throw ::gr::HipError(std::string(#expr) + " -> " + hipGetErrorString(_e) + " @" + __FILE__ + \
                           ":" + std::to_string(__LINE__));                                        \
  } while (0)

// host loops over the observation list (set-up phases) on a few threads: fn(begin, end, chunk) over `nt` contiguous chunks;
// an exception of any chunk is rethrown in the caller
inline int host_threads() { return (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency())); }
template <typename F> inline void par_chunks(size_t n, int nt, F &&fn) {
  if (nt <= 1 || n < (size_t)(1 << 14)) { fn((size_t)0, n, 0); return; }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err(nt);
  const size_t per = (n + nt - 1) / nt;
  for (int k = 0; k < nt; ++k)
    th.emplace_back([&, k] {
      try { fn(std::min(n, (size_t)k * per), std::min(n, (size_t)(k + 1) * per), k); } catch (...) { err[k] = std::current_exception(); }
    });
  for (auto &t : th) t.join();
  for (auto &e : err) if (e) std::rethrow_exception(e);
}

// Host -> device uploads of the set-up phase go through a pinned staging ring (one per host thread, 64 MB, allocated on
// first use or by gr_warm_up): a hipMemcpyAsync from PAGEABLE memory is staged by the runtime at ~2 GB/s and blocks the next
// synchronising call for as long (measured: 20 ms for the 40 MB of index arrays of Ladybug-1723 inside gr_bal_create);
// from pinned memory the same bytes take 2 ms and the call returns at once.  Arrays larger than half the ring take the
// runtime's own path.  The ring wraps behind a device synchronisation.
struct UploadRing {
  static constexpr size_t CAP = (size_t)64 << 20;
  char *base = nullptr;
  size_t used = 0;
  ~UploadRing() { if (base) (void)hipHostFree(base); }
  void ensure() { if (!base) GR_HIP(hipHostMalloc(reinterpret_cast<void **>(&base), CAP, hipHostMallocDefault)); }
  void upload(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    ensure();
    // larger than half the ring: in half-ring pieces (a pageable hipMemcpyAsync of Final-13682's 107 MB of points ran at ~5 GB/s and
    // finished INSIDE the next call that synchronised: 15-23 ms of "set-up" in a 3-iteration LM call), then waited for here
    const bool large = bytes > CAP / 2;
    for (size_t off = 0; off < bytes;) {
      const size_t chunk = std::min(bytes - off, CAP / 2), need = (chunk + 255) & ~(size_t)255;
      if (used + need > CAP) { GR_HIP(hipDeviceSynchronize()); used = 0; }
      std::memcpy(base + used, static_cast<const char *>(src) + off, chunk);
      GR_HIP(hipMemcpyAsync(static_cast<char *>(dst) + off, base + used, chunk, hipMemcpyHostToDevice, s));
      used += need; off += chunk;
    }
    if (large) GR_HIP(hipStreamSynchronize(s));
  }
  static UploadRing &get() { static thread_local UploadRing r; return r; }
};

// Device buffer owned by the engine (plain hipMalloc; sized for 288 GB HBM, so
// nothing is pooled or re-used across stages: every stage keeps its arrays
// resident for the whole optimisation).
template <typename T> struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    if (count == n && p) return;
    release();
    n = count;
    if (count) GR_HIP(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
  }
  void upload(const std::vector<T> &h, hipStream_t s) {
    alloc(h.size());
    if (!h.empty()) UploadRing::get().upload(p, h.data(), h.size() * sizeof(T), s);
  }
  void zero(hipStream_t s) {
    if (n) GR_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
  std::vector<T> download(hipStream_t s) const {
    std::vector<T> h(n);
    if (n) {
      GR_HIP(hipMemcpyAsync(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, s));
      GR_HIP(hipStreamSynchronize(s));
    }
    return h;
  }
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

} // namespace gr
