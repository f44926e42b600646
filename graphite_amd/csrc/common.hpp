// Shared host/device helpers of libgraphite_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
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
    if (!h.empty()) GR_HIP(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
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

// ---- wave64 / block reductions -----------------------------------------------
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v; // valid in lane 0
}

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
