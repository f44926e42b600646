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

} // namespace gr

#include "../../include/graphite_mi355x_device.hpp"
