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
// runtime's own path.  A half of the ring is reused once the copies that read it have finished (an event per copy).
struct UploadRing {
  static constexpr size_t CAP = (size_t)64 << 20, HALF = CAP / 2;
  char *base = nullptr;
  size_t used = 0;
  // A half of the ring is written again only when every copy that read it has finished: one event per copy, recorded on the copy's
  // own stream (the wrap used to synchronise the CURRENT device only — a thread that drives engines on two devices or
  // streams could overwrite staging memory with a copy to the other one still in flight).  Events belong to the device they were
  // created on: the pools are per device.
  struct Pending { hipEvent_t ev; int dev; };
  std::vector<Pending> inflight[2];
  std::vector<Pending> spare;
  ~UploadRing() {
    for (auto &h : inflight) for (auto &p : h) (void)hipEventDestroy(p.ev);
    for (auto &p : spare) (void)hipEventDestroy(p.ev);
    if (base) (void)hipHostFree(base);
  }
  void ensure() { if (!base) GR_HIP(hipHostMalloc(reinterpret_cast<void **>(&base), CAP, hipHostMallocDefault)); }
  void drain(int half) {
    for (auto &p : inflight[half]) { GR_HIP(hipEventSynchronize(p.ev)); spare.push_back(p); }
    inflight[half].clear();
  }
  void mark(int half, hipStream_t s) {
    int dev = 0;
    GR_HIP(hipGetDevice(&dev));
    Pending p{nullptr, dev};
    for (size_t i = 0; i < spare.size(); ++i)
      if (spare[i].dev == dev) { p = spare[i]; spare[i] = spare.back(); spare.pop_back(); break; }
    if (!p.ev) GR_HIP(hipEventCreateWithFlags(&p.ev, hipEventDisableTiming));
    GR_HIP(hipEventRecord(p.ev, s));
    inflight[half].push_back(p);
  }
  void upload(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    ensure();
    // larger than half the ring: in half-ring pieces (a pageable hipMemcpyAsync of Final-13682's 107 MB of points ran at ~5 GB/s and
    // finished INSIDE the next call that synchronised: 15-23 ms of "set-up" in a 3-iteration LM call), then waited for here
    const bool large = bytes > HALF;
    for (size_t off = 0; off < bytes;) {
      const size_t chunk = std::min(bytes - off, HALF), need = (chunk + 255) & ~(size_t)255;
      // a piece never straddles the two halves: it starts the next half (or wraps) instead
      if (used < HALF && used + need > HALF) used = HALF;
      if (used + need > CAP) used = 0;
      const int half = used < HALF ? 0 : 1;
      if (used == 0 || used == HALF) drain(half); // entering a half: every copy that read it has finished
      std::memcpy(base + used, static_cast<const char *>(src) + off, chunk);
      GR_HIP(hipMemcpyAsync(static_cast<char *>(dst) + off, base + used, chunk, hipMemcpyHostToDevice, s));
      mark(half, s);
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
