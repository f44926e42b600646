// Communication layer of the landmark-sharded solver: one process per GPU, RCCL over xGMI.
//
// The reference is single-GPU; this is the MI355X addition of SURVEY §8(e):
// points (and all their observations) are split over the ranks, cameras are replicated, and
// only camera-space sums travel: per linearisation [Hcc (81 Nc), bc (9 Nc), chi2], per PCG
// iteration the camera rows of the operator (9 Nc) plus a 64-slot record of partial dot
// products — two latency-bound all-reduces per iteration, no data-path collective on the
// observations or points.
//
// librccl is dlopen()ed when a communicator is created, so the library still loads on hosts
// without RCCL and shares the copy torch.distributed has already mapped.
// LocalGroup is an in-process stand-in (one engine per host thread, same GPU) used by the
// tests to exercise the sharded algorithm on a 1-GPU box; it is not a product path.
#pragma once
#include "common.hpp"
#include <condition_variable>
#include <dlfcn.h>
#include <cstring>
#include <memory>
#include <mutex>

namespace gr {

struct Comm {
  int rank = 0, size = 1;
  virtual ~Comm() = default;
  // in-place sum over ranks of `count` scalars (is_double selects the element type)
  virtual void allreduce(void *buf, size_t count, bool is_double, hipStream_t stream) = 0;
  virtual void group_start() {}
  virtual void group_end() {}
};

// ---- RCCL ------------------------------------------------------------------------------
struct RcclApi {
  struct UID { char b[128]; }; // ncclUniqueId (rccl.h:43), passed by value
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, UID, int) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  static RcclApi &get() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
      }
      if (!api.lib) return;
      api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
      api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
      api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
      api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
      api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.lib, "ncclGroupStart"));
      api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.lib, "ncclGroupEnd"));
      api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
    });
    return api;
  }
  bool ok() const { return lib && GetUniqueId && CommInitRank && AllReduce && CommDestroy && GroupStart && GroupEnd; }
  void check(int rc, const char *what) const {
    if (rc != 0) throw std::runtime_error(std::string("RCCL ") + what + ": " + (GetErrorString ? GetErrorString(rc) : "error"));
  }
};

struct RcclComm final : Comm {
  void *comm = nullptr;
  RcclComm(const void *unique_id_128, int rank_, int size_) {
    rank = rank_; size = size_;
    RcclApi &api = RcclApi::get();
    if (!api.ok()) throw std::runtime_error("librccl.so.1 not found");
    RcclApi::UID id;
    std::memcpy(id.b, unique_id_128, 128);
    api.check(api.CommInitRank(&comm, size, id, rank), "ncclCommInitRank");
  }
  ~RcclComm() override {
    if (comm) (void)RcclApi::get().CommDestroy(comm);
  }
  void allreduce(void *buf, size_t count, bool is_double, hipStream_t stream) override {
    // ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0 (rccl.h)
    RcclApi &api = RcclApi::get();
    api.check(api.AllReduce(buf, buf, count, is_double ? 8 : 7, 0, comm, stream), "ncclAllReduce");
  }
  void group_start() override { RcclApi::get().check(RcclApi::get().GroupStart(), "ncclGroupStart"); }
  void group_end() override { RcclApi::get().check(RcclApi::get().GroupEnd(), "ncclGroupEnd"); }
};

// ---- in-process test backend -------------------------------------------------------------
template <typename T> __global__ void k_sum_into_all(size_t count, T *const *bufs, int nbufs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  T s = 0;
  for (int r = 0; r < nbufs; ++r) s += bufs[r][i]; // fixed rank order
  for (int r = 0; r < nbufs; ++r) bufs[r][i] = s;
}

struct LocalGroup {
  int size;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long generation = 0;
  std::vector<void *> bufs;
  void **d_bufs = nullptr;
  explicit LocalGroup(int n) : size(n), bufs(n, nullptr) { GR_HIP(hipMalloc(reinterpret_cast<void **>(&d_bufs), n * sizeof(void *))); }
  ~LocalGroup() { if (d_bufs) (void)hipFree(d_bufs); }
  template <typename F> void barrier(F &&last_arriver) {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long long gen = generation;
    if (++arrived == size) {
      last_arriver();
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else cv.wait(lk, [&] { return generation != gen; });
  }
};

struct LocalComm final : Comm {
  std::shared_ptr<LocalGroup> g;
  LocalComm(std::shared_ptr<LocalGroup> grp, int rank_) : g(std::move(grp)) { rank = rank_; size = g->size; }
  void allreduce(void *buf, size_t count, bool is_double, hipStream_t stream) override {
    GR_HIP(hipStreamSynchronize(stream)); // my producers are done
    {
      std::lock_guard<std::mutex> lk(g->m);
      g->bufs[rank] = buf;
    }
    g->barrier([&] {
      GR_HIP(hipMemcpy(g->d_bufs, g->bufs.data(), g->size * sizeof(void *), hipMemcpyHostToDevice));
      const int grid = (int)((count + 255) / 256);
      if (is_double) k_sum_into_all<double><<<grid, 256, 0, stream>>>(count, reinterpret_cast<double *const *>(g->d_bufs), g->size);
      else k_sum_into_all<float><<<grid, 256, 0, stream>>>(count, reinterpret_cast<float *const *>(g->d_bufs), g->size);
      GR_HIP(hipStreamSynchronize(stream));
    });
  }
};

} // namespace gr
