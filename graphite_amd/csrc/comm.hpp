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
#include <algorithm>

namespace gr {

struct CommError : std::runtime_error { using std::runtime_error::runtime_error; };
struct Comm {
  int rank = 0, size = 1;
  virtual ~Comm() = default;
  // in-place sum over ranks of `count` scalars (is_double selects the element type)
  virtual void allreduce(void *buf, size_t count, bool is_double, hipStream_t stream) = 0;
  virtual void group_start() {}
  virtual void group_end() {}
  // a transport that can lose a message without hanging reports it here (host check after a stream synchronisation)
  virtual bool failed() { return false; }
  virtual void set_timeout_ms(int) {}
  // what a SCALE record is audited with (gr_bal_comm_info): 1 RCCL, 2 IPC mailboxes, 3 in-process test group
  virtual int transport() const { return 0; }
  virtual int rccl_ranks() { return 0; }          // ncclCommCount of the RCCL communicator in use (0: none)
  virtual int mailboxes_opened() const { return 0; } // peer mailboxes mapped through hipIpcOpenMemHandle
  virtual void message_counts(int64_t &oneshot, int64_t &fallback_) const { oneshot = 0; fallback_ = 0; }
};

// ---- RCCL ------------------------------------------------------------------------------
struct RcclApi {
  struct UID { char b[128]; }; // ncclUniqueId (rccl.h:43), passed by value
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, UID, int) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*CommCount)(void *, int *) = nullptr;
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
      api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
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
  int transport() const override { return 1; }
  int rccl_ranks() override {
    int n = 0;
    RcclApi &api = RcclApi::get();
    if (!api.CommCount || api.CommCount(comm, &n) != 0) return -1;
    return n;
  }
};

// ---- one-shot peer all-reduce over IPC-mapped mailboxes ----------------------------------------------
// The messages of this solver are small (<= 9 Nc scalars + a few dot-product records: 64 KB on Venice-1778 fp32) and
// there are several per LM iteration, so what a collective costs is its latency: ~16 us through RCCL even on one
// rank.  Over xGMI every GPU can write into every other GPU's memory directly, so an all-reduce of a
// small message is ONE hop: every rank stores its contribution into its slot of every peer's mailbox (`push`), then
// sums the `size` slots of its own mailbox in rank order (`reduce`).  No ring, no tree; the sum has the same order on
// every rank, so the result is bit-identical everywhere.
//   mailbox (one per rank, FINE-GRAINED device memory: hipExtMallocWithFlags(hipDeviceMallocFinegrained) + hipIpcGetMemHandle,
//   opened by every peer with hipIpcOpenMemHandle):
//     header: flags[2][size] (uint64 sequence numbers), error word;  then 2 sets x size slots of `slot_bytes`
//   set = seq & 1: a rank can only push sequence s + 2 after it has reduced s + 1, which needed every peer's push of
//   s + 1, issued (stream order) after that peer's reduce of s: two sets are enough.
//   visibility: the payload is stored with system-scope (write-through) stores, drained, then the flag is stored
//   system-scope by the block that finishes last; the reader polls the flags system-scope and fences before it loads.
// Grouped calls (group_start .. group_end) travel as ONE message.  Messages beyond the slot size go to the fallback
// communicator (RCCL).  Every spin is bounded (gr_bal_tuning.ipc_timeout_ms, default 30 s: ranks may arrive seconds apart
// after host-side set-up work) and raises the mailbox's error word instead of hanging; the host checks that word after every
// all-reduce whose result IT consumes and before it returns from an optimisation, and a communicator that has failed once
// refuses every later collective (CommError) — a rank that timed out holds rank-local values and must not carry on.
constexpr int IPC_MAX_PARTS = 8;
constexpr size_t IPC_HEADER = 4096;
// FUSED channel (round 4): the per-inner-iteration message of the single-reduction PCG on landmark shards does not go through
// a kernel of its own.  The OPERATOR's finishing workgroups push this rank's camera rows and dot-product records straight
// into every peer's mailbox, and the UPDATE kernel's prologue waits for the flags and sums the slots in rank order: the
// all-reduce is the seam between two kernels that exist anyway (4 launches per inner iteration -> 2).  It has its own flags
// (header offset IPC_FUSED_FLAGS) and its own 2 x size slots behind the collective channel's, and its sequence number lives on
// the DEVICE (it advances only when a launch really pushes: launches that find the PCG loop finished return at once on every
// rank alike), so the two channels never disturb each other's double buffering.
constexpr size_t IPC_FUSED_FLAGS = 1024; // flags[2][size] of the fused channel (the collective channel's start at 0; size <= 64)
struct IpcFused {
  char *const *boxes = nullptr;       // device array [size] of mailbox base pointers as mapped on this rank; nullptr = not fused
  int rank = 0, size = 0;
  size_t slot_bytes = 0;
  unsigned long long *seq = nullptr;  // device: fused messages pushed so far by this rank (identical on all ranks between launches)
  unsigned *counter = nullptr;        // device: [0] completion count of the pushing launch
  long long timeout_ticks = 0;
  int *h_err = nullptr;
  // PROJECTION ONLY (tools/shard_projection.py, gr_bal_tuning.shard_virtual_ranks = V on a ONE-rank communicator): the rank
  // plays all V ranks of a V-rank message on its own mailbox — it stores its rows into slot 0 and zeros into slots 1 .. V - 1,
  // raises all V flags, and the consumer sums V slots: the stores, loads and waits of a V-rank message, without the xGMI hops
  int virt = 0;
  // [Nc] bit r: rank r holds observations of camera c (nullptr: not known, every rank pushes every row).  Shards cut by camera
  // locality see about 1 / size of the cameras: a rank pushes only the rows of its own cameras and a consumer sums, in rank order,
  // only the slots of a camera's contributors — the same bits as the full sum (the skipped terms were +0.0), 1 / size of the bytes
  const unsigned *contrib = nullptr;
  __device__ __forceinline__ unsigned contributors(unsigned c) const { return contrib ? contrib[c] : ~0u; }
  // bit r of a contributor mask; masks are 32 bits wide (gr_bal_comm_set_contributors: world_size <= 32): ranks beyond them only
  // exist without masks (who == ~0u), where every rank contributes — never a shift by >= 32
  static __device__ __forceinline__ bool rank_in(unsigned who, int r) { return r >= 32 || ((who >> r) & 1u); }
  __device__ __forceinline__ int push_box(int r) const { return virt ? 0 : r; }
  __device__ __forceinline__ int push_slot(int r) const { return virt ? r : rank; }
  template <typename U> __device__ __forceinline__ U push_value(int r, U v) const { return (virt && r) ? U(0) : v; }
  __device__ __forceinline__ char *slot(int box, int set, int r) const { return boxes[box] + IPC_HEADER + ((size_t)(2 + set) * size + r) * slot_bytes; }
  __device__ __forceinline__ unsigned long long *flag(int box, int set, int r) const {
    return reinterpret_cast<unsigned long long *>(boxes[box] + IPC_FUSED_FLAGS) + (size_t)set * size + r;
  }
};
struct IpcPart { void *ptr; unsigned long long count; unsigned long long offset; int is_double; int pad; }; // offset: bytes inside the slot
struct IpcMsg { IpcPart part[IPC_MAX_PARTS]; int nparts; };

template <typename T> __device__ __forceinline__ void ipc_store(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <typename T> __device__ __forceinline__ T ipc_load(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// One launch per all-reduce.  Phase 1: this rank's values -> slot `rank` of set `set` in EVERY mailbox (its own included);
// the last block to finish raises this rank's flag in every mailbox.  Phase 2: every block waits for all ranks' flags in
// its OWN mailbox, then buf = sum over ranks, in rank order, of the slots (identical bits on every rank).
// The grid is at most 64 blocks, launched on a stream whose earlier kernels have finished: all blocks are resident, so
// waiting inside the kernel cannot starve the blocks that still have to push.  Waits are bounded (timeout_ticks of the
// 100 MHz wall clock): a peer that died or never joined turns into the error word (IpcComm::failed), not a hung GPU.
__global__ void __launch_bounds__(256) k_ipc_allreduce(IpcMsg msg, char *const *__restrict__ boxes, int rank, int size, int set, size_t slot_bytes,
                                                       unsigned long long seq, unsigned *__restrict__ ticket, long long timeout_ticks, int *__restrict__ h_err) {
  // a communicator on which an earlier all-reduce timed out holds rank-local values: every collective already enqueued
  // behind the failed one returns at once (only the FIRST failure pays the wait bound; the host sees h_err / failed()).
  // The error word is read from this rank's own mailbox in HBM by ONE thread per workgroup — its pinned host mirror h_err
  // would be a PCIe read per thread (measured: +3 us per all-reduce, 180 us in a 1 500-workgroup kernel)
  __shared__ int s_dead;
  if (threadIdx.x == 0) s_dead = ipc_load(reinterpret_cast<const unsigned long long *>(boxes[rank]) + 500) != 0ull ? 1 : 0;
  __syncthreads();
  if (s_dead) return;
  const size_t slot_off = IPC_HEADER + ((size_t)set * size + rank) * slot_bytes;
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gstride = (size_t)gridDim.x * blockDim.x;
  for (int q = 0; q < msg.nparts; ++q) {
    const IpcPart pt = msg.part[q];
    if (pt.is_double) {
      const double *src = static_cast<const double *>(pt.ptr);
      for (size_t i = gtid; i < pt.count; i += gstride) {
        const double v = src[i];
        for (int r = 0; r < size; ++r) ipc_store(reinterpret_cast<double *>(boxes[r] + slot_off + pt.offset) + i, v);
      }
    } else {
      const float *src = static_cast<const float *>(pt.ptr);
      for (size_t i = gtid; i < pt.count; i += gstride) {
        const float v = src[i];
        for (int r = 0; r < size; ++r) ipc_store(reinterpret_cast<float *>(boxes[r] + slot_off + pt.offset) + i, v);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (tk == gridDim.x - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int r = 0; r < size; ++r)
        ipc_store(reinterpret_cast<unsigned long long *>(boxes[r]) + (size_t)set * size + rank, seq);
    }
  }
  char *box = boxes[rank];
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  if ((int)threadIdx.x < size) { // thread r waits for rank r
    const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(box) + (size_t)set * size + threadIdx.x;
    const long long t0 = wall_clock64();
    while (ipc_load(flag) < seq) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > timeout_ticks) {
        ipc_store(reinterpret_cast<unsigned long long *>(box) + 500, 1ull); // error word
        if (h_err) { *h_err = 1; __threadfence_system(); } // its pinned host mirror: IpcComm::failed() costs no copy
        s_bad = 1;
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if (s_bad) return;
  const char *slots = box + IPC_HEADER + (size_t)set * size * slot_bytes;
  for (int q = 0; q < msg.nparts; ++q) {
    const IpcPart pt = msg.part[q];
    if (pt.is_double) {
      double *dst = static_cast<double *>(pt.ptr);
      for (size_t i = gtid; i < pt.count; i += gstride) {
        double sum = 0;
        for (int r = 0; r < size; ++r) sum += ipc_load(reinterpret_cast<const double *>(slots + (size_t)r * slot_bytes + pt.offset) + i);
        dst[i] = sum;
      }
    } else {
      float *dst = static_cast<float *>(pt.ptr);
      for (size_t i = gtid; i < pt.count; i += gstride) {
        float sum = 0;
        for (int r = 0; r < size; ++r) sum += ipc_load(reinterpret_cast<const float *>(slots + (size_t)r * slot_bytes + pt.offset) + i);
        dst[i] = sum;
      }
    }
  }
}

struct IpcComm final : Comm {
  std::vector<char *> boxes;   // boxes[r] = rank r's mailbox as mapped here (boxes[rank] = my own allocation)
  std::vector<bool> opened;    // opened through hipIpcOpenMemHandle (to be closed)
  char **d_boxes = nullptr;
  unsigned *d_ticket = nullptr;
  int *h_err = nullptr; // pinned: raised by a kernel whose wait timed out
  size_t slot_bytes = 0;
  unsigned long long seq = 0;
  std::unique_ptr<Comm> fallback; // messages larger than a slot (may be null: then they are an error)
  bool grouping = false;
  IpcMsg pending{};
  size_t pending_bytes = 0;
  hipStream_t pending_stream = nullptr;
  int64_t n_oneshot = 0, n_fallback = 0;
  long long timeout_ticks = 3000000000ll; // 30 s at 100 MHz (gr_bal_tuning.ipc_timeout_ms)
  bool dead = false;                      // a wait timed out once: every later collective is refused

  static size_t mailbox_bytes(int size, size_t slot) { return IPC_HEADER + 4 * (size_t)size * slot; } // 2 sets x size slots per channel
  // the fused channel's view of this communicator (kernels_mf.hpp: k_pcg_operator<..., FUSE>, k_pcg_update)
  unsigned long long *d_seq2 = nullptr;
  unsigned *d_counter2 = nullptr;
  int virtual_ranks = 0;   // projection only (IpcFused::virt)
  size_t fused_slot_bytes() const { return (size == 1 && virtual_ranks > 1) ? ((slot_bytes / (size_t)virtual_ranks) & ~(size_t)255) : slot_bytes; }
  IpcFused fused() {
    if (!d_seq2) {
      GR_HIP(hipMalloc(reinterpret_cast<void **>(&d_seq2), sizeof(unsigned long long)));
      GR_HIP(hipMemset(d_seq2, 0, sizeof(unsigned long long)));
      GR_HIP(hipMalloc(reinterpret_cast<void **>(&d_counter2), 4 * sizeof(unsigned)));
      GR_HIP(hipMemset(d_counter2, 0, 4 * sizeof(unsigned)));
    }
    IpcFused f;
    f.boxes = d_boxes; f.rank = rank; f.size = size; f.slot_bytes = slot_bytes; f.seq = d_seq2; f.counter = d_counter2;
    f.timeout_ticks = timeout_ticks; f.h_err = h_err;
    if (size == 1 && virtual_ranks > 1 && virtual_ranks <= 64) { // the one rank's 4 slots cut into 4 x V smaller ones
      f.size = virtual_ranks; f.virt = 1;
      f.slot_bytes = (slot_bytes / (size_t)virtual_ranks) & ~(size_t)255;
    }
    return f;
  }
  IpcComm(int rank_, int size_, size_t slot_bytes_, const std::vector<char *> &boxes_, const std::vector<bool> &opened_) : boxes(boxes_), opened(opened_), slot_bytes(slot_bytes_) {
    rank = rank_; size = size_;
    GR_HIP(hipMalloc(reinterpret_cast<void **>(&d_boxes), size * sizeof(char *)));
    GR_HIP(hipMemcpy(d_boxes, boxes.data(), size * sizeof(char *), hipMemcpyHostToDevice));
    GR_HIP(hipMalloc(reinterpret_cast<void **>(&d_ticket), sizeof(unsigned)));
    GR_HIP(hipMemset(d_ticket, 0, sizeof(unsigned)));
    GR_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_err), sizeof(int), hipHostMallocCoherent | hipHostMallocMapped));
    *h_err = 0;
  }
  ~IpcComm() override {
    for (int r = 0; r < size; ++r) if (opened[r]) (void)hipIpcCloseMemHandle(boxes[r]);
    if (!boxes.empty() && boxes[rank]) (void)hipFree(boxes[rank]);
    if (d_boxes) (void)hipFree(d_boxes);
    if (d_ticket) (void)hipFree(d_ticket);
    if (d_seq2) (void)hipFree(d_seq2);
    if (d_counter2) (void)hipFree(d_counter2);
    if (h_err) (void)hipHostFree(h_err);
  }
  void flush(hipStream_t stream) {
    if (!pending.nparts) return;
    ++seq;
    const int set = (int)(seq & 1);
    size_t total = 0;
    for (int q = 0; q < pending.nparts; ++q) total += pending.part[q].count;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>(64, (total + 2047) / 2048));
    k_ipc_allreduce<<<grid, 256, 0, stream>>>(pending, d_boxes, rank, size, set, slot_bytes, seq, d_ticket, timeout_ticks, h_err);
    ++n_oneshot;
    pending.nparts = 0; pending_bytes = 0;
  }
  void allreduce(void *buf, size_t count, bool is_double, hipStream_t stream) override {
    if (dead) throw CommError("IPC all-reduce: an earlier all-reduce timed out waiting for a peer; this communicator is no longer usable");
    const size_t bytes = (count * (is_double ? 8 : 4) + 15) / 16 * 16;
    if (bytes > slot_bytes || (grouping && (pending_bytes + bytes > slot_bytes || pending.nparts == IPC_MAX_PARTS))) {
      if (grouping) flush(stream);
      if (bytes > slot_bytes) {
        if (!fallback) throw std::runtime_error("IPC all-reduce: message larger than the mailbox slot and no fallback communicator");
        fallback->allreduce(buf, count, is_double, stream);
        ++n_fallback;
        return;
      }
    }
    IpcPart &pt = pending.part[pending.nparts++];
    pt.ptr = buf; pt.count = count; pt.offset = pending_bytes; pt.is_double = is_double ? 1 : 0; pt.pad = 0;
    pending_bytes += bytes;
    pending_stream = stream;
    if (!grouping) flush(stream);
  }
  void group_start() override { grouping = true; }
  void group_end() override { grouping = false; flush(pending_stream); }
  void set_timeout_ms(int ms) override { timeout_ticks = (long long)std::max(1, ms) * 100000ll; }
  bool failed() override { // a bounded spin gave up (host check after a stream synchronisation; reads pinned host memory)
    if (__atomic_load_n(h_err, __ATOMIC_ACQUIRE) != 0) dead = true;
    return dead;
  }
  int transport() const override { return 2; }
  int rccl_ranks() override { return fallback ? fallback->rccl_ranks() : 0; }
  int mailboxes_opened() const override { int n = 0; for (int r = 0; r < size; ++r) n += opened[r] ? 1 : 0; return n; }
  void message_counts(int64_t &oneshot, int64_t &fallback_) const override { oneshot = n_oneshot; fallback_ = n_fallback; }
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
  int transport() const override { return 3; }
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
