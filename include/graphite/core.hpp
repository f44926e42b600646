// graphite/core.hpp — the generic vertex / factor layer of sfu-rsl/graphite for gfx950 (MI355X).
//
// Header-only; instantiated by hipcc in the USER's translation unit, because everything here
// calls the user's traits (`parameters`, `update`, `error`, `jacobian`) from device code — the
// reference has the same constraint ("Graphite code should be called inside a .cu file",
// docs/markdown/main.md:54-55).  Names, template parameters and member functions follow
//   vertex.hpp:54-392 (VertexDescriptor), factor.hpp:120-830 (FactorDescriptor), graph.hpp:30-340
//   (Graph), solver/pcg.hpp (PCGSolver), preconditioner/*.hpp, optimizer/levenberg_marquardt.hpp,
//   loss.hpp, dual.hpp, differentiation.hpp, stream.hpp, vector.hpp (managed_vector)
// of the reference, so that a problem definition written against it (docs/markdown/main.md:89-315,
// examples/circle.cu) is source compatible apart from the matrix library it uses.
//
// Scope: this is the CALLER side of the hot path.  It is a small-graph,
// one-thread-per-factor implementation with stored Jacobians, atomics for the vertex-side sums and a
// matrix-free PCG; the BAL-specialised kernels of libgraphite_mi355x.so are what carry
// the performance claims.  Differences from the reference, all deliberate:
//   * user vertices must live in device-visible memory (managed_vector = pinned, mapped host memory;
//     the reference uses CUDA managed memory, vertex.hpp:65) and are dereferenced in place;
//   * losses are plain structs with inline member functions (no device-side virtual dispatch,
//     ops/chi2.hpp:34-44);
//   * the direct solver role (EigenLDLTSolver) is a dense assembly + the MFMA Cholesky exported by
//     libgraphite_mi355x.so (gr_dense_cholesky_solve) instead of Eigen::SimplicialLDLT on the host.
#pragma once
#include <hip/hip_runtime.h>
#include "../graphite_mi355x.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#define d_fn __device__
#define hd_fn __host__ __device__

namespace graphite {

#define GRAPHITE_HIP(expr)                                                                              \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess)                                                                               \
      throw std::runtime_error(std::string(#expr) + " -> " + hipGetErrorString(_e) + " @" + __FILE__ +  \
                               ":" + std::to_string(__LINE__));                                         \
  } while (0)

// types.hpp:8-43: 16-bit STORAGE types for the Jacobians (Graph<T, S> with S = __nv_bfloat16, bal.cu --precision *-BF16).
// The reference takes them from cuda_bf16.h; here a storage-only bfloat16 (round to nearest even) that converts
// implicitly to float, so every expression that reads a stored Jacobian computes in float or wider.
struct bfloat16 {
  uint16_t bits;
  bfloat16() = default;
  template <typename V, typename = std::enable_if_t<std::is_arithmetic<V>::value>> hd_fn bfloat16(V v) {
    const float f = (float)v;
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) bits = (uint16_t)((u >> 16) | 0x40); // NaN stays NaN
    else bits = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
  hd_fn operator float() const { const uint32_t u = (uint32_t)bits << 16; float f; memcpy(&f, &u, 4); return f; }
};
template <typename T> struct is_half_or_bfloat16 : std::false_type {};
template <> struct is_half_or_bfloat16<bfloat16> : std::true_type {};
template <typename T> using is_low_precision = is_half_or_bfloat16<T>;
template <typename T, typename S> using InvP = std::conditional_t<is_low_precision<S>::value, T, S>; // types.hpp:19-20

// block.hpp:5-17
class BlockCoordinates {
public:
  size_t row;
  size_t col;
  hd_fn bool operator==(const BlockCoordinates &other) const { return (row == other.row) && (col == other.col); }
};
using BlockDimension = BlockCoordinates;

// utils.hpp:79-103: ids handed out by add_factor stay valid until released; a released id is re-used first
template <typename H> class HandleManager {
  std::vector<H> handles;
  H last_handle;
public:
  HandleManager() : last_handle(0) {}
  H get() {
    if (handles.empty()) return last_handle++;
    H handle = handles.back();
    handles.pop_back();
    return handle;
  }
  void release(H handle) { handles.push_back(handle); }
  void clear() { handles.clear(); last_handle = 0; }
};

struct Empty {};

struct DifferentiationMode {
  struct Auto {};
  struct Manual {};
};

// ---- loss.hpp:15-51 ---------------------------------------------------------------------------
template <typename T, int E> struct DefaultLoss {
  hd_fn T loss(const T &x) const { return x; }
  hd_fn T loss_derivative(const T &) const { return T(1); }
};
template <typename T, int E> struct HuberLoss {
  T delta = T(100);
  hd_fn HuberLoss() {}
  hd_fn explicit HuberLoss(T d) : delta(d) {}
  hd_fn T loss(const T &x) const { return x <= delta * delta ? x : 2 * sqrt(x) * delta - delta * delta; }
  hd_fn T loss_derivative(const T &x) const { return x <= delta * delta ? T(1) : delta / sqrt(x); }
};

// ---- dual.hpp: forward-mode dual number, one derivative direction --------------------------------
template <typename T, typename D = T> struct Dual {
  T real;
  D dual;
  using DT = Dual<T, D>;
  hd_fn Dual() : real(0), dual(0) {}
  hd_fn Dual(T r, D d) : real(r), dual(d) {}
  hd_fn Dual(T r) : real(r), dual(0) {}
  hd_fn DT operator+(const DT &o) const { return DT(real + o.real, dual + o.dual); }
  hd_fn DT operator-(const DT &o) const { return DT(real - o.real, dual - o.dual); }
  hd_fn DT operator-() const { return DT(-real, -dual); }
  hd_fn DT operator*(const DT &o) const { return DT(real * o.real, real * o.dual + dual * o.real); }
  hd_fn DT operator/(const DT &o) const {
    if (o.real == 0) return DT(std::numeric_limits<T>::infinity(), std::numeric_limits<D>::infinity());
    const T den = o.real * o.real;
    return DT(real / o.real, (dual * o.real - real * o.dual) / den);
  }
  hd_fn DT &operator+=(const DT &o) { return *this = *this + o; }
  hd_fn DT &operator-=(const DT &o) { return *this = *this - o; }
  hd_fn DT &operator*=(const DT &o) { return *this = *this * o; }
  hd_fn DT &operator/=(const DT &o) { return *this = *this / o; }
  hd_fn bool operator<(const DT &o) const { return real < o.real; }
  hd_fn bool operator>(const DT &o) const { return real > o.real; }
  hd_fn bool operator<=(const DT &o) const { return real <= o.real; }
  hd_fn bool operator>=(const DT &o) const { return real >= o.real; }
  hd_fn bool operator==(const DT &o) const { return real == o.real; }
  hd_fn bool operator!=(const DT &o) const { return real != o.real; }
  hd_fn friend DT operator+(T a, const DT &b) { return DT(a) + b; }
  hd_fn friend DT operator-(T a, const DT &b) { return DT(a) - b; }
  hd_fn friend DT operator*(T a, const DT &b) { return DT(a) * b; }
  hd_fn friend DT operator/(T a, const DT &b) { return DT(a) / b; }
  hd_fn friend DT sin(const DT &x) { return DT(::sin(x.real), x.dual * ::cos(x.real)); }
  hd_fn friend DT cos(const DT &x) { return DT(::cos(x.real), -x.dual * ::sin(x.real)); }
  hd_fn friend DT tan(const DT &x) { const T c = ::cos(x.real); return DT(::tan(x.real), x.dual / (c * c)); }
  hd_fn friend DT exp(const DT &x) { const T e = ::exp(x.real); return DT(e, x.dual * e); }
  hd_fn friend DT log(const DT &x) { return DT(::log(x.real), x.dual / x.real); }
  hd_fn friend DT sqrt(const DT &x) { const T s = ::sqrt(x.real); return DT(s, s == 0 ? D(0) : x.dual / (2 * s)); }
  hd_fn friend DT abs(const DT &x) { return x.real < 0 ? -x : x; }
  hd_fn friend DT atan2(const DT &y, const DT &x) {
    const T den = x.real * x.real + y.real * y.real;
    return DT(::atan2(y.real, x.real), (x.real * y.dual - y.real * x.dual) / den);
  }
  hd_fn friend DT acos(const DT &x) { return DT(::acos(x.real), -x.dual / ::sqrt(1 - x.real * x.real)); }
  hd_fn friend DT asin(const DT &x) { return DT(::asin(x.real), x.dual / ::sqrt(1 - x.real * x.real)); }
};

// ---- vector.hpp:25-60 — device-visible storage with the std::vector surface the examples use -----
// Pinned, mapped host memory: one virtual address on host and device.  Like the reference's
// managed_vector it is neither copyable nor movable; growth re-allocates (reserve first when
// addresses must stay fixed, circle.cu:106).
template <typename T> class managed_vector {
  T *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
  void grow(size_t cap) {
    if (cap <= cap_) return;
    T *q = nullptr;
    GRAPHITE_HIP(hipHostMalloc(reinterpret_cast<void **>(&q), std::max<size_t>(cap, 1) * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent));
    for (size_t i = 0; i < n_; ++i) new (q + i) T(p_[i]);
    if (p_) (void)hipHostFree(p_);
    p_ = q; cap_ = cap;
  }
public:
  struct pointer { T *p; T *get() const { return p; } };
  managed_vector() = default;
  explicit managed_vector(size_t n) { resize(n); }
  managed_vector(size_t n, const T &v) { resize(n, v); }
  managed_vector(const managed_vector &) = delete;
  managed_vector &operator=(const managed_vector &) = delete;
  ~managed_vector() { if (p_) (void)hipHostFree(p_); }
  void reserve(size_t cap) { grow(cap); }
  void resize(size_t n, const T &v = T()) {
    if (n > cap_) grow(std::max(n, 2 * cap_));
    for (size_t i = n_; i < n; ++i) new (p_ + i) T(v);
    n_ = n;
  }
  void push_back(const T &v) {
    if (n_ == cap_) grow(std::max<size_t>(2 * cap_, 8));
    new (p_ + n_++) T(v);
  }
  void pop_back() { if (n_) --n_; } // (a no-op on an empty vector, tests/vector.cu:31-35)
  void clear() { n_ = 0; }
  size_t size() const { return n_; }
  size_t capacity() const { return cap_; }
  bool empty() const { return n_ == 0; }
  T &operator[](size_t i) { return p_[i]; }
  const T &operator[](size_t i) const { return p_[i]; }
  T &back() { return p_[n_ - 1]; }
  pointer data() const { return pointer{p_}; } // .data().get(), as with thrust::universal_vector
  T *raw() const { return p_; }
  T *begin() const { return p_; }
  T *end() const { return p_ + n_; }
};

// Storage of the library's OWN arrays (Jacobians, residuals, gradient, solver vectors, dense blocks): HBM, where the
// reference keeps thrust::device_vector (factor.hpp:158-174, graph.hpp:40-60).  Kernels stream device memory at HBM
// speed (measured 6.7 TB/s against 0.09 TB/s for the pinned host memory of managed_vector, which every access crosses
// PCIe for).  Host access is by copy, as with thrust: operator[] returns a proxy, to_host() / assign() move whole
// arrays.  Plain-data element types only.
template <typename T> class hbm_vector {
  static_assert(std::is_trivially_copyable<T>::value, "hbm_vector holds plain data");
  T *p_ = nullptr;
  size_t n_ = 0, cap_ = 0, hw_ = 0; // hw_: elements ever exposed (beyond it the allocation is still zero)
  void grow(size_t cap) {
    if (cap <= cap_) return;
    T *q = nullptr;
    cap = std::max<size_t>(cap, 1);
    GRAPHITE_HIP(hipMalloc(reinterpret_cast<void **>(&q), cap * sizeof(T)));
    if (n_) GRAPHITE_HIP(hipMemcpy(q, p_, n_ * sizeof(T), hipMemcpyDeviceToDevice));
    GRAPHITE_HIP(hipMemset(q + n_, 0, (cap - n_) * sizeof(T))); // value-initialised tail: growing by one element costs nothing later
    if (p_) (void)hipFree(p_);
    p_ = q; cap_ = cap; hw_ = n_;
  }
public:
  struct pointer { T *p; T *get() const { return p; } };
  hbm_vector() = default;
  explicit hbm_vector(size_t n) { resize(n); }
  hbm_vector(size_t n, const T &v) { resize(n, v); }
  hbm_vector(const hbm_vector &) = delete;
  hbm_vector &operator=(const hbm_vector &) = delete;
  ~hbm_vector() { if (p_) (void)hipFree(p_); }
  void reserve(size_t cap) { grow(cap); }
  void resize(size_t n) { // new elements are value-initialised (zero bytes)
    if (n > cap_) grow(std::max(n, 2 * cap_));
    if (n > n_ && n_ < hw_) GRAPHITE_HIP(hipMemset(p_ + n_, 0, (std::min(n, hw_) - n_) * sizeof(T))); // re-exposed after a shrink
    n_ = n; hw_ = std::max(hw_, n);
  }
  void resize(size_t n, const T &v); // new elements = v (filled on the device)
  void resize_uninit(size_t n) { if (n > cap_) grow(n); n_ = n; hw_ = std::max(hw_, n); }
  void clear() { n_ = 0; }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  // host element access as with thrust::device_vector (a copy per access: for tests and occasional scalars)
  struct reference {
    T *p;
    operator T() const { T v; GRAPHITE_HIP(hipMemcpy(&v, p, sizeof(T), hipMemcpyDeviceToHost)); return v; }
    reference &operator=(const T &v) { GRAPHITE_HIP(hipMemcpy(p, &v, sizeof(T), hipMemcpyHostToDevice)); return *this; }
  };
  reference operator[](size_t i) { return reference{p_ + i}; }
  T operator[](size_t i) const { T v; GRAPHITE_HIP(hipMemcpy(&v, p_ + i, sizeof(T), hipMemcpyDeviceToHost)); return v; }
  pointer data() const { return pointer{p_}; }
  T *raw() const { return p_; }
  T *begin() const { return p_; }
  T *end() const { return p_ + n_; }
  std::vector<T> to_host() const {
    std::vector<T> h(n_);
    if (n_) GRAPHITE_HIP(hipMemcpy(h.data(), p_, n_ * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
  void assign(const T *src, size_t n) { // from host (or any) memory
    if (n > cap_) { n_ = 0; grow(n); }
    n_ = n; hw_ = std::max(hw_, n); // the assigned elements have been exposed: a later shrink + resize must zero them again
    if (n) GRAPHITE_HIP(hipMemcpy(p_, src, n * sizeof(T), hipMemcpyDefault));
  }
};

// ---- stream.hpp:7-24 ---------------------------------------------------------------------------
class StreamPool {
  std::vector<hipStream_t> s_;
public:
  explicit StreamPool(size_t n) : s_(std::max<size_t>(n, 1)) { for (auto &s : s_) GRAPHITE_HIP(hipStreamCreate(&s)); }
  StreamPool(const StreamPool &) = delete;
  ~StreamPool() { for (auto s : s_) (void)hipStreamDestroy(s); }
  hipStream_t &select(size_t i) { return s_[i % s_.size()]; }
  void sync_all() { for (auto s : s_) (void)hipStreamSynchronize(s); }
  void sync_n(size_t n) { for (size_t i = 0; i < std::min(n, s_.size()); ++i) (void)hipStreamSynchronize(s_[i]); }
};

namespace detail {
constexpr int TPB = 256;
inline int blocks(size_t n) { return (int)((n + TPB - 1) / TPB); }
inline void sync() { GRAPHITE_HIP(hipDeviceSynchronize()); }
// Host loops over per-factor tables (680 k factors on Ladybug-1723) on a few threads: fn(begin, end) over contiguous chunks;
// an exception of any chunk is rethrown in the caller.  Small ranges stay on the calling thread.
template <typename F> inline void parallel_chunks(size_t n, size_t min_chunk, F &&fn) {
  const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
  const size_t nt = std::max<size_t>(1, std::min<size_t>(hw, n / std::max<size_t>(1, min_chunk)));
  if (nt <= 1) { fn((size_t)0, n, (size_t)0); return; }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err(nt);
  const size_t per = (n + nt - 1) / nt;
  for (size_t k = 0; k < nt; ++k)
    th.emplace_back([&, k] {
      try { fn(std::min(n, k * per), std::min(n, (k + 1) * per), k); } catch (...) { err[k] = std::current_exception(); }
    });
  for (auto &t : th) t.join();
  for (auto &e : err) if (e) std::rethrow_exception(e);
}
// order-sensitive 64-bit digest of a byte range (four independent multiply-add lanes over 8-byte words, then the tail):
// EngineCache's guard against writes to a descriptor's public arrays that no API call announced
inline uint64_t digest_serial(uint64_t seed, const void *data, size_t bytes);
// large ranges: the digests of 1 MB-aligned chunks (computed side by side), digested in chunk order
inline uint64_t digest(uint64_t seed, const void *data, size_t bytes) {
  constexpr size_t CH = (size_t)1 << 20;
  if (bytes < 4 * CH) return digest_serial(seed, data, bytes);
  const size_t nch = (bytes + CH - 1) / CH;
  std::vector<uint64_t> part(nch);
  const unsigned char *p = static_cast<const unsigned char *>(data);
  parallel_chunks(nch, 2, [&](size_t b, size_t e, size_t) {
    for (size_t c = b; c < e; ++c) part[c] = digest_serial(seed + c, p + c * CH, std::min(CH, bytes - c * CH));
  });
  return digest_serial(seed ^ (uint64_t)bytes, part.data(), part.size() * sizeof(uint64_t));
}
inline uint64_t digest_serial(uint64_t seed, const void *data, size_t bytes) {
  const unsigned char *p = static_cast<const unsigned char *>(data);
  uint64_t a = seed, b = seed ^ 0xC2B2AE3D27D4EB4Full, c = seed + 0x165667B19E3779F9ull, d = ~seed;
  size_t i = 0;
  for (; i + 32 <= bytes; i += 32) {
    uint64_t w[4];
    std::memcpy(w, p + i, 32);
    a = a * 0x9E3779B97F4A7C15ull + w[0]; b = b * 0xC2B2AE3D27D4EB4Full + w[1];
    c = c * 0x165667B19E3779F9ull + w[2]; d = d * 0x27D4EB2F165667C5ull + w[3];
  }
  uint64_t t = 0;
  for (; i < bytes; ++i) t = t * 131 + p[i];
  uint64_t h = a ^ (b << 1 | b >> 63) ^ (c << 7 | c >> 57) ^ (d << 13 | d >> 51) ^ (t * 0x9E3779B97F4A7C15ull) ^ (uint64_t)bytes;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  return h;
}

// active.hpp:11-21
hd_fn inline bool is_factor_active(uint8_t v, uint8_t level) { return (v & 0x7F) <= level && (v & 0x80) == 0; }
hd_fn inline bool is_vertex_active(const uint8_t *state, size_t id) { return state[id] == 0; }

template <typename T> __global__ void k_fill(T *p, size_t n, T v) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
template <typename T> inline void fill(T *p, size_t n, T v) { if (n) k_fill<T><<<blocks(n), TPB>>>(p, n, v); }
} // namespace detail
template <typename T> void hbm_vector<T>::resize(size_t n, const T &v) {
  if (n > cap_) grow(std::max(n, 2 * cap_));
  if (n > n_) { detail::fill(p_ + n_, n - n_, v); GRAPHITE_HIP(hipDeviceSynchronize()); }
  n_ = n; hw_ = std::max(hw_, n);
}
namespace detail {

// detection of the optional State / get_state / set_state of vertex traits (main.md:100-111)
template <typename Tr, typename = void> struct state_of { using type = typename Tr::Vertex; static constexpr bool custom = false; };
template <typename Tr> struct state_of<Tr, std::void_t<typename Tr::State>> { using type = typename Tr::State; static constexpr bool custom = true; };

template <typename Fn, typename... A> struct invocable {
  template <typename F, typename = decltype(std::declval<F>()(std::declval<A>()...))> static std::true_type test(int);
  template <typename> static std::false_type test(...);
  static constexpr bool value = decltype(test<Fn>(0))::value;
};
} // namespace detail

// =================================================================================================
// VertexDescriptor (vertex.hpp:28-392)
// =================================================================================================
// block.hpp:19-30 — hashable, as the reference's tests build their own block -> offset maps
} // namespace graphite
namespace std {
template <> struct hash<graphite::BlockCoordinates> {
  size_t operator()(const graphite::BlockCoordinates &b) const noexcept { return std::hash<size_t>()(b.row) * 0x9E3779B97F4A7C15ull ^ std::hash<size_t>()(b.col); }
};
} // namespace std
namespace graphite {

namespace detail { struct PoseEngineOptions; struct PoseEngineResult; namespace pe { struct FactorInfo; struct VertexInfo; template <typename T> struct SolveArgs; template <typename T> struct FactorArgs; } } // engine_pose.hpp
template <typename T, typename S> class BaseVertexDescriptor {
public:
  virtual ~BaseVertexDescriptor() = default;
  virtual size_t dimension() const = 0;
  virtual size_t count() const = 0;
  virtual bool is_fixed(size_t id) const = 0;
  virtual bool is_active(size_t id) const = 0;
  virtual bool exists(size_t id) const = 0;
  virtual size_t get_local_id(size_t id) const = 0;
  virtual uint8_t *get_active_state() const = 0;
  virtual size_t *get_hessian_ids() const = 0;
  virtual const size_t *get_block_ids() const = 0;                               // block column of each vertex (vertex.hpp:47)
  virtual const std::unordered_map<size_t, size_t> &get_global_map() const = 0; // global id -> local id (vertex.hpp:37)
  // what kernels read: HBM copies while an optimiser loop has the vertices mirrored (begin_mirror/end_mirror), else the above
  virtual const uint8_t *device_active_state() const = 0;
  virtual const size_t *device_hessian_ids() const = 0;
  virtual void begin_mirror() = 0;
  virtual void end_mirror() = 0;
  virtual const std::vector<size_t> &local_to_global() const = 0;
  virtual void apply_update(const T *delta_x, const T *scales) = 0;
  virtual void backup_parameters() = 0;
  virtual void restore_parameters() = 0;
  // parameter blocks of all vertices in local order, [count x dimension] (engine hand-over, see solve.hpp)
  virtual void gather_parameters(T *out) = 0;
  virtual void scatter_parameters(const T *in) = 0;
  bool eliminate = false; // set_eliminate (vertex.hpp:98): kept for API parity, the PCG path ignores it
  // The pose-graph engine (engine_pose.hpp), the pieces that see THIS descriptor's traits: what the driver needs to know (false: the vertex
  // type is not plain data or its tangent dimension is above 7), the resident workgroups per CU of its solve kernel, the solve launch
  // (every PCG iteration and the trial step through Traits::update) and the launch that takes a rejected last step back
  virtual bool pose_engine_vertex(detail::pe::VertexInfo &) { return false; }
  virtual int pose_engine_occupancy(bool /*one_pass*/) { return 0; }
  virtual void pose_engine_solve(const detail::pe::SolveArgs<T> &, int /*grid*/, bool /*cooperative*/, bool /*one_pass*/) {}
  virtual void pose_engine_finish(const void * /*Ctl*/, const int * /*k2l*/, int /*NV*/) {}
  std::shared_ptr<void> pose_engine_state; // the engine's buffers and structure cache for graphs on this descriptor, kept between optimiser calls
  void set_eliminate(bool e) { eliminate = e; ++structure_epoch; }
  // bumped by every call that changes the vertex set, a vertex's address or its fixed flag: what a cached engine problem
  // (solve.hpp, EngineCache) is keyed on, next to the factor descriptor's epoch and content fingerprint
  size_t structure_epoch = 0;
  // digest of the vertex addresses and state bytes (public arrays, writable without an API call)
  virtual uint64_t content_fingerprint() const { return 0; }
};

namespace detail {
template <typename T, typename Tr>
__global__ void k_vertex_update(typename Tr::Vertex **x, const uint8_t *state, const size_t *hid, size_t n,
                                const T *dx, const T *scales) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n || !is_vertex_active(state, i)) return;
  T d[Tr::dimension];
  for (size_t k = 0; k < Tr::dimension; ++k) d[k] = dx[hid[i] + k] * (scales ? scales[hid[i] + k] : T(1)); // ops/update.hpp:26
  Tr::update(*x[i], d);
}
template <typename T, typename Tr>
__global__ void k_vertex_gather(typename Tr::Vertex **x, size_t n, T *out) {
  const size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (v >= n) return;
  T p[Tr::dimension];
  Tr::parameters(*x[v], p);
  for (size_t k = 0; k < Tr::dimension; ++k) out[v * Tr::dimension + k] = p[k];
}
// moves every vertex onto the given parameters through Traits::update (the only write access the traits offer)
template <typename T, typename Tr>
__global__ void k_vertex_scatter(typename Tr::Vertex **x, size_t n, const T *in) {
  const size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (v >= n) return;
  T p[Tr::dimension], d[Tr::dimension];
  Tr::parameters(*x[v], p);
  for (size_t k = 0; k < Tr::dimension; ++k) d[k] = in[v * Tr::dimension + k] - p[k];
  Tr::update(*x[v], d);
}
template <typename Tr, typename St>
__global__ void k_vertex_backup(typename Tr::Vertex **x, const uint8_t *state, size_t n, St *bak) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n || !is_vertex_active(state, i)) return;
  if constexpr (state_of<Tr>::custom) bak[i] = Tr::get_state(*x[i]);
  else bak[i] = *x[i];
}
template <typename Tr, typename St>
__global__ void k_vertex_restore(typename Tr::Vertex **x, const uint8_t *state, size_t n, const St *bak) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n || !is_vertex_active(state, i)) return;
  if constexpr (state_of<Tr>::custom) Tr::set_state(*x[i], bak[i]);
  else *x[i] = bak[i];
}
template <typename V> __global__ void k_mirror_in(V *const *user, size_t n, V *mirror, V **ptrs) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) { mirror[i] = *user[i]; ptrs[i] = mirror + i; }
}
template <typename V> __global__ void k_mirror_out(V *const *user, size_t n, const V *mirror) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) *user[i] = mirror[i];
}
} // namespace detail

namespace detail {
// ops/hessian.hpp:80-112 augment_hessian_diagonal_kernel: the diagonal of every ACTIVE vertex's block <- d + mu clamp(d, 1e-6, 1e32)
// (or d + mu), d from the scalar diagonal, in double; blocks indexed by LOCAL vertex id, everything else untouched
template <typename P, int D> __global__ void k_augment_block_diagonal(P *blocks, const P *scalar_diagonal, P mu, int identity, const uint8_t *state, size_t count) {
  const size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (v >= count || !is_vertex_active(state, v)) return;
  P *block = blocks + v * D * D;
  for (int i = 0; i < D; ++i) {
    const double d = (double)scalar_diagonal[v * D + i];
    const double cl = d < 1.0e-6 ? 1.0e-6 : (d > 1.0e32 ? 1.0e32 : d);
    block[i * D + i] = (P)(identity ? d + (double)mu : d + (double)mu * cl);
  }
}
// ops/hessian.hpp:127-152 apply_block_jacobi_kernel: z[col .. col + d) = block * r[col .. col + d) per active vertex
// (column-major block indexed by local vertex id, `hid` = the vertex's scalar column)
template <typename T, typename P> __global__ void k_block_apply(const P *inv, const size_t *hid, const uint8_t *state, size_t count, int d, T *z, const T *r) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= count * d) return;
  const size_t v = t / d, row = t % d;
  if (!is_vertex_active(state, v)) return;
  T s = 0;
  for (int c = 0; c < d; ++c) s += (T)inv[v * d * d + row + c * d] * r[hid[v] + c];
  z[hid[v] + row] = s;
}
} // namespace detail

template <typename T, typename S, typename VTraits> class VertexDescriptor : public BaseVertexDescriptor<T, S> {
public:
  using Traits = VTraits;
  using VertexType = typename Traits::Vertex;
  using StateType = typename detail::state_of<Traits>::type;
  static constexpr size_t dim = Traits::dimension;

  managed_vector<VertexType *> x_device;  // pointers into user memory (vertex.hpp:65)
  managed_vector<uint8_t> active_state;   // bit0 fixed, bit7 not used by an active factor
  managed_vector<size_t> hessian_ids;     // scalar column of each vertex
  managed_vector<size_t> block_ids;       // block column of each vertex (vertex.hpp:74)
  managed_vector<StateType> backup_state;
  std::unordered_map<size_t, size_t> global_to_local_map;
  std::vector<size_t> local_to_global_map;
  // get_local_id is called once per factor slot by every initialize_optimization (1.36 M times on Ladybug-1723: ~40 ms of
  // hash look-ups).  Ids that span a range of at most a few times their number (the usual 0 .. n - 1 or offset blocks) get a
  // plain table [id - base] -> local id, rebuilt lazily after any change of the vertex set.
  mutable std::vector<uint32_t> dense_local;
  mutable size_t dense_base = 0;
  mutable bool dense_dirty = true, dense_usable = false;
  void build_dense_ids() const {
    dense_dirty = false; dense_usable = false; dense_local.clear();
    const size_t n = local_to_global_map.size();
    if (!n || n >= 0xffffffffu) return;
    size_t lo = local_to_global_map[0], hi = lo;
    for (size_t g : local_to_global_map) { lo = std::min(lo, g); hi = std::max(hi, g); }
    if (hi - lo >= 4 * n + 1024) return; // sparse ids: stay with the hash map
    dense_base = lo;
    dense_local.assign(hi - lo + 1, 0xffffffffu);
    for (size_t l = 0; l < n; ++l) dense_local[local_to_global_map[l] - lo] = (uint32_t)l;
    dense_usable = true;
  }

  void reserve(size_t n) { x_device.reserve(n); active_state.reserve(n); hessian_ids.reserve(n); block_ids.reserve(n); backup_state.reserve(n); local_to_global_map.reserve(n); global_to_local_map.reserve(n); }
  void add_vertex(size_t id, VertexType *vertex, bool fixed = false) { // vertex.hpp:241-256
    dense_dirty = true; ++this->structure_epoch;
    global_to_local_map[id] = x_device.size();
    local_to_global_map.push_back(id);
    x_device.push_back(vertex);
    active_state.push_back(static_cast<uint8_t>(fixed));
    hessian_ids.push_back(0); block_ids.push_back(0);
    backup_state.resize(x_device.size());
  }
  void remove_vertex(size_t id) { // swap with last, vertex.hpp:185-215
    auto it = global_to_local_map.find(id);
    if (it == global_to_local_map.end()) { std::cerr << "Vertex with id " << id << " not found." << std::endl; return; }
    dense_dirty = true; ++this->structure_epoch;
    const size_t l = it->second, last = x_device.size() - 1;
    x_device[l] = x_device[last]; active_state[l] = active_state[last]; hessian_ids[l] = hessian_ids[last]; block_ids[l] = block_ids[last];
    const size_t moved = local_to_global_map[last];
    local_to_global_map[l] = moved; global_to_local_map[moved] = l;
    global_to_local_map.erase(id);
    x_device.pop_back(); active_state.pop_back(); hessian_ids.pop_back(); block_ids.pop_back(); local_to_global_map.pop_back();
    backup_state.resize(x_device.size());
  }
  void replace_vertex(size_t id, VertexType *vertex) {
    auto it = global_to_local_map.find(id);
    if (it == global_to_local_map.end()) { std::cerr << "Vertex with id " << id << " not found." << std::endl; return; }
    x_device[it->second] = vertex; ++this->structure_epoch;
  }
  void set_fixed(size_t id, bool fixed) { active_state[global_to_local_map.at(id)] = static_cast<uint8_t>(fixed); ++this->structure_epoch; }
  void set_hessian_column(size_t id, size_t column, size_t block) { const size_t l = global_to_local_map.at(id); hessian_ids[l] = column; block_ids[l] = block; } // vertex.hpp:288-296
  const size_t *get_block_ids() const override { return block_ids.raw(); }
  const std::unordered_map<size_t, size_t> &get_global_map() const override { return global_to_local_map; }
  // The reference's per-descriptor block-Jacobi steps (vertex.hpp:97-110, ops/hessian.hpp:80-167), as BlockJacobiPreconditioner
  // drives them there.  Here the preconditioner damps and inverts in one launch (solve.hpp k_block_inverse) and applies through
  // the same k_block_apply; the two members stay for callers — and the reference's tests/vertex.cu:121-226 — that use them directly.
  template <typename P> void augment_block_diagonal_async(P *block_diagonal, P *scalar_diagonal, const T mu, const bool use_identity, hipStream_t stream) {
    if (count()) detail::k_augment_block_diagonal<P, (int)dim><<<detail::blocks(count()), detail::TPB, 0, stream>>>(block_diagonal, scalar_diagonal, (P)mu, use_identity ? 1 : 0, get_active_state(), count());
  }
  template <typename P> void apply_block_jacobi(T *z, const T *r, P *block_diagonal, hipStream_t stream) {
    if (count()) detail::k_block_apply<T, P><<<detail::blocks(count() * dim), detail::TPB, 0, stream>>>(block_diagonal, get_hessian_ids(), get_active_state(), count(), (int)dim, z, r);
  }
  bool is_fixed(size_t id) const override { return (active_state[global_to_local_map.at(id)] & 0x1) > 0; }
  bool is_active(size_t id) const override { return detail::is_vertex_active(active_state.raw(), global_to_local_map.at(id)); }
  bool exists(size_t id) const override { return global_to_local_map.count(id) > 0; }
  VertexType *get_vertex(size_t id) { return x_device[global_to_local_map.at(id)]; }
  size_t get_local_id(size_t id) const override {
    if (dense_dirty) build_dense_ids();
    if (dense_usable) {
      const size_t k = id - dense_base; // wraps for id < base: caught by the range test
      if (k < dense_local.size() && dense_local[k] != 0xffffffffu) return dense_local[k];
    }
    return global_to_local_map.at(id); // throws for an unknown id, as before
  }
  size_t dimension() const override { return dim; }
  size_t count() const override { return x_device.size(); }
  // Device mirror (the reference keeps user vertices in CUDA unified memory, which migrates to the GPU on first touch,
  // vertex.hpp:65; pinned host memory does not): while an optimiser loop runs, the vertex VALUES, their active state and
  // Hessian columns and the backup states live in HBM and every kernel reads those; end_mirror() writes the values back
  // into the user's objects.  Outside a loop kernels dereference the user's memory directly, as before.
  static constexpr bool can_mirror = std::is_trivially_copyable<VertexType>::value && std::is_trivially_copyable<StateType>::value;
  struct Empty1 {};
  std::conditional_t<can_mirror, hbm_vector<VertexType>, Empty1> mirror;
  std::conditional_t<can_mirror, hbm_vector<StateType>, Empty1> mirror_backup;
  hbm_vector<VertexType *> mirror_ptrs;
  hbm_vector<uint8_t> mirror_state;
  hbm_vector<size_t> mirror_hid;
  bool mirrored = false;
  void begin_mirror() override {
    if constexpr (can_mirror) {
      if (mirrored || !count()) return;
      mirror.resize_uninit(count()); mirror_backup.resize_uninit(count()); mirror_ptrs.resize_uninit(count());
      detail::k_mirror_in<VertexType><<<detail::blocks(count()), detail::TPB>>>(x_device.raw(), count(), mirror.raw(), mirror_ptrs.raw());
      mirror_state.assign(active_state.raw(), count());
      mirror_hid.assign(hessian_ids.raw(), count());
      GRAPHITE_HIP(hipDeviceSynchronize());
      mirrored = true;
    }
  }
  void end_mirror() override {
    if constexpr (can_mirror) {
      if (!mirrored) return;
      detail::k_mirror_out<VertexType><<<detail::blocks(count()), detail::TPB>>>(x_device.raw(), count(), mirror.raw());
      GRAPHITE_HIP(hipDeviceSynchronize());
      mirrored = false;
    }
  }
  VertexType **vertices() const { return mirrored ? mirror_ptrs.raw() : x_device.raw(); }
  StateType *backup_ptr() {
    if constexpr (can_mirror) { if (mirrored) return mirror_backup.raw(); }
    return backup_state.raw();
  }
  uint8_t *get_active_state() const override { return active_state.raw(); }
  size_t *get_hessian_ids() const override { return hessian_ids.raw(); }
  const uint8_t *device_active_state() const override { return mirrored ? mirror_state.raw() : active_state.raw(); }
  const size_t *device_hessian_ids() const override { return mirrored ? mirror_hid.raw() : hessian_ids.raw(); }
  const std::vector<size_t> &local_to_global() const override { return local_to_global_map; }
  void to_device() {}
  void clear() { x_device.clear(); active_state.clear(); hessian_ids.clear(); block_ids.clear(); backup_state.clear(); global_to_local_map.clear(); local_to_global_map.clear(); dense_dirty = true; ++this->structure_epoch; }

  bool pose_engine_vertex(detail::pe::VertexInfo &vi) override; // engine_pose.hpp
  int pose_engine_occupancy(bool one_pass) override;
  void pose_engine_solve(const detail::pe::SolveArgs<T> &sa, int G, bool cooperative, bool one_pass) override;
  void pose_engine_finish(const void *ctl, const int *k2l, int NV) override;
  void apply_update(const T *delta_x, const T *scales) override {
    if (count()) detail::k_vertex_update<T, Traits><<<detail::blocks(count()), detail::TPB>>>(vertices(), device_active_state(), device_hessian_ids(), count(), delta_x, scales);
  }
  void backup_parameters() override {
    if (count()) detail::k_vertex_backup<Traits, StateType><<<detail::blocks(count()), detail::TPB>>>(vertices(), device_active_state(), count(), backup_ptr());
  }
  void restore_parameters() override {
    if (count()) detail::k_vertex_restore<Traits, StateType><<<detail::blocks(count()), detail::TPB>>>(vertices(), device_active_state(), count(), backup_ptr());
  }
  uint64_t content_fingerprint() const override {
    uint64_t h = detail::digest(0x51ED270B9F3C8A11ull ^ (uint64_t)count(), x_device.raw(), x_device.size() * sizeof(VertexType *));
    h = detail::digest(h, active_state.raw(), active_state.size());
    return detail::digest(h, local_to_global_map.data(), local_to_global_map.size() * sizeof(size_t));
  }
  void gather_parameters(T *out) override {
    if (count()) detail::k_vertex_gather<T, Traits><<<detail::blocks(count()), detail::TPB>>>(vertices(), count(), out);
  }
  void scatter_parameters(const T *in) override {
    if (count()) detail::k_vertex_scatter<T, Traits><<<detail::blocks(count()), detail::TPB>>>(vertices(), count(), in);
  }
};

// =================================================================================================
// FactorDescriptor (factor.hpp:120-830)
// =================================================================================================
namespace detail { class EngineModelBase; } // engine_model.hpp
// (pose-graph engine types: declared above BaseVertexDescriptor)
template <typename T, typename S> class BaseFactorDescriptor {
public:
  virtual ~BaseFactorDescriptor() = default;
  // The pose-graph engine (engine_pose.hpp), the pieces that see THIS descriptor's traits: what the driver needs to know (false: more than two
  // slots or an error dimension above 7), the local vertex ids of its active factors ([active][2], -1 for a unary factor's missing slot) and
  // the launch of the factor kernel (error, chi2, Jacobians, product records, LM decision by the last workgroup of the graph's launches)
  virtual bool pose_engine_factor(detail::pe::FactorInfo &, bool /*check precision-matrix symmetry*/) { return false; }
  virtual void pose_engine_ids(int * /*lij*/) {}
  virtual void pose_engine_launch(const detail::pe::FactorArgs<T> &, void * /*vertex mirror*/, int /*mode*/) {}
  // The engine's per-observation kernels instantiated on THIS descriptor's traits (engine_model.hpp), for the active factors:
  // local pose / landmark ids per active factor (active_indices order) in cam / pt.  nullptr: not a (<= 9, <= 3) -> <= 2 binary
  // factor of plain-data types, or a precision matrix that is not symmetric positive semi-definite.
  virtual std::shared_ptr<detail::EngineModelBase> make_engine_model(std::vector<int32_t> & /*cam*/, std::vector<int32_t> & /*pt*/, size_t /*num_poses*/, size_t /*num_landmarks*/) { return nullptr; }
  virtual size_t internal_count() const = 0;
  virtual size_t active_count() const = 0;
  // active list, local ids, vertex use flags; `light`: without what only the generic kernels need (Jacobian storage, the
  // per-vertex factor lists of the gathers) — the probe / export of the engine hand-over (solve.hpp) runs on a light one
  virtual void initialize(uint8_t level, bool light = false) = 0;
  virtual void flag_active_vertices() = 0;
  virtual void compute_error() = 0;
  virtual void compute_jacobians() = 0;
  virtual void compute_chi2() = 0;
  virtual T chi2() = 0;
  virtual void scalar_diagonal(T *diag) = 0;      // ops/hessian.hpp:419-474
  virtual void scale_jacobians(const T *scales) = 0;
  virtual void compute_b(T *b) = 0;               // ops/linearize.hpp:240-303
  virtual void compute_Jv(T *res, const T *x) = 0;   // ops/product.hpp:195
  virtual void compute_Jtv(T *out, const T *res) = 0; // ops/product.hpp:405
  virtual T *work_residual() = 0;                 // [internal_count * E] scratch for J v
  virtual bool compute_Jv_fresh(T * /*res*/, const T * /*x*/) { return false; } // res = J x for the active factors, written not added; false: not available
  virtual size_t error_dimension() const = 0;
  virtual void block_diagonal(size_t slot, T *blocks) = 0; // ops/hessian.hpp:171-270
  virtual size_t num_slots() const = 0;
  virtual BaseVertexDescriptor<T, S> *slot_descriptor(size_t slot) const = 0;
  virtual void dense_hessian(T *H, size_t n) = 0; // upper + lower, for the direct solver
  // block-sparse Hessian (sparse.hpp, hessian.hpp:45-330): upper block coordinates touched by the active factors
  // (scalar_to_block: Hessian column -> block column), where each (factor, vertex pair) block lives, accumulation
  // The symbolic phase runs on the device (the reference walks an unordered_map on the host, hessian.hpp:257-288): every
  // descriptor writes one 64-bit key (col << 32 | row, ~0 = no block) per active factor and vertex pair i < k, the Hessian
  // sorts / uniques them, and each descriptor then finds its blocks by binary search in the sorted key list.
  virtual size_t block_key_count() = 0;                                                       // active factors x N (N - 1) / 2
  virtual void emit_block_keys(uint64_t *keys, const size_t *scalar_to_block) = 0;            // device pointers
  virtual void sparse_setup(const uint64_t *keys, size_t num_keys, const size_t *value_offsets, const size_t *scalar_to_block) = 0;
  virtual void sparse_hessian(S *values) = 0;
  // A factor descriptor whose traits declare `static constexpr bool bal_reprojection_model = true` (camera 9 =
  // [angle-axis, t, f, k1, k2], point 3, pixel residual of examples/reprojection_error.cuh) can hand its active
  // factors to the specialised BAL engine: local camera / point ids, 2 observation scalars per factor, loss.
  // false = not that model, or something the engine does not represent (precision matrices, mixed losses...).
  // Only the ACTIVE factors are exported (active_indices, factor.hpp:419-465 / active.hpp:18-21): a deactivated outlier or a
  // level mask is an index list at export, not a reason to leave the engine.
  virtual bool export_bal(std::vector<int32_t> &, std::vector<int32_t> &, std::vector<T> &, int &, double &) { return false; }
  // bumped by every call that changes the factor set or a factor's level (add / remove / set_active / clear)
  size_t structure_epoch = 0;
  // 64-bit digest of everything an engine problem is built from and that is reachable WITHOUT such a call (the reference
  // exposes these arrays as public members, tests/factor.cu:154,177,318,796): vertex ids, observations, activity bytes,
  // precision matrices, losses.  One pass over pinned host memory, ~2 ms for 680 k factors.
  virtual uint64_t content_fingerprint() const { return 0; }
  // The evidence for that hand-over (solve.hpp compares it with gr_bal_model_evaluate): for up to max_samples active factors,
  // evenly spread, the USER's functions evaluated on the device — Traits::parameters of both vertices (9 + 3), the
  // observation (2), Traits::error (2), the two Jacobian blocks (Traits::jacobian<T, I>, or dual numbers for Auto
  // factors; 18 + 6, E x d column-major) — and update_dev, the largest relative deviation of Traits::update from
  // plain addition on the parameters.  false: not a (9, 3) -> 2 factor with a two-component observation.
  // Behind the sampled factors come SYNTHETIC triples that steer the user's functions into the branches a sample of a
  // well-conditioned graph never visits (copies of a sampled factor's vertices moved there through the user's own update(),
  // which the same probe verifies to be plain addition): rotation vector exactly 0 (the theta == 0 branch,
  // projection_jacobians.cuh:175-212: identity rotation, ZERO rotation block), the point on the other side of the camera
  // (P_z of the other sign), a large radial term (k1 r^2, k2 r^4 of order one), a quarter-turn rotation (theta > 0.5: closed
  // forms instead of series).  *num_synthetic (optional) receives how many of the returned triples are synthetic.
  virtual bool probe_bal(size_t /*max_samples*/, std::vector<T> & /*cam*/, std::vector<T> & /*pt*/, std::vector<T> & /*obs*/, std::vector<T> & /*res*/,
                         std::vector<T> & /*Jc*/, std::vector<T> & /*Jp*/, double & /*update_dev*/, size_t * /*num_synthetic*/ = nullptr) { return false; }
  virtual bool declares_bal_model() const { return false; } // the optional tag: a mismatch is then reported, not silent
};

namespace detail {
// Everything a per-factor kernel needs, by value.
template <typename F> struct FactorView {
  using T = typename F::Scalar;
  using S = typename F::Storage;
  static constexpr size_t N = F::N, E = F::E;
  const size_t *active_ids;
  size_t n_active;
  const size_t *ids; // [nf][N] local vertex ids
  const typename F::ObservationType *obs;
  const typename F::ConstraintDataType *data;
  const typename F::LossType *loss;
  const S *pmat; // [nf][E*E]
  T *residuals;  // [nf][E]
  T *chi2;       // [nf]
  S *dchi2;      // [nf]
  std::array<S *, N> jac;
  bool dynamic;    // set_jacobian_storage(false): no stored blocks, every consumer recomputes them
  const T *scales; // column scales applied to recomputed blocks (nullptr = unscaled)
  std::array<void *, N> verts; // Vertex** of each slot
  std::array<const uint8_t *, N> vstate;
  std::array<const size_t *, N> hid;
};

template <typename Tr, typename = void> struct has_bal_tag : std::false_type {};
template <typename Tr> struct has_bal_tag<Tr, std::enable_if_t<Tr::bal_reprojection_model>> : std::true_type {};
template <typename O> __host__ __device__ auto obs_component(const O &o, int i, int) -> decltype((double)o(i)) { return (double)o(i); }
template <typename O> __host__ __device__ auto obs_component(const O &o, int i, long) -> decltype((double)o[i]) { return (double)o[i]; }
template <typename O, typename = void> struct obs_indexable : std::false_type {};
template <typename O> struct obs_indexable<O, std::void_t<decltype(obs_component(std::declval<const O &>(), 0, 0))>> : std::true_type {};
template <typename F, size_t I> using slot_traits = typename std::tuple_element<I, typename F::Traits::VertexDescriptors>::type::Traits;
template <typename F, size_t I> using slot_vertex = typename slot_traits<F, I>::Vertex;
template <typename F, size_t I> constexpr size_t slot_dim() { return slot_traits<F, I>::dimension; }

// call Traits::error in whichever of the documented argument orders it was written (main.md:283-291):
// (vertices..., params..., [obs], [data], err) | (params..., [obs], [data], err) | (vertices..., [obs], [data], err)
template <typename Fn, typename Tup> struct applicable;
template <typename Fn, typename... A> struct applicable<Fn, std::tuple<A...>> : std::integral_constant<bool, invocable<Fn, A...>::value> {};
template <typename Fn, typename Tup, size_t... Ks> __device__ inline void apply_tuple(Fn &fn, Tup &&t, std::index_sequence<Ks...>) { fn(std::get<Ks>(t)...); }
template <typename Fn, typename Tup> __device__ inline void apply_tuple(Fn &fn, Tup &&t) {
  apply_tuple(fn, t, std::make_index_sequence<std::tuple_size<typename std::decay<Tup>::type>::value>{});
}

template <typename F, typename D, typename VT, typename PT, size_t... Is>
__device__ inline void call_error(const VT &v, PT &p, const typename F::ObservationType &obs,
                                  const typename F::ConstraintDataType &data, D *err, std::index_sequence<Is...>) {
  using Tr = typename F::Traits;
  constexpr bool has_obs = !std::is_empty<typename F::ObservationType>::value, has_dat = !std::is_empty<typename F::ConstraintDataType>::value;
  auto fn = [](auto &&...a) -> decltype(Tr::template error<D>(std::forward<decltype(a)>(a)...)) { return Tr::template error<D>(std::forward<decltype(a)>(a)...); };
  using Fn = decltype(fn);
  auto verts = std::forward_as_tuple(*std::get<Is>(v)...);
  auto params = std::make_tuple(static_cast<const D *>(std::get<Is>(p))...);
  auto tail = [&] {
    if constexpr (has_obs && has_dat) return std::tuple<const typename F::ObservationType &, const typename F::ConstraintDataType &, D *>(obs, data, err);
    else if constexpr (has_obs) return std::tuple<const typename F::ObservationType &, D *>(obs, err);
    else if constexpr (has_dat) return std::tuple<const typename F::ConstraintDataType &, D *>(data, err);
    else return std::tuple<D *>(err);
  }();
  auto both = std::tuple_cat(verts, params, tail);
  auto ponly = std::tuple_cat(params, tail);
  auto vonly = std::tuple_cat(verts, tail);
  if constexpr (applicable<Fn, decltype(both)>::value) apply_tuple(fn, both);
  else if constexpr (applicable<Fn, decltype(ponly)>::value) apply_tuple(fn, ponly);
  else {
    static_assert(applicable<Fn, decltype(vonly)>::value, "Traits::error<D> matches none of the documented argument orders");
    apply_tuple(fn, vonly);
  }
}

// Traits::jacobian<T, I>(vertices..., [obs], [data], T *jac)  (main.md:294-315).  As ops/linearize.hpp:43-79 does it:
// the user function is always instantiated in the GRAPH precision T; with S != T the block is evaluated into a T
// buffer and converted on the way into the storage.
template <typename F, size_t I, typename Jt, typename VT, size_t... Is>
__device__ inline void call_jacobian_t(const VT &v, const typename F::ObservationType &obs, const typename F::ConstraintDataType &data,
                                       Jt *jac, std::index_sequence<Is...>) {
  using Tr = typename F::Traits;
  constexpr bool has_obs = !std::is_empty<typename F::ObservationType>::value, has_dat = !std::is_empty<typename F::ConstraintDataType>::value;
  if constexpr (has_obs && has_dat) Tr::template jacobian<Jt, I>(*std::get<Is>(v)..., obs, data, jac);
  else if constexpr (has_obs) Tr::template jacobian<Jt, I>(*std::get<Is>(v)..., obs, jac);
  else if constexpr (has_dat) Tr::template jacobian<Jt, I>(*std::get<Is>(v)..., data, jac);
  else Tr::template jacobian<Jt, I>(*std::get<Is>(v)..., jac);
}
template <typename F, size_t I, typename VT, size_t... Is>
__device__ inline void call_jacobian(const VT &v, const typename F::ObservationType &obs, const typename F::ConstraintDataType &data,
                                     typename F::Storage *jac, std::index_sequence<Is...> seq) {
  using T = typename F::Scalar;
  using Sj = typename F::Storage;
  if constexpr (std::is_same<T, Sj>::value) call_jacobian_t<F, I, T>(v, obs, data, jac, seq);
  else {
    constexpr size_t sz = F::E * slot_dim<F, I>();
    T tmp[sz];
    for (size_t i = 0; i < sz; ++i) tmp[i] = T(0);
    call_jacobian_t<F, I, T>(v, obs, data, tmp, seq);
    for (size_t i = 0; i < sz; ++i) jac[i] = (Sj)tmp[i];
  }
}

template <typename F, typename D, size_t... Is>
__device__ inline auto gather_vertices(const FactorView<F> &fv, size_t f, std::index_sequence<Is...>) {
  return std::make_tuple((reinterpret_cast<slot_vertex<F, Is> **>(fv.verts[Is])[fv.ids[f * F::N + Is]])...);
}

// residual of every active factor (ops/error.hpp:253-323)
template <typename F, size_t... Is>
__global__ void k_error(FactorView<F> fv, std::index_sequence<Is...> seq) {
  using T = typename F::Scalar;
  const size_t a = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (a >= fv.n_active) return;
  const size_t f = fv.active_ids[a];
  auto v = gather_vertices<F, T>(fv, f, seq);
  std::tuple<T[slot_dim<F, Is>()]...> p;
  ((slot_traits<F, Is>::parameters(*std::get<Is>(v), (T *)std::get<Is>(p))), ...);
  T err[F::E];
  call_error<F, T>(v, p, fv.obs[f], fv.data[f], err, seq);
  for (size_t i = 0; i < F::E; ++i) fv.residuals[f * F::E + i] = err[i];
}

// Jacobian block of slot I: analytic (one thread per factor) or dual numbers (one thread per column)
template <typename F, size_t I, size_t... Is>
__global__ void k_jacobian(FactorView<F> fv, std::index_sequence<Is...> seq) {
  using T = typename F::Scalar;
  using Sj = typename F::Storage;
  constexpr size_t d = slot_dim<F, I>();
  constexpr bool manual = std::is_same<typename F::Traits::Differentiation, DifferentiationMode::Manual>::value;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if constexpr (manual) {
    if (t >= fv.n_active) return;
    const size_t f = fv.active_ids[t];
    if (!is_vertex_active(fv.vstate[I], fv.ids[f * F::N + I])) return;
    auto v = gather_vertices<F, T>(fv, f, seq);
    call_jacobian<F, I>(v, fv.obs[f], fv.data[f], fv.jac[I] + f * F::E * d, seq);
  } else {
    if (t >= fv.n_active * d) return;
    const size_t f = fv.active_ids[t / d], col = t % d;
    if (!is_vertex_active(fv.vstate[I], fv.ids[f * F::N + I])) return;
    using D = Dual<T, T>;
    auto v = gather_vertices<F, T>(fv, f, seq);
    std::tuple<D[slot_dim<F, Is>()]...> p;
    ((slot_traits<F, Is>::parameters(*std::get<Is>(v), (D *)std::get<Is>(p))), ...);
    std::get<I>(p)[col].dual = T(1);
    D err[F::E];
    call_error<F, D>(v, p, fv.obs[f], fv.data[f], err, seq);
    for (size_t i = 0; i < F::E; ++i) fv.jac[I][f * F::E * d + col * F::E + i] = (Sj)err[i].dual;
  }
}

// Engine hand-over probe (BaseFactorDescriptor::probe_bal): the user's parameters / error / Jacobian blocks / update on
// sampled factors of a (9, 3) -> 2 descriptor.  out: [ns][PROBE_W] = cam 9 | pt 3 | obs 2 | r 2 | Jc 18 | Jp 6 | update deviation 1
constexpr size_t PROBE_W = 41;
template <typename Tr, typename T> __device__ inline T probe_update_deviation(const typename Tr::Vertex &vtx, const T *p) {
  using V = typename Tr::Vertex;
  if constexpr (std::is_copy_constructible<V>::value) {
    constexpr size_t d = Tr::dimension;
    V copy(vtx);
    T delta[d], q[d];
    for (size_t k = 0; k < d; ++k) delta[k] = T(0.0009765625) * T(k + 1) * (p[k] < T(0) ? T(-1) : T(1)); // exact binary steps
    Tr::update(copy, delta);
    Tr::parameters(copy, q);
    T dev = 0;
    for (size_t k = 0; k < d; ++k) {
      const T want = p[k] + delta[k];
      const T e = fabs((double)(q[k] - want)) / fmax(1.0, fabs((double)want));
      dev = e > dev || e != e ? e : dev;
    }
    return dev;
  } else return std::numeric_limits<T>::infinity();
}
template <typename F, typename VT, size_t... Is>
__device__ inline void probe_evaluate(const FactorView<F> &fv, size_t f, VT &v, typename F::Scalar *o, std::index_sequence<Is...> seq) {
  using T = typename F::Scalar;
  std::tuple<T[slot_dim<F, Is>()]...> p;
  ((slot_traits<F, Is>::parameters(*std::get<Is>(v), (T *)std::get<Is>(p))), ...);
  for (int k = 0; k < 9; ++k) o[k] = std::get<0>(p)[k];
  for (int k = 0; k < 3; ++k) o[9 + k] = std::get<1>(p)[k];
  o[12] = (T)obs_component(fv.obs[f], 0, 0); o[13] = (T)obs_component(fv.obs[f], 1, 0);
  T err[2];
  call_error<F, T>(v, p, fv.obs[f], fv.data[f], err, seq);
  o[14] = err[0]; o[15] = err[1];
  T *jc = o + 16, *jp = o + 34;
  for (int k = 0; k < 24; ++k) jc[k] = T(0); // the reference zero-fills the storage before the user function writes (ops/linearize.hpp:127)
  if constexpr (std::is_same<typename F::Traits::Differentiation, DifferentiationMode::Manual>::value) {
    call_jacobian_t<F, 0, T>(v, fv.obs[f], fv.data[f], jc, seq);
    call_jacobian_t<F, 1, T>(v, fv.obs[f], fv.data[f], jp, seq);
  } else {
    using D = Dual<T, T>;
    for (int col = 0; col < 12; ++col) {
      std::tuple<D[slot_dim<F, Is>()]...> pd;
      ((slot_traits<F, Is>::parameters(*std::get<Is>(v), (D *)std::get<Is>(pd))), ...);
      if (col < 9) std::get<0>(pd)[col].dual = T(1); else std::get<1>(pd)[col - 9].dual = T(1);
      D e[2];
      call_error<F, D>(v, pd, fv.obs[f], fv.data[f], e, seq);
      jc[2 * col] = e[0].dual; jc[2 * col + 1] = e[1].dual; // jp follows jc: columns 9..11 land in the point block
    }
  }
  const T d0 = probe_update_deviation<slot_traits<F, 0>, T>(*std::get<0>(v), std::get<0>(p));
  const T d1 = probe_update_deviation<slot_traits<F, 1>, T>(*std::get<1>(v), std::get<1>(p));
  o[40] = d0 > d1 || d0 != d0 ? d0 : d1;
}
// variant 0: the factor as it is; 1: r = 0; 2: t_z -> t_z - 2 P_z' with P_z' ~ the camera-frame depth (the point lands on the
// other side of the camera); 3: k1 = 5, k2 = 20 and the point pulled off the optical axis (radial terms of order one);
// 4: r = (1.1, -0.7, 0.9) (theta = 1.6: closed forms).  Variants > 0 move COPIES of the vertices with the user's update().
constexpr int PROBE_VARIANTS = 5;
template <typename Tr, typename T> __device__ inline void probe_move(typename Tr::Vertex &vtx, const T *target) {
  constexpr size_t d = Tr::dimension;
  T p[d], delta[d];
  Tr::parameters(vtx, p);
  for (size_t k = 0; k < d; ++k) delta[k] = target[k] - p[k];
  Tr::update(vtx, delta);
}
template <typename F, size_t... Is>
__global__ void k_probe_bal(FactorView<F> fv, size_t stride, size_t ns, size_t nsyn, typename F::Scalar *out, std::index_sequence<Is...> seq) {
  using T = typename F::Scalar;
  using V0 = typename slot_traits<F, 0>::Vertex;
  using V1 = typename slot_traits<F, 1>::Vertex;
  const size_t a = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (a >= ns + nsyn) return;
  const int variant = a < ns ? 0 : 1 + (int)((a - ns) % (PROBE_VARIANTS - 1));
  const size_t src = a < ns ? a : ((a - ns) / (PROBE_VARIANTS - 1)) % ns;
  const size_t f = fv.active_ids[src * stride];
  auto v = gather_vertices<F, T>(fv, f, seq);
  T *o = out + a * PROBE_W;
  if constexpr (std::is_copy_constructible<V0>::value && std::is_copy_constructible<V1>::value) {
    V0 cam_copy(*std::get<0>(v));
    V1 pt_copy(*std::get<1>(v));
    if (variant > 0) {
      T c[9], x[3];
      slot_traits<F, 0>::parameters(cam_copy, c);
      slot_traits<F, 1>::parameters(pt_copy, x);
      if (variant == 1) { c[0] = c[1] = c[2] = T(0); }
      else if (variant == 4) { c[0] = T(1.1); c[1] = T(-0.7); c[2] = T(0.9); }
      else if (variant == 2) {
        // depth along the optical axis with the small-angle rotation R ~ I + [r]x: enough to flip its sign
        const T pz = x[2] + c[0] * x[1] - c[1] * x[0] + c[5];
        c[5] -= T(2) * pz + (pz < T(0) ? T(-1) : T(1));
      } else { c[7] = T(5); c[8] = T(20); x[0] += T(1.5); x[1] -= T(1.25); }
      probe_move<slot_traits<F, 0>, T>(cam_copy, c);
      probe_move<slot_traits<F, 1>, T>(pt_copy, x);
      std::get<0>(v) = &cam_copy;
      std::get<1>(v) = &pt_copy;
    }
    probe_evaluate<F>(fv, f, v, o, seq);
  } else {
    if (variant > 0) { for (size_t k = 0; k < PROBE_W; ++k) o[k] = T(0); o[40] = std::numeric_limits<T>::infinity(); return; }
    probe_evaluate<F>(fv, f, v, o, seq);
  }
}

// Jacobian block of slot I of factor f: the stored (already scaled) block, or with set_jacobian_storage(false)
// (factor.hpp:626-640; the *_dynamic kernels of ops/linearize.hpp:308, ops/hessian.hpp:272,478, ops/product.hpp:103,292)
// the analytic block recomputed into `buf` and scaled the way the stored one would have been.
template <typename F, size_t I, size_t... Is>
__device__ inline const typename F::Storage *jac_block(const FactorView<F> &fv, size_t f, typename F::Storage *buf, std::index_sequence<Is...> seq) {
  using T = typename F::Scalar;
  using Sj = typename F::Storage;
  constexpr size_t d = slot_dim<F, I>(), E = F::E;
  if constexpr (std::is_same<typename F::Traits::Differentiation, DifferentiationMode::Manual>::value) {
    if (fv.dynamic) {
      auto v = gather_vertices<F, T>(fv, f, seq);
      for (size_t k = 0; k < E * d; ++k) buf[k] = Sj(0);
      call_jacobian<F, I>(v, fv.obs[f], fv.data[f], buf, seq);
      if (fv.scales) {
        const size_t col0 = fv.hid[I][fv.ids[f * F::N + I]];
        for (size_t c = 0; c < d; ++c)
          for (size_t i = 0; i < E; ++i) buf[c * E + i] = (Sj)((T)buf[c * E + i] * fv.scales[col0 + c]);
      }
      return buf;
    }
  }
  return fv.jac[I] + f * E * d;
}

// chi2 = rho(r^T P r), dchi2 = rho'  (ops/chi2.hpp:10-44)
template <typename F> __global__ void k_chi2(FactorView<F> fv) {
  using T = typename F::Scalar;
  const size_t a = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (a >= fv.n_active) return;
  const size_t f = fv.active_ids[a];
  T value = 0;
  for (size_t i = 0; i < F::E; ++i) {
    T r2 = 0;
    for (size_t j = 0; j < F::E; ++j) r2 += (T)fv.pmat[f * F::E * F::E + i * F::E + j] * fv.residuals[f * F::E + j];
    value += r2 * fv.residuals[f * F::E + i];
  }
  fv.chi2[f] = fv.loss[f].loss(value);
  fv.dchi2[f] = (typename F::Storage)fv.loss[f].loss_derivative(value);
}

// sum of v over the active factors in two stages with a fixed partition (block b sums its contiguous share, tree in LDS;
// the partials are added in block order): the same bits every run.  partial: SUM_BLOCKS elements.
constexpr int SUM_BLOCKS = 256;
template <typename T> __global__ void k_sum_active(const T *v, const size_t *active, size_t n, T *partial) {
  __shared__ T red[TPB];
  const size_t per = (n + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  T s = 0;
  for (size_t i = i0 + threadIdx.x; i < i1; i += TPB) s += v[active[i]];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = TPB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
template <typename T> __global__ void k_sum_partials(const T *partial, int nb, T *out) { // one wave, fixed order
  T s = 0;
  for (int b = threadIdx.x; b < nb; b += 64) s += partial[b];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) *out = s;
}

// J_c^T P J_c' of two columns of (possibly different) slots
template <typename F> __device__ inline typename F::Scalar jtpj(const FactorView<F> &fv, size_t f, const typename F::Storage *Ja, const typename F::Storage *Jb) {
  using T = typename F::Scalar;
  T value = 0;
  for (size_t i = 0; i < F::E; ++i) {
    T pj = 0;
    for (size_t j = 0; j < F::E; ++j) pj += (T)fv.pmat[f * F::E * F::E + i * F::E + j] * (T)Jb[j];
    value += (T)Ja[i] * pj;
  }
  return value;
}

// which: 0 scalar diagonal, 1 scale Jacobians, 2 b -= J^T rho' P r, 3 J^T v, 4 block diagonal
template <typename F, size_t I, int WHICH>
__global__ void k_slot(FactorView<F> fv, typename F::Scalar *out, const typename F::Scalar *in) {
  using T = typename F::Scalar;
  constexpr size_t d = slot_dim<F, I>(), E = F::E;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= fv.n_active * d) return;
  const size_t f = fv.active_ids[t / d], c = t % d;
  const size_t v = fv.ids[f * F::N + I];
  if (!is_vertex_active(fv.vstate[I], v)) return;
  typename F::Storage buf[E * d];
  const auto *Jb = jac_block<F, I>(fv, f, buf, std::make_index_sequence<F::N>{});
  const auto *J = Jb + c * E;
  const size_t col = fv.hid[I][v] + c;
  if constexpr (WHICH == 0) {
    atomicAdd(&out[col], jtpj(fv, f, J, J) * (T)fv.dchi2[f]);
  } else if constexpr (WHICH == 1) { // stored blocks only
    auto *Jw = fv.jac[I] + f * E * d + c * E;
    for (size_t i = 0; i < E; ++i) Jw[i] = (typename F::Storage)((T)Jw[i] * in[col]);
  } else if constexpr (WHICH == 2) {
    T s = 0;
    for (size_t i = 0; i < E; ++i) {
      T pr = 0;
      for (size_t j = 0; j < E; ++j) pr += (T)fv.pmat[f * E * E + i * E + j] * fv.residuals[f * E + j];
      s += (T)J[i] * pr;
    }
    atomicAdd(&out[col], -s * (T)fv.dchi2[f]);
  } else if constexpr (WHICH == 3) { // out[col] += J[:,c]^T (rho' P in_f)
    T s = 0;
    for (size_t i = 0; i < E; ++i) {
      T pr = 0;
      for (size_t j = 0; j < E; ++j) pr += (T)fv.pmat[f * E * E + i * E + j] * in[f * E + j];
      s += (T)J[i] * pr;
    }
    atomicAdd(&out[col], s * (T)fv.dchi2[f]);
  } else { // per-vertex d x d block, column-major: block(row, c) += rho' J_row^T P J_c
    for (size_t row = 0; row < d; ++row) atomicAdd(&out[v * d * d + row + c * d], jtpj(fv, f, Jb + row * E, J) * (T)fv.dchi2[f]);
  }
}

// res_f += J_I x_v  (ops/product.hpp:195-300), one thread per (factor, residual row)
template <typename F, size_t I> __global__ void k_Jv(FactorView<F> fv, typename F::Scalar *res, const typename F::Scalar *x) {
  using T = typename F::Scalar;
  constexpr size_t d = slot_dim<F, I>(), E = F::E;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= fv.n_active * E) return;
  const size_t f = fv.active_ids[t / E], i = t % E;
  const size_t v = fv.ids[f * F::N + I];
  if (!is_vertex_active(fv.vstate[I], v)) return;
  typename F::Storage buf[E * d];
  const auto *Jb = jac_block<F, I>(fv, f, buf, std::make_index_sequence<F::N>{});
  T s = 0;
  for (size_t c = 0; c < d; ++c) s += (T)Jb[c * E + i] * x[fv.hid[I][v] + c];
  res[f * E + i] += s; // slots are launched one after the other on one stream: no race
}

// ---- vertex-centric accumulation: the CDNA4 form of the vertex-side sums ---------------------------------------------------
// The reference adds every factor's contribution to its vertices with one atomicAdd per output scalar (ops/hessian.hpp:171-270,
// 419-474, ops/linearize.hpp:240-303, ops/product.hpp:228-292): on a bundle-adjustment graph a camera's 81-entry block takes
// 81 x (its hundreds of observations) atomics to the same addresses, and the sums come out in arrival order.  Here the
// active factors of every slot are grouped by vertex once per initialize() (vptr / vfac, ascending factor order) and a
// group of W lanes (a power of two <= 64 chosen from the slot's mean degree) walks ONE vertex's list: lane j takes factors
// j, j + W, ..., keeps its partial sums in registers, a fixed butterfly over the W lanes follows, lane 0 adds the total to the
// output.  No atomics, no zero-filled scratch, the same bits every run; a factor's stored Jacobian block is read once per
// gather.  WHICH as in k_slot: 0 scalar diagonal, 2 b -= J^T rho' P r, 3 out += J^T (rho' P in), 4 block diagonal (one group
// per (vertex, row)).
constexpr size_t GATHER_ROWS = 3;
constexpr size_t GATHER_SYM_MAX_DIM = 9; // 45 accumulators
// every E x E precision matrix symmetric?  (information matrices are; the single-pass block diagonal of k_gather relies on it and
// the three-pass form is kept for descriptors where some matrix is not)
template <typename S> __global__ void k_pmat_asymmetric(const S *pmat, size_t nf, size_t E, int *flag) {
  const size_t f = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (f >= nf) return;
  for (size_t i = 0; i < E; ++i)
    for (size_t j = i + 1; j < E; ++j)
      if (!((double)pmat[f * E * E + i * E + j] == (double)pmat[f * E * E + j * E + i])) *flag = 1;
}
template <typename F, size_t I, int WHICH, bool SYM = false>
__global__ void k_gather(FactorView<F> fv, const size_t *__restrict__ vptr, const size_t *__restrict__ vfac, size_t nv, int W,
                         typename F::Scalar *__restrict__ out, const typename F::Scalar *__restrict__ in) {
  using T = typename F::Scalar;
  constexpr size_t d = slot_dim<F, I>(), E = F::E;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t g = t / (size_t)W;
  const int j = (int)(t % (size_t)W);
  // block diagonal: a group takes GATHER_ROWS rows of its vertex's d x d block, so a factor's Jacobian block is read
  // ceil(d / GATHER_ROWS) times instead of d times — or, SYM (every precision matrix symmetric, checked at initialize(); d <= 9):
  // the whole lower triangle in ONE pass, d (d + 1) / 2 accumulators, the block read once (Ladybug-1723 camera blocks: the three
  // passes moved 294 MB of stored Jacobians for 98 MB of data)
  constexpr size_t RB = WHICH == 4 ? (SYM ? d : (d < GATHER_ROWS ? d : GATHER_ROWS)) : 1, NRB = (d + RB - 1) / RB,
                   NACC = WHICH == 4 ? (SYM ? d * (d + 1) / 2 : RB * d) : d;
  const size_t v = WHICH == 4 ? g / NRB : g, row0 = WHICH == 4 ? (g % NRB) * RB : 0;
  const bool on = v < nv && is_vertex_active(fv.vstate[I], v);
  T acc[NACC];
  for (size_t c = 0; c < NACC; ++c) acc[c] = T(0);
  if (on) {
    for (size_t k = vptr[v] + (size_t)j; k < vptr[v + 1]; k += (size_t)W) {
      const size_t f = vfac[k];
      typename F::Storage buf[E * d];
      const auto *Jb = jac_block<F, I>(fv, f, buf, std::make_index_sequence<F::N>{});
      const T w = (T)fv.dchi2[f];
      if constexpr (WHICH == 0) {
        for (size_t c = 0; c < d; ++c) acc[c] += jtpj(fv, f, Jb + c * E, Jb + c * E) * w;
      } else if constexpr (WHICH == 4 && SYM) {
        for (size_t r = 0; r < d; ++r)
          for (size_t c = 0; c <= r; ++c) acc[r * (r + 1) / 2 + c] += jtpj(fv, f, Jb + r * E, Jb + c * E) * w;
      } else if constexpr (WHICH == 4) {
        for (size_t rr = 0; rr < RB; ++rr)
          if (row0 + rr < d)
            for (size_t c = 0; c < d; ++c) acc[rr * d + c] += jtpj(fv, f, Jb + (row0 + rr) * E, Jb + c * E) * w;
      } else {
        const T *vec = WHICH == 2 ? fv.residuals + f * E : in + f * E;
        T pr[E];
        for (size_t i = 0; i < E; ++i) {
          T q = 0;
          for (size_t jj = 0; jj < E; ++jj) q += (T)fv.pmat[f * E * E + i * E + jj] * vec[jj];
          pr[i] = q;
        }
        for (size_t c = 0; c < d; ++c) {
          T q = 0;
          for (size_t i = 0; i < E; ++i) q += (T)Jb[c * E + i] * pr[i];
          acc[c] += q * w;
        }
      }
    }
  }
  for (int o = 1; o < W && o < 64; o <<= 1)
    for (size_t c = 0; c < NACC; ++c) acc[c] += __shfl_xor(acc[c], o, 64);
  if (W > 64) {
    // a whole WORKGROUP per vertex (W = TPB = 256, high-degree vertices: a camera of a bundle-adjustment graph sees hundreds of
    // factors, and one wave per camera left most of the chip idle): the four waves' sums meet in LDS and are added in wave order
    __shared__ T red[TPB / 64][NACC];
    if ((threadIdx.x & 63) == 0)
      for (size_t c = 0; c < NACC; ++c) red[threadIdx.x >> 6][c] = acc[c];
    __syncthreads();
    if (threadIdx.x == 0)
      for (size_t c = 0; c < NACC; ++c) { T s = red[0][c]; for (int wv = 1; wv < TPB / 64; ++wv) s += red[wv][c]; acc[c] = s; }
  }
  if (!on || j != 0) return;
  if constexpr (WHICH == 4 && SYM) {
    for (size_t r = 0; r < d; ++r)
      for (size_t c = 0; c <= r; ++c) {
        const T a = acc[r * (r + 1) / 2 + c];
        out[v * d * d + r + c * d] += a;
        if (c != r) out[v * d * d + c + r * d] += a;
      }
  } else if constexpr (WHICH == 4) {
    for (size_t rr = 0; rr < RB; ++rr)
      if (row0 + rr < d)
        for (size_t c = 0; c < d; ++c) out[v * d * d + (row0 + rr) + c * d] += acc[rr * d + c];
  } else {
    const size_t col = fv.hid[I][v];
    for (size_t c = 0; c < d; ++c) out[col + c] += (WHICH == 2 ? -acc[c] : acc[c]);
  }
}
// res_f (+)= sum over the slots of J_s x_vs in ONE launch.  ACCUM = true is compute_Jv's public meaning (the reference adds into
// y and leaves it alone where every vertex is fixed / unused, tests/factor.cu:597-756); ACCUM = false writes the rows, for the
// matrix-free product of the PCG solver (no zero fill of the scratch before it)
template <typename F, bool ACCUM, size_t... Is> __global__ void k_Jv_all(FactorView<F> fv, typename F::Scalar *res, const typename F::Scalar *x, std::index_sequence<Is...> seq) {
  using T = typename F::Scalar;
  constexpr size_t E = F::E;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= fv.n_active * E) return;
  const size_t f = fv.active_ids[t / E], i = t % E;
  T s = 0;
  auto one = [&](auto slot) {
    constexpr size_t I = decltype(slot)::value;
    constexpr size_t d = slot_dim<F, I>();
    const size_t v = fv.ids[f * F::N + I];
    if (!is_vertex_active(fv.vstate[I], v)) return;
    typename F::Storage buf[E * d];
    const auto *Jb = jac_block<F, I>(fv, f, buf, seq);
    const size_t col = fv.hid[I][v];
    for (size_t c = 0; c < d; ++c) s += (T)Jb[c * E + i] * x[col + c];
  };
  (one(std::integral_constant<size_t, Is>{}), ...);
  if (ACCUM) res[f * E + i] += s; else res[f * E + i] = s;
}

// dense H(col_a, col_b) += rho' J_a^T P J_b for every pair of slots (direct solver only)
template <typename F, size_t I, size_t K> __global__ void k_dense_pair(FactorView<F> fv, typename F::Scalar *H, size_t n) {
  using T = typename F::Scalar;
  constexpr size_t di = slot_dim<F, I>(), dk = slot_dim<F, K>(), E = F::E;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= fv.n_active * di * dk) return;
  const size_t f = fv.active_ids[t / (di * dk)], r = (t / dk) % di, c = t % dk;
  const size_t vi = fv.ids[f * F::N + I], vk = fv.ids[f * F::N + K];
  if (!is_vertex_active(fv.vstate[I], vi) || !is_vertex_active(fv.vstate[K], vk)) return;
  typename F::Storage bi[E * di], bk[E * dk];
  const auto *Ji = jac_block<F, I>(fv, f, bi, std::make_index_sequence<F::N>{});
  const auto *Jk = jac_block<F, K>(fv, f, bk, std::make_index_sequence<F::N>{});
  const T val = jtpj(fv, f, Ji + r * E, Jk + c * E) * (T)fv.dchi2[f];
  atomicAdd(&H[(fv.hid[I][vi] + r) * n + fv.hid[K][vk] + c], val);
}

// block-sparse H: the (I, K) vertex pair of every active factor adds rho' J_I^T P J_K into its block (column-major,
// rows = the vertex with the lower block index; `dst` = value offset, top bit set when that vertex is slot K)
template <typename F, size_t I, size_t K> __global__ void k_sparse_pair(FactorView<F> fv, typename F::Storage *values, const size_t *dst, size_t pair_index, size_t num_pairs) {
  using T = typename F::Scalar;
  constexpr size_t di = slot_dim<F, I>(), dk = slot_dim<F, K>(), E = F::E;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= fv.n_active * di * dk) return;
  const size_t a = t / (di * dk), f = fv.active_ids[a], r = (t / dk) % di, c = t % dk;
  const size_t d = dst[a * num_pairs + pair_index];
  if (d == ~size_t(0)) return; // a fixed / inactive vertex: no block
  const bool transposed = (d >> 63) != 0;
  const size_t off = d & ~(size_t(1) << 63);
  typename F::Storage bi[E * di], bk[E * dk];
  const auto *Ji = jac_block<F, I>(fv, f, bi, std::make_index_sequence<F::N>{});
  const auto *Jk = jac_block<F, K>(fv, f, bk, std::make_index_sequence<F::N>{});
  const T val = jtpj(fv, f, Ji + r * E, Jk + c * E) * (T)fv.dchi2[f];
  const size_t idx = transposed ? (c + dk * r) : (r + di * c);
  atomicAdd(&values[off + idx], (typename F::Storage)val);
}

hd_fn inline uint64_t pack_block_key(size_t row, size_t col) { return ((uint64_t)col << 32) | (uint64_t)row; }
// position of `key` in the sorted unique key list (it is there by construction)
__device__ inline size_t find_block_key(const uint64_t *keys, size_t n, uint64_t key) {
  size_t lo = 0, hi = n;
  while (lo < hi) { const size_t mid = (lo + hi) / 2; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
// block column of every slot of factor f (npos: that vertex has no column — fixed / unused)
template <typename F> __device__ inline void slot_blocks(const FactorView<F> &fv, size_t f, const size_t *s2b, size_t (&blk)[F::N]) {
  for (size_t i = 0; i < F::N; ++i) {
    const size_t v = fv.ids[f * F::N + i];
    blk[i] = is_vertex_active(fv.vstate[i], v) ? s2b[fv.hid[i][v]] : ~size_t(0);
  }
}
template <typename F> __global__ void k_block_keys(FactorView<F> fv, const size_t *s2b, uint64_t *keys) {
  const size_t a = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (a >= fv.n_active) return;
  size_t blk[F::N];
  slot_blocks<F>(fv, fv.active_ids[a], s2b, blk);
  size_t p = a * (F::N * (F::N - 1) / 2);
  for (size_t i = 0; i < F::N; ++i)
    for (size_t k = i + 1; k < F::N; ++k, ++p) {
      const bool none = blk[i] == ~size_t(0) || blk[k] == ~size_t(0) || blk[i] == blk[k];
      keys[p] = none ? ~uint64_t(0) : pack_block_key(blk[i] < blk[k] ? blk[i] : blk[k], blk[i] < blk[k] ? blk[k] : blk[i]);
    }
}
// per active factor and vertex pair i <= k (row-major upper enumeration): value offset of its block, top bit set when the
// block's rows belong to slot k (the transpose is stored); npos when the pair has no block
// ... and, per slot and vertex, the value offset of the vertex's DIAGONAL block (vdiag: every factor of a vertex writes the same
// number), for the gathered diagonal accumulation (k_diag_store)
template <size_t N> struct VertexDiag { size_t *p[N]; };
template <typename F> __global__ void k_sparse_dst(FactorView<F> fv, const size_t *s2b, const uint64_t *keys, size_t nkeys, const size_t *offsets, size_t *dst, VertexDiag<F::N> vdiag) {
  const size_t a = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (a >= fv.n_active) return;
  size_t blk[F::N];
  const size_t f = fv.active_ids[a];
  slot_blocks<F>(fv, f, s2b, blk);
  size_t p = a * (F::N * (F::N + 1) / 2);
  for (size_t i = 0; i < F::N; ++i)
    for (size_t k = i; k < F::N; ++k, ++p) {
      size_t d = ~size_t(0);
      if (blk[i] != ~size_t(0) && blk[k] != ~size_t(0) && (i == k || blk[i] != blk[k])) {
        const bool swap = blk[k] < blk[i];
        d = offsets[find_block_key(keys, nkeys, pack_block_key(swap ? blk[k] : blk[i], swap ? blk[i] : blk[k]))];
        if (swap) d |= size_t(1) << 63;
        if (i == k) vdiag.p[i][fv.ids[f * F::N + i]] = d;
      }
      dst[p] = d;
    }
}
// values[diagonal block of vertex v] += the gathered d x d block of v (k_gather<.., 4>: column-major, one writer per entry)
template <typename T, typename S> __global__ void k_diag_store(size_t nv, size_t dd, const T *blocks, const size_t *vdiag, S *values) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= nv * dd) return;
  const size_t off = vdiag[t / dd];
  if (off != ~size_t(0)) values[off + t % dd] += (S)blocks[t];
}

// ---- small dense inverses, one thread per block ------------------------------------------------------------------------------
// The two d x d work matrices of a thread live in LDS laid out [entry][thread] (a wave's accesses to one entry are consecutive
// words) instead of dynamically indexed per-thread arrays, which the compiler can only keep in scratch memory (4 KB per thread
// for d <= 16: the 1 723 camera blocks of a bundle-adjustment S took 320 us that way).  BLOCK_INV_THREADS threads per
// workgroup: 2 d^2 doubles each, d <= 16 -> at most 128 KB of the CU's 160 KB.
constexpr int BLOCK_INV_THREADS = 32;
inline size_t block_inverse_lds_bytes(size_t d) { return 2 * d * d * BLOCK_INV_THREADS * sizeof(double); }
// d >= 12 needs more than the default 64 KB of dynamic LDS.  The attribute is per DEVICE, so it is set before every launch (it
// costs a table write), not once per process; launch_check() after the <<<>>> turns a refused launch into an exception instead
// of stale inverses.
inline void allow_block_inverse_lds(const void *kernel) {
  GRAPHITE_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)block_inverse_lds_bytes(16)));
}
inline void launch_check() { GRAPHITE_HIP(hipGetLastError()); }
// Gauss-Jordan with partial pivoting (the reference: cuBLAS matinvBatched): A (column-major, destroyed), R = A^-1; entry e of
// this thread is A[e * nt]
__device__ inline void lds_gauss_jordan(double *A, double *R, int d, int nt) {
  for (int k = 0; k < d; ++k) {
    int piv = k;
    for (int r = k + 1; r < d; ++r) if (fabs(A[(r + k * d) * nt]) > fabs(A[(piv + k * d) * nt])) piv = r;
    if (piv != k)
      for (int c = 0; c < d; ++c) {
        double t = A[(k + c * d) * nt]; A[(k + c * d) * nt] = A[(piv + c * d) * nt]; A[(piv + c * d) * nt] = t;
        t = R[(k + c * d) * nt]; R[(k + c * d) * nt] = R[(piv + c * d) * nt]; R[(piv + c * d) * nt] = t;
      }
    const double ip = 1.0 / A[(k + k * d) * nt];
    for (int c = 0; c < d; ++c) { A[(k + c * d) * nt] *= ip; R[(k + c * d) * nt] *= ip; }
    for (int r = 0; r < d; ++r) {
      if (r == k) continue;
      const double f = A[(r + k * d) * nt];
      for (int c = 0; c < d; ++c) { A[(r + c * d) * nt] -= f * A[(k + c * d) * nt]; R[(r + c * d) * nt] -= f * R[(k + c * d) * nt]; }
    }
  }
}

// ops/active.hpp:14-31 flag_active_vertices_kernel: over ALL stored factors, those active at `level` mark slot I's vertex
template <typename VD> __global__ void k_flag_vertices_level(const uint8_t *factor_active, size_t nf, const size_t *ids, size_t N, size_t I, uint8_t level, uint8_t *state) {
  const size_t f = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (f >= nf || !is_factor_active(factor_active[f], level)) return;
  state[ids[f * N + I]] |= 0x80;
}
template <typename VD> __global__ void k_flag_vertices(const size_t *active_ids, size_t n_active, const size_t *ids, size_t N, size_t I, uint8_t *state) {
  const size_t a = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (a >= n_active) return;
  state[ids[active_ids[a] * N + I]] |= 0x80; // ops/active.hpp:14-30 (XOR'd afterwards by the graph)
}
} // namespace detail

template <typename T, typename S, typename FTraits> class FactorDescriptor : public BaseFactorDescriptor<T, S> {
public:
  using Traits = FTraits;
  using Scalar = T;
  using Storage = S;
  using ObservationType = typename Traits::Observation;
  using ConstraintDataType = typename Traits::Data;
  using LossType = typename Traits::Loss;
  using VDTuple = typename Traits::VertexDescriptors;
  static constexpr size_t N = std::tuple_size<VDTuple>::value;
  static constexpr size_t E = Traits::dimension;
  static constexpr size_t error_dim = E;
  static constexpr size_t get_num_vertices() { return N; }

  struct JacobianStorage { hbm_vector<S> data; size_t dimensions[2] = {0, 0}; };

  std::array<BaseVertexDescriptor<T, S> *, N> vertex_descriptors{};
  std::array<void *, N> typed_descriptors{};
  std::vector<size_t> host_ids;            // global ids, N per factor
  managed_vector<size_t> device_ids;       // local ids, N per factor
  managed_vector<ObservationType> device_obs;
  managed_vector<ConstraintDataType> data;
  managed_vector<LossType> loss;
  managed_vector<S> precision_matrices;    // E*E per factor, read row-major
  managed_vector<uint8_t> active;
  managed_vector<size_t> active_indices;
  hbm_vector<T> residuals, chi2_vec, work;
  hbm_vector<S> chi2_derivative;
  std::array<JacobianStorage, N> jacobians;
  managed_vector<T> scalar;                // device scalar for reductions
  hbm_vector<T> sum_partials;
  hbm_vector<size_t> d_sparse_dst;         // [active factor][vertex pair] -> value offset in the block-sparse Hessian (written by k_sparse_dst)
  std::array<hbm_vector<size_t>, N> slot_vdiag; // [slot][vertex] -> value offset of the vertex's diagonal block (npos: none)
  hbm_vector<T> diag_blocks;               // gathered diagonal blocks of one slot
  HandleManager<size_t> hm;                // factor.hpp:158-174: ids returned by add_factor are stable handles
  std::unordered_map<size_t, size_t> global_to_local_map;
  std::vector<size_t> local_to_global_map;
  bool store_jacobians = true;
  const T *dynamic_scales = nullptr; // column scales of the current linearisation (dynamic Jacobians only)
  // active factors grouped by vertex, per slot (k_gather): built by initialize(); GRAPHITE_GENERIC_ATOMICS=1 (debugging
  // override, read once per initialize) keeps the reference-style atomic kernels instead
  std::array<hbm_vector<size_t>, N> slot_vptr, slot_vfac;
  std::array<int, N> slot_lanes{};
  bool gather_ready = false;

  template <typename... VDs> explicit FactorDescriptor(VDs *...vds) {
    static_assert(sizeof...(VDs) == N, "one vertex descriptor per slot");
    size_t i = 0;
    ((vertex_descriptors[i] = vds, typed_descriptors[i] = vds, ++i), ...);
    scalar.resize(1);
  }
  void reserve(size_t n) {
    host_ids.reserve(N * n); device_ids.reserve(N * n); device_obs.reserve(n); data.reserve(n); loss.reserve(n);
    precision_matrices.reserve(E * E * n); active.reserve(n); residuals.reserve(E * n); chi2_vec.reserve(n); chi2_derivative.reserve(n);
    work.reserve(E * n); global_to_local_map.reserve(n); local_to_global_map.reserve(n);
  }
  // factor.hpp:373-412; precision_matrix == nullptr -> identity
  size_t add_factor(const std::array<size_t, N> &ids, const ObservationType &obs, const S *precision_matrix,
                    const ConstraintDataType &constraint_data, const LossType &loss_function) {
    tables_mirrored = false; ++this->structure_epoch;
    const size_t handle = hm.get(), id = internal_count(); // id: local index
    global_to_local_map.insert({handle, id});
    local_to_global_map.push_back(handle);
    for (size_t i = 0; i < N; ++i) { host_ids.push_back(ids[i]); device_ids.push_back(0); }
    device_obs.push_back(obs); data.push_back(constraint_data); loss.push_back(loss_function);
    for (size_t i = 0; i < E; ++i)
      for (size_t j = 0; j < E; ++j) precision_matrices.push_back(precision_matrix ? precision_matrix[i * E + j] : (i == j ? S(1) : S(0)));
    active.push_back(0);
    residuals.resize(E * (id + 1)); chi2_vec.resize(id + 1); chi2_derivative.resize(id + 1); work.resize(E * (id + 1));
    return handle; // only global within this descriptor (factor.hpp:411)
  }
  // local index of a factor handle; std::out_of_range for an unknown id, like the reference's .at() (factor.hpp:460)
  size_t local_id(size_t handle) const { return global_to_local_map.at(handle); }
  void remove_factor(size_t handle) { // swap with last, fix the id maps, release the handle (factor.hpp:308-371)
    tables_mirrored = false; ++this->structure_epoch;
    auto it = global_to_local_map.find(handle);
    if (it == global_to_local_map.end()) { std::cerr << "Factor with id " << handle << " not found." << std::endl; return; }
    const size_t id = it->second, last = internal_count() - 1;
    const size_t last_handle = local_to_global_map[last];
    global_to_local_map[last_handle] = id;
    local_to_global_map[id] = last_handle;
    global_to_local_map.erase(handle);
    local_to_global_map.pop_back();
    hm.release(handle);
    for (size_t i = 0; i < N; ++i) host_ids[id * N + i] = host_ids[last * N + i];
    device_obs[id] = device_obs[last]; data[id] = data[last]; loss[id] = loss[last]; active[id] = active[last];
    for (size_t i = 0; i < E * E; ++i) precision_matrices[id * E * E + i] = precision_matrices[last * E * E + i];
    host_ids.resize(last * N); device_ids.resize(last * N); device_obs.pop_back(); data.pop_back(); loss.pop_back(); active.pop_back();
    precision_matrices.resize(last * E * E); residuals.resize(E * last); chi2_vec.resize(last); chi2_derivative.resize(last); work.resize(E * last);
  }
  void set_active(size_t handle, uint8_t active_value) { tables_mirrored = false; ++this->structure_epoch; const size_t id = local_id(handle); active[id] = (active[id] & 0x80) | (active_value & 0x7F); } // factor.hpp:419-431
  void reset_active() { ++this->structure_epoch; for (size_t i = 0; i < active.size(); ++i) active[i] = 0; }
  // factor.hpp:626-640: false = no stored Jacobians, every product recomputes the analytic blocks (Manual
  // differentiation only; an Auto factor keeps storing, ops/linearize.hpp:109)
  void set_jacobian_storage(bool on) { if (store_jacobians != on) ++this->structure_epoch; store_jacobians = on; }
  bool dynamic_jacobians() const { return !store_jacobians && supports_dynamic_jacobians(); }
  size_t add_factor(const std::array<size_t, N> &ids, const ObservationType &obs) { return add_factor(ids, obs, nullptr, ConstraintDataType(), LossType()); }
  static constexpr bool use_autodiff() { return std::is_same<typename Traits::Differentiation, DifferentiationMode::Auto>::value; }
  static constexpr bool supports_dynamic_jacobians() { return !use_autodiff(); }
  void initialize_device_ids(uint8_t level) { initialize(level); } // factor.hpp:439-470
  void to_device() {}
  void initialize_jacobian_storage() {}
  void compute_jacobians(StreamPool &) { compute_jacobians(); }
  size_t internal_count() const override { return device_obs.size(); }
  size_t active_count() const override { return active_indices.size(); }
  std::array<size_t, N> get_vertex_ids(size_t handle) const { const size_t id = local_id(handle); std::array<size_t, N> r; for (size_t i = 0; i < N; ++i) r[i] = host_ids[id * N + i]; return r; }
  const ObservationType &get_observation(size_t handle) const { return device_obs[local_id(handle)]; }
  const ConstraintDataType &get_constraint_data(size_t handle) const { return data[local_id(handle)]; }
  void clear() {
    tables_mirrored = false; ++this->structure_epoch;
    host_ids.clear(); device_ids.clear(); device_obs.clear(); data.clear(); loss.clear(); precision_matrices.clear(); active.clear();
    active_indices.clear(); residuals.clear(); chi2_vec.clear(); chi2_derivative.clear(); work.clear();
    global_to_local_map.clear(); local_to_global_map.clear(); hm.clear();
  }

  size_t num_slots() const override { return N; }
  size_t error_dimension() const override { return E; }
  BaseVertexDescriptor<T, S> *slot_descriptor(size_t s) const override { return vertex_descriptors[s]; }

  void initialize(uint8_t level, bool light = false) override {
    const size_t nf = internal_count();
    // global -> local vertex ids of every factor slot (factor.hpp:439-470): 1.36 M look-ups on Ladybug-1723, on a few threads
    // (the descriptors' dense id tables are built before the threads start; an unknown id throws std::out_of_range as before)
    for (size_t i = 0; i < N && nf; ++i) (void)vertex_descriptors[i]->get_local_id(host_ids[i]);
    detail::parallel_chunks(nf, 1 << 15, [&](size_t b, size_t e, size_t) {
      for (size_t f = b; f < e; ++f)
        for (size_t i = 0; i < N; ++i) device_ids[f * N + i] = vertex_descriptors[i]->get_local_id(host_ids[f * N + i]);
    });
    size_t na = 0;
    for (size_t f = 0; f < nf; ++f) na += detail::is_factor_active(active[f], level);
    active_indices.resize(na);
    for (size_t f = 0, a = 0; f < nf; ++f) if (detail::is_factor_active(active[f], level)) active_indices[a++] = f;
    refresh_table_mirrors(light);
    gather_ready = false; jacobians_sized = false;
    if (light) return;
    init_jacobians(std::make_index_sequence<N>{});
    build_vertex_lists();
    // (read by gather_one after the synchronisation that ends Graph::initialize_optimization)
    pmat_flag.resize(1); pmat_flag[0] = 0;
    if (nf) detail::k_pmat_asymmetric<S><<<detail::blocks(nf), detail::TPB>>>(m_pmat.get(precision_matrices, tables_mirrored), nf, E, pmat_flag.raw());
  }
  managed_vector<int> pmat_flag; // [0] != 0: some precision matrix is not symmetric
  void build_vertex_lists() {
    gather_ready = false;
    if (getenv("GRAPHITE_GENERIC_ATOMICS") && atoi(getenv("GRAPHITE_GENERIC_ATOMICS")) != 0) return;
    const size_t na = active_count();
    for (size_t i = 0; i < N; ++i) {
      const size_t nv = vertex_descriptors[i]->count();
      std::vector<size_t> ptr(nv + 1, 0), fac(na);
      for (size_t a = 0; a < na; ++a) ptr[device_ids[active_indices[a] * N + i] + 1]++;
      size_t used = 0;
      for (size_t v = 0; v < nv; ++v) { used += ptr[v + 1] != 0; ptr[v + 1] += ptr[v]; }
      std::vector<size_t> fill(ptr.begin(), ptr.end() - 1);
      for (size_t a = 0; a < na; ++a) { const size_t f = active_indices[a]; fac[fill[device_ids[f * N + i]]++] = f; } // ascending factor order per vertex
      slot_vptr[i].assign(ptr.data(), ptr.size());
      slot_vfac[i].assign(fac.data(), fac.size());
      // lanes per vertex: half the mean degree of the vertices in use, rounded up to a power of two, at most a wave — or a
      // whole workgroup (k_gather: W = 256) from a mean degree of 768 on (300 cameras x 1 000 factors: 1.01 -> 0.98 ms per LM iteration; at 394 factors per camera it lost: 145 -> 204 us for the camera blocks)
      const size_t mean = used ? (na + used - 1) / used : 1;
      int w = 1;
      while (w < 64 && (size_t)(2 * w) <= mean) w <<= 1;
      if (mean >= 768) w = detail::TPB;
      slot_lanes[i] = w;
    }
    gather_ready = true;
  }
  void flag_active_vertices() override {
    for (size_t i = 0; i < N; ++i)
      if (active_count()) detail::k_flag_vertices<void><<<detail::blocks(active_count()), detail::TPB>>>(m_active.get(active_indices, tables_mirrored), active_count(), m_ids.get(device_ids, tables_mirrored), N, i, vertex_descriptors[i]->get_active_state());
  }
  // factor.hpp:226-228 / ops/active.hpp:33-59: the same marks taken from the per-factor activity bytes at a GIVEN level (every stored
  // factor is tested, not the active list of the last initialize_device_ids) — what tests/factor.cu:324-358 drives directly
  void flag_active_vertices_async(const uint8_t level) {
    const size_t nf = internal_count();
    for (size_t i = 0; i < N && nf; ++i)
      detail::k_flag_vertices_level<void><<<detail::blocks(nf), detail::TPB>>>(active.raw(), nf, device_ids.raw(), N, i, level, vertex_descriptors[i]->get_active_state());
  }

  // ---- the reference's own Hessian-assembly members (factor.hpp:657-825), for callers that drive the assembly themselves ------------
  // (tests/factor.cu:854-967 does: coordinates -> its own block -> offset map -> setup -> execute).  Hessian::build_structure /
  // update_values (sparse.hpp) do the same work through sorted block keys and k_sparse_pair; these three members run THAT kernel on
  // the caller's offsets.  Vector arguments: anything with to_host() / operator=(std::vector) / raw() (sparse.hpp device_vector).
  // One coordinate per (slot pair i <= j, active factor) whose two vertices are active; row = the lower block index.
  template <typename DV> void get_hessian_block_coordinates(DV &block_coords) {
    detail::sync();
    std::vector<BlockCoordinates> h = block_coords.to_host();
    for (size_t i = 0; i < N; ++i)
      for (size_t j = i; j < N; ++j) {
        const uint8_t *si = vertex_descriptors[i]->get_active_state(), *sj = vertex_descriptors[j]->get_active_state();
        const size_t *bi = vertex_descriptors[i]->get_block_ids(), *bj = vertex_descriptors[j]->get_block_ids();
        for (size_t a = 0; a < active_count(); ++a) {
          const size_t f = active_indices[a], vi = device_ids[f * N + i], vj = device_ids[f * N + j];
          if (!detail::is_vertex_active(si, vi) || !detail::is_vertex_active(sj, vj)) continue;
          const size_t r = bi[vi], c = bj[vj];
          h.push_back(r > c ? BlockCoordinates{c, r} : BlockCoordinates{r, c});
        }
      }
    block_coords = h;
  }
  // h_block_offsets[(pair index) * active_count() + a] = value offset of the block that pair of factor a adds to (0 where a vertex
  // of the pair is not active); returns pairs x active factors (factor.hpp:702-765)
  template <typename DV> size_t setup_hessian_computation(std::unordered_map<BlockCoordinates, size_t> &block_indices, DV & /*d_hessian*/, size_t *h_block_offsets, StreamPool &) {
    detail::sync();
    size_t w = 0;
    for (size_t i = 0; i < N; ++i)
      for (size_t j = i; j < N; ++j) {
        const uint8_t *si = vertex_descriptors[i]->get_active_state(), *sj = vertex_descriptors[j]->get_active_state();
        const size_t *bi = vertex_descriptors[i]->get_block_ids(), *bj = vertex_descriptors[j]->get_block_ids();
        for (size_t a = 0; a < active_count(); ++a) {
          const size_t f = active_indices[a], vi = device_ids[f * N + i], vj = device_ids[f * N + j];
          size_t off = 0;
          if (detail::is_vertex_active(si, vi) && detail::is_vertex_active(sj, vj)) {
            const size_t r = bi[vi], c = bj[vj];
            auto it = block_indices.find(r > c ? BlockCoordinates{c, r} : BlockCoordinates{r, c});
            if (it != block_indices.end()) off = it->second;
            else std::cerr << "Error: Hessian block coordinate not found!" << std::endl;
          }
          h_block_offsets[w++] = off;
        }
      }
    return NUM_PAIRS * active_count();
  }
  // rho' J_i^T P J_j of every pair of every active factor, added into d_hessian at the offsets setup_hessian_computation produced
  // (device copy of them: d_block_offsets), blocks column-major with the rows of the vertex that has the lower scalar column
  // (ops/hessian.hpp:10-78); returns pairs x active factors (factor.hpp:773-825)
  template <typename DV> size_t execute_hessian_computation(std::unordered_map<BlockCoordinates, size_t> &, DV &d_hessian, const size_t *d_block_offsets, StreamPool &) {
    const size_t na = active_count();
    if (!na) return 0;
    detail::sync();
    std::vector<size_t> off(NUM_PAIRS * na), dst(NUM_PAIRS * na);
    GRAPHITE_HIP(hipMemcpy(off.data(), d_block_offsets, off.size() * sizeof(size_t), hipMemcpyDeviceToHost));
    size_t p = 0;
    for (size_t i = 0; i < N; ++i)
      for (size_t j = i; j < N; ++j, ++p) {
        const uint8_t *si = vertex_descriptors[i]->get_active_state(), *sj = vertex_descriptors[j]->get_active_state();
        const size_t *hi = vertex_descriptors[i]->get_hessian_ids(), *hj = vertex_descriptors[j]->get_hessian_ids();
        for (size_t a = 0; a < na; ++a) {
          const size_t f = active_indices[a], vi = device_ids[f * N + i], vj = device_ids[f * N + j];
          size_t d = ~size_t(0); // k_sparse_pair: no block
          if (detail::is_vertex_active(si, vi) && detail::is_vertex_active(sj, vj)) d = off[p * na + a] | (hi[vi] > hj[vj] ? size_t(1) << 63 : 0);
          dst[a * NUM_PAIRS + p] = d;
        }
      }
    d_sparse_dst.resize_uninit(dst.size());
    GRAPHITE_HIP(hipMemcpy(d_sparse_dst.raw(), dst.data(), dst.size() * sizeof(size_t), hipMemcpyHostToDevice));
    const bool was_ready = gather_ready;
    gather_ready = false; // every pair through k_sparse_pair (the per-vertex gather of the diagonal pairs needs sparse_setup's tables)
    sparse_all(d_hessian.raw(), std::make_index_sequence<N>{});
    detail::sync();
    gather_ready = was_ready;
    return NUM_PAIRS * na;
  }

  // HBM copies of the per-factor tables the kernels only read (built on the host in pinned memory: ids, observations,
  // constraint data, losses, precision matrices, the active list); refreshed by initialize(), dropped by any mutation
  template <typename U> struct TableMirror {
    static constexpr bool ok = std::is_trivially_copyable<U>::value && !std::is_empty<U>::value;
    struct Nothing {};
    std::conditional_t<ok, hbm_vector<U>, Nothing> d;
    const U *get(const managed_vector<U> &h, bool valid) const {
      if constexpr (ok) { if (valid && d.size() == h.size() && h.size()) return d.raw(); }
      return h.raw();
    }
    void refresh(const managed_vector<U> &h) { if constexpr (ok) d.assign(h.raw(), h.size()); }
    void drop() { if constexpr (ok) d.resize(0); }
  };
  TableMirror<size_t> m_active, m_ids;
  TableMirror<ObservationType> m_obs;
  TableMirror<ConstraintDataType> m_data;
  TableMirror<LossType> m_loss;
  TableMirror<S> m_pmat;
  bool tables_mirrored = false;
  // light (engine hand-over: probe + final residuals only): the precision matrices stay in pinned host memory — their HBM copy
  // is 32 bytes per factor that only chi2() of the generic kernels would read
  void refresh_table_mirrors(bool light = false) {
    m_active.refresh(active_indices); m_ids.refresh(device_ids); m_obs.refresh(device_obs); m_data.refresh(data);
    m_loss.refresh(loss);
    if (light) m_pmat.drop(); else m_pmat.refresh(precision_matrices);
    tables_mirrored = true;
  }
  detail::FactorView<FactorDescriptor> view() {
    detail::FactorView<FactorDescriptor> fv;
    const bool mv = tables_mirrored;
    fv.active_ids = m_active.get(active_indices, mv); fv.n_active = active_count(); fv.ids = m_ids.get(device_ids, mv); fv.obs = m_obs.get(device_obs, mv);
    fv.data = m_data.get(data, mv); fv.loss = m_loss.get(loss, mv); fv.pmat = m_pmat.get(precision_matrices, mv); fv.residuals = residuals.raw();
    fv.chi2 = chi2_vec.raw(); fv.dchi2 = chi2_derivative.raw();
    fv.dynamic = dynamic_jacobians(); fv.scales = dynamic_scales;
    fill_view(fv, std::make_index_sequence<N>{});
    return fv;
  }
  void compute_error() override {
    if (!active_count()) return;
    detail::k_error<FactorDescriptor><<<detail::blocks(active_count()), detail::TPB>>>(view(), std::make_index_sequence<N>{});
  }
  void compute_jacobians() override {
    dynamic_scales = nullptr; // a new linearisation: the scalar diagonal is taken from unscaled blocks
    if (!jacobians_sized) init_jacobians(std::make_index_sequence<N>{});
    if (!dynamic_jacobians()) jac_all(std::make_index_sequence<N>{});
  }
  void compute_chi2() override {
    if (active_count()) detail::k_chi2<FactorDescriptor><<<detail::blocks(active_count()), detail::TPB>>>(view());
  }
  T chi2() override { // factor.hpp:551-557
    compute_chi2();
    if (!active_count()) return T(0);
    const int nb = (int)std::max<size_t>(1, std::min<size_t>(detail::SUM_BLOCKS, (active_count() + 4 * detail::TPB - 1) / (4 * detail::TPB)));
    if (sum_partials.size() < (size_t)detail::SUM_BLOCKS) sum_partials.resize(detail::SUM_BLOCKS);
    detail::k_sum_active<T><<<nb, detail::TPB>>>(chi2_vec.raw(), m_active.get(active_indices, tables_mirrored), active_count(), sum_partials.raw());
    detail::k_sum_partials<T><<<1, 64>>>(sum_partials.raw(), nb, scalar.raw());
    detail::sync();
    return scalar[0];
  }
  T chi2(size_t handle) {
    T v;
    GRAPHITE_HIP(hipMemcpy(&v, chi2_vec.raw() + local_id(handle), sizeof(T), hipMemcpyDeviceToHost));
    return v;
  }
  void scalar_diagonal(T *diag) override { if (gather_ready) gather_all<0>(diag, nullptr, std::make_index_sequence<N>{}); else slot_all<0>(diag, nullptr, std::make_index_sequence<N>{}); }
  void scale_jacobians(const T *scales) override {
    if (dynamic_jacobians()) dynamic_scales = scales;
    else slot_all<1>(nullptr, scales, std::make_index_sequence<N>{});
  }
  void compute_b(T *b) override { if (gather_ready) gather_all<2>(b, nullptr, std::make_index_sequence<N>{}); else slot_all<2>(b, nullptr, std::make_index_sequence<N>{}); }
  void compute_Jtv(T *out, const T *res) override { if (gather_ready) gather_all<3>(out, res, std::make_index_sequence<N>{}); else slot_all<3>(out, res, std::make_index_sequence<N>{}); }
  void compute_Jv(T *res, const T *x) override {
    if (gather_ready) { if (active_count()) detail::k_Jv_all<FactorDescriptor, true><<<detail::blocks(active_count() * E), detail::TPB>>>(view(), res, x, std::make_index_sequence<N>{}); }
    else jv_all(res, x, std::make_index_sequence<N>{});
  }
  bool compute_Jv_fresh(T *res, const T *x) override {
    if (!gather_ready) return false;
    if (active_count()) detail::k_Jv_all<FactorDescriptor, false><<<detail::blocks(active_count() * E), detail::TPB>>>(view(), res, x, std::make_index_sequence<N>{});
    return true;
  }
  T *work_residual() override { return work.raw(); }
  void block_diagonal(size_t slot, T *blocks) override { block_one(slot, blocks, std::make_index_sequence<N>{}); }
  void dense_hessian(T *H, size_t n) override { dense_all(H, n, std::make_index_sequence<N>{}); }
  // ---- block-sparse Hessian (sparse.hpp) ------------------------------------------------------
  static constexpr size_t NUM_PAIRS = N * (N + 1) / 2;
  size_t block_key_count() override { return active_count() * (N * (N - 1) / 2); }
  void emit_block_keys(uint64_t *keys, const size_t *s2b) override {
    if (N > 1 && active_count()) detail::k_block_keys<FactorDescriptor><<<detail::blocks(active_count()), detail::TPB>>>(view(), s2b, keys);
  }
  void sparse_setup(const uint64_t *keys, size_t num_keys, const size_t *value_offsets, const size_t *s2b) override {
    d_sparse_dst.resize_uninit(active_count() * NUM_PAIRS);
    detail::VertexDiag<N> vd;
    for (size_t i = 0; i < N; ++i) {
      const size_t nv = vertex_descriptors[i]->count();
      slot_vdiag[i].resize_uninit(nv);
      if (nv) GRAPHITE_HIP(hipMemset(slot_vdiag[i].raw(), 0xff, nv * sizeof(size_t))); // npos: no diagonal block through this slot
      vd.p[i] = slot_vdiag[i].raw();
    }
    if (active_count()) detail::k_sparse_dst<FactorDescriptor><<<detail::blocks(active_count()), detail::TPB>>>(view(), s2b, keys, num_keys, value_offsets, d_sparse_dst.raw(), vd);
  }
  void sparse_hessian(S *values) override {
    if constexpr (is_low_precision<S>::value) { (void)values; throw std::invalid_argument("block-sparse Hessian: 16-bit storage types are not supported (the reference refuses them too, bal.cu:181-203)"); }
    else sparse_all(values, std::make_index_sequence<N>{});
  }

  bool export_bal(std::vector<int32_t> &cam, std::vector<int32_t> &pt, std::vector<T> &obs, int &loss_kind, double &loss_delta) override {
    if constexpr (bal_shaped()) {
      constexpr bool plain = std::is_same<LossType, DefaultLoss<T, 2>>::value, huber = std::is_same<LossType, HuberLoss<T, 2>>::value;
      if constexpr (plain || huber) {
        const size_t na = active_count();
        if (!na) return false;
        detail::sync();
        loss_kind = huber ? 1 : 0; loss_delta = 0;
        cam.resize(na); pt.resize(na); obs.resize(2 * na);
        if constexpr (huber) loss_delta = (double)loss[active_indices[0]].delta;
        const double delta0 = loss_delta;
        std::atomic<bool> representable{true};
        detail::parallel_chunks(na, 1 << 15, [&](size_t b, size_t e, size_t) {
          for (size_t a = b; a < e; ++a) { // the ACTIVE factors, in active_indices order (factor.hpp:433-465)
            const size_t f = active_indices[a];
            for (size_t i = 0; i < 2; ++i)
              for (size_t j = 0; j < 2; ++j)
                if (precision_matrices[f * 4 + i * 2 + j] != (i == j ? S(1) : S(0))) representable = false;
            if constexpr (huber) { if ((double)loss[f].delta != delta0) representable = false; }
            cam[a] = (int32_t)device_ids[2 * f]; pt[a] = (int32_t)device_ids[2 * f + 1];
            obs[2 * a] = (T)detail::obs_component(device_obs[f], 0, 0); obs[2 * a + 1] = (T)detail::obs_component(device_obs[f], 1, 0);
          }
        });
        return representable.load();
      }
    }
    return false;
  }
  uint64_t content_fingerprint() const override {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)internal_count();
    h = detail::digest(h, host_ids.data(), host_ids.size() * sizeof(size_t));
    if constexpr (std::is_trivially_copyable<ObservationType>::value) h = detail::digest(h, device_obs.raw(), device_obs.size() * sizeof(ObservationType));
    h = detail::digest(h, active.raw(), active.size());
    h = detail::digest(h, precision_matrices.raw(), precision_matrices.size() * sizeof(S));
    if constexpr (std::is_trivially_copyable<ConstraintDataType>::value && !std::is_empty<ConstraintDataType>::value) h = detail::digest(h, data.raw(), data.size() * sizeof(ConstraintDataType));
    if constexpr (std::is_trivially_copyable<LossType>::value && !std::is_empty<LossType>::value && !std::is_polymorphic<LossType>::value)
      h = detail::digest(h, loss.raw(), loss.size() * sizeof(LossType));
    return h;
  }

  // what the engine can represent at all: two slots of dimension 9 and 3, a two-component residual and observation, no
  // per-factor constraint data (whether the user's functions ARE the engine's model is then verified by value, probe_bal)
  static constexpr bool bal_shaped() {
    if constexpr (N == 2 && E == 2 && std::is_empty<ConstraintDataType>::value && detail::obs_indexable<ObservationType>::value)
      return std::tuple_element<0, VDTuple>::type::Traits::dimension == 9 && std::tuple_element<1, VDTuple>::type::Traits::dimension == 3;
    else return false;
  }
  bool declares_bal_model() const override { return detail::has_bal_tag<Traits>::value; }
  std::shared_ptr<detail::EngineModelBase> make_engine_model(std::vector<int32_t> &cam, std::vector<int32_t> &pt, size_t num_poses, size_t num_landmarks) override; // engine_model.hpp
  bool pose_engine_factor(detail::pe::FactorInfo &fi, bool check_symmetry) override; // engine_pose.hpp
  void pose_engine_ids(int *lij) override;
  void pose_engine_launch(const detail::pe::FactorArgs<T> &fa, void *mirror, int mode) override;
  bool pose_stage_attr_set = false; // the factor kernel's LDS staging above 64 KB was allowed once
  bool probe_bal(size_t max_samples, std::vector<T> &cam, std::vector<T> &pt, std::vector<T> &obs, std::vector<T> &res, std::vector<T> &Jc, std::vector<T> &Jp, double &update_dev,
                 size_t *num_synthetic = nullptr) override {
    if constexpr (bal_shaped()) {
      const size_t na = active_count();
      if (!na || !max_samples) return false;
      const size_t ns = std::min(na, max_samples), stride = na / ns;
      const size_t nsyn = (detail::PROBE_VARIANTS - 1) * std::min<size_t>(ns, 8); // every branch-steering variant on up to 8 of the sampled factors
      const size_t nt = ns + nsyn;
      if (num_synthetic) *num_synthetic = nsyn;
      hbm_vector<T> out(nt * detail::PROBE_W);
      detail::k_probe_bal<FactorDescriptor><<<detail::blocks(nt), detail::TPB>>>(view(), stride, ns, nsyn, out.raw(), std::make_index_sequence<N>{});
      detail::sync();
      const std::vector<T> h = out.to_host();
      cam.resize(9 * nt); pt.resize(3 * nt); obs.resize(2 * nt); res.resize(2 * nt); Jc.resize(18 * nt); Jp.resize(6 * nt);
      update_dev = 0;
      for (size_t a = 0; a < nt; ++a) {
        const T *o = h.data() + a * detail::PROBE_W;
        std::copy(o, o + 9, cam.begin() + 9 * a); std::copy(o + 9, o + 12, pt.begin() + 3 * a); std::copy(o + 12, o + 14, obs.begin() + 2 * a);
        std::copy(o + 14, o + 16, res.begin() + 2 * a); std::copy(o + 16, o + 34, Jc.begin() + 18 * a); std::copy(o + 34, o + 40, Jp.begin() + 6 * a);
        const double d = (double)o[40];
        if (!(d <= update_dev)) update_dev = d; // NaN sticks
      }
      return true;
    } else { (void)max_samples; (void)cam; (void)pt; (void)obs; (void)res; (void)Jc; (void)Jp; (void)update_dev; (void)num_synthetic; return false; }
  }

private:
  bool jacobians_sized = false; // a light initialize() leaves the storage to the first compute_jacobians()
  template <size_t... Is> void init_jacobians(std::index_sequence<Is...>) {
    jacobians_sized = true;
    ((jacobians[Is].dimensions[0] = E, jacobians[Is].dimensions[1] = detail::slot_dim<FactorDescriptor, Is>(),
      jacobians[Is].data.resize(dynamic_jacobians() ? 0 : E * detail::slot_dim<FactorDescriptor, Is>() * internal_count())), ...);
  }
  template <size_t... Is> void fill_view(detail::FactorView<FactorDescriptor> &fv, std::index_sequence<Is...>) {
    ((fv.jac[Is] = jacobians[Is].data.raw(),
      fv.verts[Is] = static_cast<typename std::tuple_element<Is, VDTuple>::type *>(typed_descriptors[Is])->vertices(),
      fv.vstate[Is] = vertex_descriptors[Is]->device_active_state(), fv.hid[Is] = vertex_descriptors[Is]->device_hessian_ids()), ...);
  }
  template <size_t... Is> void jac_all(std::index_sequence<Is...> seq) {
    if (!active_count()) return;
    constexpr bool manual = std::is_same<typename Traits::Differentiation, DifferentiationMode::Manual>::value;
    auto fv = view();
    // the reference clears the Jacobian storage first (ops/linearize.hpp:127): fixed vertices keep zeros
    ((detail::fill<S>(jacobians[Is].data.raw(), jacobians[Is].data.size(), S(0)),
      detail::k_jacobian<FactorDescriptor, Is><<<detail::blocks(active_count() * (manual ? 1 : detail::slot_dim<FactorDescriptor, Is>())), detail::TPB>>>(fv, seq)), ...);
  }
  template <int WHICH, size_t... Is> void slot_all(T *out, const T *in, std::index_sequence<Is...>) {
    if (!active_count()) return;
    auto fv = view();
    ((detail::k_slot<FactorDescriptor, Is, WHICH><<<detail::blocks(active_count() * detail::slot_dim<FactorDescriptor, Is>()), detail::TPB>>>(fv, out, in)), ...);
  }
  template <size_t... Is> void jv_all(T *res, const T *x, std::index_sequence<Is...>) {
    if (!active_count()) return;
    auto fv = view();
    ((detail::k_Jv<FactorDescriptor, Is><<<detail::blocks(active_count() * E), detail::TPB>>>(fv, res, x)), ...);
  }
  template <int WHICH, size_t I> void gather_one(detail::FactorView<FactorDescriptor> &fv, T *out, const T *in) {
    constexpr size_t dI = detail::slot_dim<FactorDescriptor, I>(), rbI = dI < detail::GATHER_ROWS ? dI : detail::GATHER_ROWS;
    const size_t nv = vertex_descriptors[I]->count();
    if (!nv) return;
    if constexpr (WHICH == 4 && dI <= detail::GATHER_SYM_MAX_DIM) {
      if (pmat_flag.size() && pmat_flag[0] == 0) { // symmetric precision matrices (checked by initialize()): one pass, the block read once
        detail::k_gather<FactorDescriptor, I, WHICH, true><<<detail::blocks(nv * (size_t)slot_lanes[I]), detail::TPB>>>(fv, slot_vptr[I].raw(), slot_vfac[I].raw(), nv, slot_lanes[I], out, in);
        return;
      }
    }
    const size_t groups = nv * (WHICH == 4 ? (dI + rbI - 1) / rbI : 1);
    detail::k_gather<FactorDescriptor, I, WHICH><<<detail::blocks(groups * (size_t)slot_lanes[I]), detail::TPB>>>(fv, slot_vptr[I].raw(), slot_vfac[I].raw(), nv, slot_lanes[I], out, in);
  }
  template <int WHICH, size_t... Is> void gather_all(T *out, const T *in, std::index_sequence<Is...>) {
    if (!active_count()) return;
    auto fv = view();
    ((gather_one<WHICH, Is>(fv, out, in)), ...);
  }
  template <size_t... Is> void block_one(size_t slot, T *blocks, std::index_sequence<Is...>) {
    if (!active_count()) return;
    auto fv = view();
    if (gather_ready) { ((slot == Is ? gather_one<4, Is>(fv, blocks, nullptr) : (void)0), ...); return; }
    ((slot == Is ? (void)(detail::k_slot<FactorDescriptor, Is, 4><<<detail::blocks(active_count() * detail::slot_dim<FactorDescriptor, Is>()), detail::TPB>>>(fv, blocks, nullptr)) : (void)0), ...);
  }
  template <size_t I, size_t... Ks> void dense_row(detail::FactorView<FactorDescriptor> &fv, T *H, size_t n, std::index_sequence<Ks...>) {
    ((detail::k_dense_pair<FactorDescriptor, I, Ks><<<detail::blocks(active_count() * detail::slot_dim<FactorDescriptor, I>() * detail::slot_dim<FactorDescriptor, Ks>()), detail::TPB>>>(fv, H, n)), ...);
  }
  template <size_t I, size_t K> void sparse_pair(detail::FactorView<FactorDescriptor> &fv, S *values) {
    if constexpr (K >= I) {
      constexpr size_t di = detail::slot_dim<FactorDescriptor, I>(), dk = detail::slot_dim<FactorDescriptor, K>();
      if (K == I && gather_ready) {
        // a vertex's diagonal block collects one term per factor of the vertex: gathered per vertex (no atomics: a camera's
        // 81 entries would take 81 x its observations of them), then added to where the block lives
        const size_t nv = vertex_descriptors[I]->count();
        if (!nv) return;
        diag_blocks.resize_uninit(nv * di * di);
        GRAPHITE_HIP(hipMemsetAsync(diag_blocks.raw(), 0, nv * di * di * sizeof(T), 0));
        gather_one<4, I>(fv, diag_blocks.raw(), nullptr);
        detail::k_diag_store<T, S><<<detail::blocks(nv * di * di), detail::TPB>>>(nv, di * di, diag_blocks.raw(), slot_vdiag[I].raw(), values);
        return;
      }
      // pair index of (I, K) in the row-major upper enumeration used by sparse_setup
      detail::k_sparse_pair<FactorDescriptor, I, K><<<detail::blocks(active_count() * di * dk), detail::TPB>>>(fv, values, d_sparse_dst.raw(), I * N - I * (I - 1) / 2 + (K - I), NUM_PAIRS);
    }
  }
  template <size_t I, size_t... Ks> void sparse_row(detail::FactorView<FactorDescriptor> &fv, S *values, std::index_sequence<Ks...>) {
    ((sparse_pair<I, Ks>(fv, values)), ...);
  }
  template <size_t... Is> void sparse_all(S *values, std::index_sequence<Is...> seq) {
    if (!active_count()) return;
    auto fv = view();
    ((sparse_row<Is>(fv, values, seq)), ...);
  }
  template <size_t... Is> void dense_all(T *H, size_t n, std::index_sequence<Is...> seq) {
    if (!active_count()) return;
    auto fv = view();
    ((dense_row<Is>(fv, H, n, seq)), ...);
  }
};

// =================================================================================================
// Graph (graph.hpp:30-340)
// =================================================================================================
namespace detail {
template <typename T> __global__ void k_scales(T *diag_to_scale, size_t n) { // graph.hpp:262-270
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) diag_to_scale[i] = (T)(1.0 / (std::numeric_limits<double>::epsilon() + ::sqrt((double)diag_to_scale[i])));
}
template <typename T> __global__ void k_xor_msb(uint8_t *s, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) s[i] ^= 0x80;
}
template <typename T> __global__ void k_clear_msb(uint8_t *s, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) s[i] &= 0x7F;
}
} // namespace detail

template <typename T, typename S> class Graph {
  std::vector<BaseVertexDescriptor<T, S> *> vertex_descriptors;
  std::vector<BaseFactorDescriptor<T, S> *> factor_descriptors;
  hbm_vector<T> b, jacobian_scales;
  size_t hessian_dim = 0, pose_dim = 0, elimination_block = 0;
  std::vector<size_t> hessian_offsets; // scalar column of every block column (+ the dimension at the end), graph.hpp:40
  bool scale_jacobians_ = true;
public:
  // The first kernel launch out of a translation unit loads its whole code object (25 ms for a client of this header on
  // MI355X), the first call into libgraphite_mi355x.so that library's.  Both are one-time costs of the PROCESS; they are
  // taken when the first Graph is constructed instead of inside the first optimiser call (the reference's timing of
  // "Optimization took" in examples/bal.cu:254-271 brackets the optimiser call only).
  Graph() {
    static const bool warm = [] {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
      detail::k_clear_msb<T><<<1, 1>>>(nullptr, 0);
      (void)gr_warm_up(dev);
      (void)hipDeviceSynchronize();
      (void)hipGetLastError();
      return true;
    }();
    (void)warm;
  }
  // graph.hpp:48-55, :90
  size_t get_variable_dimension(const size_t block_index) const { return hessian_offsets[block_index + 1] - hessian_offsets[block_index]; }
  size_t get_num_block_columns() const { return hessian_offsets.empty() ? 0 : hessian_offsets.size() - 1; }
  const std::vector<size_t> &get_offset_vector() const { return hessian_offsets; }
  size_t get_elimination_block_column() const { return elimination_block; }
  void add_descriptor(BaseVertexDescriptor<T, S> *d) { vertex_descriptors.push_back(d); }
  void add_descriptor(BaseFactorDescriptor<T, S> *d) { factor_descriptors.push_back(d); }
  void add_vertex_descriptor(BaseVertexDescriptor<T, S> *d) { add_descriptor(d); }
  void add_factor_descriptor(BaseFactorDescriptor<T, S> *d) { add_descriptor(d); }
  std::vector<BaseVertexDescriptor<T, S> *> &get_vertex_descriptors() { return vertex_descriptors; }
  std::vector<BaseFactorDescriptor<T, S> *> &get_factor_descriptors() { return factor_descriptors; }
  void scale_system(bool on) { scale_jacobians_ = on; }
  bool scales_system() const { return scale_jacobians_; }
  size_t get_hessian_dimension() const { return hessian_dim; }
  // first column of the eliminated (set_eliminate) descriptors = dimension of the reduced system (pcg_schur.hpp:60-61)
  size_t get_pose_dimension() const { return pose_dim; }
  hbm_vector<T> &get_b() { return b; }
  hbm_vector<T> &get_jacobian_scales() { return jacobian_scales; }
  void clear() { vertex_descriptors.clear(); factor_descriptors.clear(); engine_cache.reset(); }
  // The engine problem (gr_bal_problem + staging buffers) that optimizer::levenberg_marquardt built for this graph, kept
  // between optimiser calls (solve.hpp, EngineCache; type-erased here, destroyed with the graph or by clear()): the SLAM-style
  // caller that optimises a slowly changing graph again and again (README.md:27) pays orderings, allocations and uploads
  // once per STRUCTURE, not once per call.
  std::shared_ptr<void> engine_cache;
  uint8_t last_init_level = 0;
  bool last_init_light = false;

  // graph.hpp:92-167: active factors, which vertices they use, then one scalar column range per
  // active vertex, descriptors in the order they were added, vertices by ascending global id
  bool initialize_optimization(uint8_t level = 0, bool light = false) {
    last_init_level = level; last_init_light = light;
    for (auto *vd : vertex_descriptors)
      if (vd->count()) detail::k_clear_msb<T><<<detail::blocks(vd->count()), detail::TPB>>>(vd->get_active_state(), vd->count());
    for (auto *fd : factor_descriptors) fd->initialize(level, light);
    for (auto *fd : factor_descriptors) fd->flag_active_vertices();
    for (auto *vd : vertex_descriptors)
      if (vd->count()) detail::k_xor_msb<T><<<detail::blocks(vd->count()), detail::TPB>>>(vd->get_active_state(), vd->count());
    detail::sync();
    size_t col = 0;
    hessian_offsets.clear();
    for (int pass = 0; pass < 2; ++pass) { // non-eliminated descriptors first, eliminated ones last (graph.hpp:100-149)
      if (pass == 1) { pose_dim = col; elimination_block = hessian_offsets.size(); }
      for (auto *vd : vertex_descriptors) {
        if ((int)vd->eliminate != pass) continue;
        // columns in ascending GLOBAL id; vertices added in id order (the usual case) need no sort
        const auto &l2g = vd->local_to_global();
        const uint8_t *state = vd->get_active_state();
        auto *hid = &vd->get_hessian_ids()[0];
        auto *bid = const_cast<size_t *>(vd->get_block_ids());
        const size_t nvd = vd->count(), dimv = vd->dimension();
        auto place = [&](size_t l) { if (detail::is_vertex_active(state, l)) { hid[l] = col; bid[l] = hessian_offsets.size(); hessian_offsets.push_back(col); col += dimv; } };
        if (std::is_sorted(l2g.begin(), l2g.begin() + nvd)) { for (size_t l = 0; l < nvd; ++l) place(l); }
        else {
          std::vector<std::pair<size_t, size_t>> order;
          order.reserve(nvd);
          for (size_t l = 0; l < nvd; ++l) order.emplace_back(l2g[l], l);
          std::sort(order.begin(), order.end());
          for (auto &e : order) place(e.second);
        }
      }
    }
    hessian_offsets.push_back(col);
    hessian_dim = col;
    b.resize(col); jacobian_scales.resize(col);
    return col > 0;
  }
  bool build_structure() { return true; }
  // vertex values / states / Hessian columns in HBM for the duration of an optimiser loop (VertexDescriptor::begin_mirror)
  void begin_device_mirror() { for (auto *vd : vertex_descriptors) vd->begin_mirror(); }
  void end_device_mirror() { for (auto *vd : vertex_descriptors) vd->end_mirror(); }
  struct DeviceMirrorScope {
    Graph *g;
    explicit DeviceMirrorScope(Graph *graph) : g(graph) { g->begin_device_mirror(); }
    ~DeviceMirrorScope() { g->end_device_mirror(); }
  };

  void compute_error() { for (auto *fd : factor_descriptors) fd->compute_error(); }
  T chi2() { T c = 0; for (auto *fd : factor_descriptors) c += fd->chi2(); return c; } // graph.hpp:212-225
  // graph.hpp:236-290
  void linearize(StreamPool &) { linearize(); }
  void linearize() {
    for (auto *fd : factor_descriptors) { fd->compute_error(); fd->compute_jacobians(); fd->compute_chi2(); }
    if (scale_jacobians_) {
      detail::fill<T>(jacobian_scales.raw(), hessian_dim, T(0));
      for (auto *fd : factor_descriptors) fd->scalar_diagonal(jacobian_scales.raw());
      if (hessian_dim) detail::k_scales<T><<<detail::blocks(hessian_dim), detail::TPB>>>(jacobian_scales.raw(), hessian_dim);
      for (auto *fd : factor_descriptors) fd->scale_jacobians(jacobian_scales.raw());
    } else detail::fill<T>(jacobian_scales.raw(), hessian_dim, T(1));
    detail::fill<T>(b.raw(), hessian_dim, T(0));
    for (auto *fd : factor_descriptors) fd->compute_b(b.raw());
    detail::sync();
  }
  void apply_update(const T *delta_x, StreamPool &) { apply_update(delta_x); }
  void apply_update(const T *delta_x) { for (auto *vd : vertex_descriptors) vd->apply_update(delta_x, jacobian_scales.raw()); detail::sync(); }
  void backup_parameters() { for (auto *vd : vertex_descriptors) vd->backup_parameters(); detail::sync(); }
  void revert_parameters() { for (auto *vd : vertex_descriptors) vd->restore_parameters(); detail::sync(); }
  // (J^T rho' P J) x, matrix-free (solver/pcg.hpp:143-163)
  void hessian_matvec(T *out, const T *x) {
    detail::fill<T>(out, hessian_dim, T(0));
    for (auto *fd : factor_descriptors) {
      if (!fd->compute_Jv_fresh(fd->work_residual(), x)) {
        detail::fill<T>(fd->work_residual(), fd->internal_count() * fd->error_dimension(), T(0));
        fd->compute_Jv(fd->work_residual(), x);
      }
      fd->compute_Jtv(out, fd->work_residual());
    }
  }
};

} // namespace graphite

#include "engine_model.hpp" // the engine's kernels on user traits: needs everything above
#include "engine_pose.hpp"  // the pose-graph engine on user traits
