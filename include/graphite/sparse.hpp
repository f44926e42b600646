// graphite/sparse.hpp — block-sparse Hessian, Schur complement and scalar CSC export of the generic layer.
//
// Same public types, layouts and call flow as the reference:
//   CSCMatrix<S, I>          hessian.hpp:15-20      d_pointers / d_indices / d_values (scalar CSC, upper triangle)
//   Hessian<T, S>            hessian.hpp:45-330     unique UPPER block coordinates sorted column-major (row inside a
//                                                   column), every block rows x cols COLUMN-major, the diagonal block
//                                                   stored FULL and last in its column (:123-126); block-CSC indices
//                                                   (csc_utils.hpp:16-50); backup_diagonal / apply_damping (:102-176)
//   SchurComplement<T, S>    schur.hpp:87-1120      S = Hpp - Hpl Hll^-1 Hpl^T over the upper pose blocks, b_S, the
//                                                   landmark back-substitution (:279-302), S x (:347-393), CSC export
// exercised the way tests/schur.cu:113-240 drives them.  What differs is how they are built: the reference walks hash
// maps of block coordinates on the host and fills per-dimension MulOp pointer tables (schur.hpp:484-585); here the
// structure is a sorted list of 64-bit block keys (col << 32 | row) BUILT ON THE DEVICE — every factor / landmark column
// emits its keys, a radix sort + unique (hipCUB) orders them column-major, exclusive scans give the value offsets and
// the operation lists, and every consumer finds its block by binary search — the products are plain index records and
// one generic kernel handles every (dim_a, dim_b, dim_c) combination.  The host keeps copies of the block-CSC arrays
// for the accessors only.  Correctness-first like the rest of the generic layer (one thread per output scalar, atomics
// where several factors / products meet in one block); bundle-adjustment graphs whose traits are the BAL model are
// optimised by the gr_bal engine instead (solve.hpp).
#pragma once
#include "core.hpp"
#include <hipcub/hipcub.hpp>

namespace graphite {

// ---- thrust::device_vector look-alike (hipMalloc storage; element reads copy one value to the host) ------------------
template <typename T> class device_vector {
  T *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
public:
  struct pointer { T *p; T *get() const { return p; } };
  device_vector() = default;
  explicit device_vector(size_t n) { resize(n); }
  device_vector(const device_vector &) = delete;
  device_vector &operator=(const device_vector &) = delete;
  ~device_vector() { if (p_) (void)hipFree(p_); }
  void resize(size_t n) {
    if (n > cap_) {
      T *q = nullptr;
      GRAPHITE_HIP(hipMalloc(reinterpret_cast<void **>(&q), n * sizeof(T)));
      if (n_) GRAPHITE_HIP(hipMemcpy(q, p_, n_ * sizeof(T), hipMemcpyDeviceToDevice));
      if (p_) (void)hipFree(p_);
      p_ = q; cap_ = n;
    }
    n_ = n;
  }
  void clear() { n_ = 0; }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  pointer data() const { return pointer{p_}; }
  T *raw() const { return p_; }
  device_vector &operator=(const std::vector<T> &h) {
    resize(h.size());
    if (!h.empty()) GRAPHITE_HIP(hipMemcpy(p_, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return *this;
  }
  std::vector<T> to_host() const {
    std::vector<T> h(n_);
    if (n_) GRAPHITE_HIP(hipMemcpy(h.data(), p_, n_ * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
  T operator[](size_t i) const { T v; GRAPHITE_HIP(hipMemcpy(&v, p_ + i, sizeof(T), hipMemcpyDeviceToHost)); return v; }
  void zero() { if (n_) GRAPHITE_HIP(hipMemset(p_, 0, n_ * sizeof(T))); }
};

// hessian.hpp:15-20
template <typename T, typename I> class CSCMatrix {
public:
  device_vector<I> d_pointers;
  device_vector<I> d_indices;
  device_vector<T> d_values;
};

namespace detail {
inline uint64_t block_key(size_t row, size_t col) { return pack_block_key(row, col); }

// ---- device symbolic phase: sorted unique keys, scans, lookups -------------------------------------------------------
struct ScratchBytes {
  void *p = nullptr; size_t cap = 0;
  ~ScratchBytes() { if (p) (void)hipFree(p); }
  void *get(size_t n) { if (n > cap) { if (p) (void)hipFree(p); GRAPHITE_HIP(hipMalloc(&p, n)); cap = n; } return p; }
};
// keys <- its sorted distinct valid (!= ~0) entries; returns how many
inline size_t sort_unique_keys(device_vector<uint64_t> &keys) {
  const size_t n = keys.size();
  if (!n) return 0;
  if (n > (size_t)std::numeric_limits<int>::max()) throw std::invalid_argument("block-sparse structure: more than 2^31 block keys");
  device_vector<uint64_t> tmp(n);
  device_vector<int> count(1);
  ScratchBytes scratch;
  size_t bytes = 0;
  GRAPHITE_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, keys.raw(), tmp.raw(), (int)n));
  GRAPHITE_HIP(hipcub::DeviceRadixSort::SortKeys(scratch.get(bytes), bytes, keys.raw(), tmp.raw(), (int)n));
  GRAPHITE_HIP(hipcub::DeviceSelect::Unique(nullptr, bytes, tmp.raw(), keys.raw(), count.raw(), (int)n));
  GRAPHITE_HIP(hipcub::DeviceSelect::Unique(scratch.get(bytes), bytes, tmp.raw(), keys.raw(), count.raw(), (int)n));
  size_t m = (size_t)count[0];
  if (m && keys[m - 1] == ~uint64_t(0)) --m; // the "no block" marker sorts last
  keys.resize(m);
  return m;
}
// out[i] = sum of in[0..i), out[n] = total (n + 1 entries); returns the total
inline size_t exclusive_scan(const device_vector<size_t> &in, device_vector<size_t> &out) {
  const size_t n = in.size();
  if (n > (size_t)std::numeric_limits<int>::max()) throw std::invalid_argument("block-sparse structure: more than 2^31 entries in a scan");
  out.resize(n + 1);
  if (!n) { out.zero(); return 0; }
  ScratchBytes scratch;
  size_t bytes = 0;
  GRAPHITE_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in.raw(), out.raw(), (int)n));
  GRAPHITE_HIP(hipcub::DeviceScan::ExclusiveSum(scratch.get(bytes), bytes, in.raw(), out.raw(), (int)n));
  const size_t total = out[n - 1] + in[n - 1];
  GRAPHITE_HIP(hipMemcpy(out.raw() + n, &total, sizeof(size_t), hipMemcpyHostToDevice));
  return total;
}
__global__ void k_scalar_to_block(size_t nb, const size_t *soff, size_t *s2b) {
  const size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (b >= nb) return;
  for (size_t c = soff[b]; c < soff[b + 1]; ++c) s2b[c] = b;
}
__global__ void k_diag_keys(size_t nb, uint64_t *keys) {
  const size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (b < nb) keys[b] = pack_block_key(b, b);
}
// sorted keys -> row index, block column and value count of every block
__global__ void k_unpack_keys(size_t n, const uint64_t *keys, const size_t *soff, size_t *rowi, size_t *bcol, size_t *count) {
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= n) return;
  const size_t r = (size_t)(keys[q] & 0xffffffffu), c = (size_t)(keys[q] >> 32);
  rowi[q] = r;
  if (bcol) bcol[q] = c;
  count[q] = (soff[r + 1] - soff[r]) * (soff[c + 1] - soff[c]);
}
// colp[b] = first key of block column >= b (b = 0 .. nb)
__global__ void k_col_pointers(size_t nb, const uint64_t *keys, size_t n, size_t *colp) {
  const size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (b <= nb) colp[b] = find_block_key(keys, n, (uint64_t)b << 32);
}
inline bool host_find_key(const std::vector<uint64_t> &keys, uint64_t key, size_t &pos) {
  const auto it = std::lower_bound(keys.begin(), keys.end(), key);
  pos = (size_t)(it - keys.begin());
  return it != keys.end() && *it == key;
}

// number of stored scalars of scalar column `col` (rows <= col), csc_utils.hpp:87-113
template <typename I> __global__ void k_csc_count(size_t dim, const size_t *s2b, const size_t *colp, const size_t *rowi, const size_t *soff, I *count) {
  const size_t col = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (col >= dim) return;
  const size_t bc = s2b[col];
  size_t nv = 0;
  for (size_t b = colp[bc]; b < colp[bc + 1]; ++b) {
    const size_t br = rowi[b], nr = soff[br + 1] - soff[br], r0 = soff[br];
    for (size_t r = 0; r < nr && r0 + r <= col; ++r) ++nv;
  }
  count[col] = (I)nv;
}
// row indices (values == nullptr) or values of the scalar upper CSC, csc_utils.hpp:123-193
template <typename S, typename I> __global__ void k_csc_fill(size_t dim, const size_t *s2b, const size_t *colp, const size_t *rowi, const size_t *boff, const size_t *soff,
                                                              const I *ptr, const S *values, I *out_idx, S *out_val) {
  const size_t col = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (col >= dim) return;
  const size_t bc = s2b[col], cin = col - soff[bc];
  size_t w = (size_t)ptr[col];
  for (size_t b = colp[bc]; b < colp[bc + 1]; ++b) {
    const size_t br = rowi[b], nr = soff[br + 1] - soff[br], r0 = soff[br];
    for (size_t r = 0; r < nr && r0 + r <= col; ++r, ++w) {
      if (out_idx) out_idx[w] = (I)(r0 + r);
      if (out_val) out_val[w] = values[boff[b] + cin * nr + r];
    }
  }
}
template <typename S, typename I>
void build_scalar_csc(size_t dim, const device_vector<size_t> &s2b, const device_vector<size_t> &colp, const device_vector<size_t> &rowi,
                      const device_vector<size_t> &boff, const device_vector<size_t> &soff, CSCMatrix<S, I> &m) {
  m.d_pointers.resize(dim + 1);
  m.d_pointers.zero();
  if (dim) k_csc_count<I><<<blocks(dim), TPB>>>(dim, s2b.raw(), colp.raw(), rowi.raw(), soff.raw(), m.d_pointers.raw());
  std::vector<I> cnt = m.d_pointers.to_host(); // exclusive scan (the reference: thrust::exclusive_scan)
  I run = 0;
  for (size_t c = 0; c <= dim; ++c) { const I v = cnt[c]; cnt[c] = run; run += v; }
  m.d_pointers = cnt;
  const size_t nnz = (size_t)cnt[dim];
  m.d_indices.resize(nnz); m.d_values.resize(nnz);
  if (dim) k_csc_fill<S, I><<<blocks(dim), TPB>>>(dim, s2b.raw(), colp.raw(), rowi.raw(), boff.raw(), soff.raw(), m.d_pointers.raw(), (const S *)nullptr, m.d_indices.raw(), (S *)nullptr);
  sync();
}
template <typename S, typename I>
void update_scalar_csc(size_t dim, const device_vector<S> &values, const device_vector<size_t> &s2b, const device_vector<size_t> &colp, const device_vector<size_t> &rowi,
                       const device_vector<size_t> &boff, const device_vector<size_t> &soff, CSCMatrix<S, I> &m) {
  if (dim) k_csc_fill<S, I><<<blocks(dim), TPB>>>(dim, s2b.raw(), colp.raw(), rowi.raw(), boff.raw(), soff.raw(), m.d_pointers.raw(), values.raw(), (I *)nullptr, m.d_values.raw());
  sync();
}

// prev_diag <- diagonal of the diagonal blocks (hessian.hpp:102-134); or H diag <- damped prev_diag (:136-176)
template <typename T, typename S, int MODE> __global__ void k_hessian_diag(size_t nb, const size_t *colp, const size_t *boff, const size_t *soff, S *values, S *prev, T mu, int identity) {
  const size_t bc = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (bc >= nb) return;
  const size_t b = colp[bc + 1] - 1; // the diagonal block is the last of its column
  const size_t d = soff[bc + 1] - soff[bc];
  S *blk = values + boff[b];
  for (size_t i = 0; i < d; ++i) {
    if (MODE == 0) prev[soff[bc] + i] = blk[i * d + i];
    else {
      const double h = (double)prev[soff[bc] + i];
      const double cl = h < 1.0e-6 ? 1.0e-6 : (h > 1.0e32 ? 1.0e32 : h);
      blk[i * d + i] = (S)(identity ? h + (double)mu : h + (double)mu * cl);
    }
  }
}
} // namespace detail

// =================================================================================================
// Hessian (hessian.hpp:45-330)
// =================================================================================================
template <typename T, typename S> class Hessian {
  std::vector<uint64_t> h_keys; // sorted (col, row) keys of the blocks: the host-side lookup
  std::vector<size_t> h_col_pointers, h_row_indices, h_offsets, h_scalar_offsets;
  device_vector<uint64_t> d_keys;
  device_vector<size_t> d_col_pointers, d_row_indices, d_block_col, d_offsets, d_hessian_offsets, scalar_to_block_map;
  device_vector<S> d_hessian, d_prev_diag;
public:
  Hessian() = default;
  const device_vector<size_t> &get_block_col_pointers() const { return d_col_pointers; }
  const device_vector<size_t> &get_block_row_indices() const { return d_row_indices; }
  const device_vector<size_t> &get_block_value_offsets() const { return d_offsets; }
  const device_vector<S> &get_values() const { return d_hessian; }
  S *get_values_ptr() { return d_hessian.raw(); }
  const S *get_values_ptr() const { return d_hessian.raw(); }
  // host copies of the block-CSC (accessors, tests)
  const std::vector<size_t> &host_col_pointers() const { return h_col_pointers; }
  const std::vector<size_t> &host_row_indices() const { return h_row_indices; }
  const std::vector<size_t> &host_value_offsets() const { return h_offsets; }
  const std::vector<size_t> &host_scalar_offsets() const { return h_scalar_offsets; }
  // ... and what SchurComplement::build_structure reads on the device
  const device_vector<size_t> &device_scalar_offsets() const { return d_hessian_offsets; }
  const device_vector<size_t> &device_block_columns() const { return d_block_col; }
  const device_vector<uint64_t> &device_block_keys() const { return d_keys; }
  bool has_block(size_t row, size_t col) const { size_t q; return detail::host_find_key(h_keys, detail::block_key(row, col), q); }
  size_t block_offset(size_t row, size_t col) const {
    size_t q;
    if (!detail::host_find_key(h_keys, detail::block_key(row, col), q)) throw std::out_of_range("Hessian: no such block");
    return h_offsets[q];
  }

  void build_structure(Graph<T, S> *graph, StreamPool &) {
    using namespace detail;
    h_scalar_offsets = graph->get_offset_vector();
    const size_t nb = graph->get_num_block_columns(), dim = graph->get_hessian_dimension();
    if (nb >= (size_t(1) << 32)) throw std::invalid_argument("Hessian: more than 2^32 block columns");
    d_hessian_offsets = h_scalar_offsets;
    scalar_to_block_map.resize(dim);
    if (nb) k_scalar_to_block<<<blocks(nb), TPB>>>(nb, d_hessian_offsets.raw(), scalar_to_block_map.raw());
    // keys of the upper blocks: every active vertex's diagonal block + one per vertex pair of an active factor
    size_t nkeys = nb;
    for (auto *fd : graph->get_factor_descriptors()) nkeys += fd->block_key_count();
    d_keys.resize(nkeys);
    if (nb) k_diag_keys<<<blocks(nb), TPB>>>(nb, d_keys.raw());
    size_t at = nb;
    for (auto *fd : graph->get_factor_descriptors()) { fd->emit_block_keys(d_keys.raw() + at, scalar_to_block_map.raw()); at += fd->block_key_count(); }
    const size_t nblk = sort_unique_keys(d_keys); // column-major, row inside a column (the diagonal block last: upper triangle)
    device_vector<size_t> counts(nblk);
    d_row_indices.resize(nblk); d_block_col.resize(nblk);
    if (nblk) k_unpack_keys<<<blocks(nblk), TPB>>>(nblk, d_keys.raw(), d_hessian_offsets.raw(), d_row_indices.raw(), d_block_col.raw(), counts.raw());
    const size_t nvalues = exclusive_scan(counts, d_offsets);
    d_offsets.resize(nblk);
    d_col_pointers.resize(nb + 1);
    k_col_pointers<<<blocks(nb + 1), TPB>>>(nb, d_keys.raw(), nblk, d_col_pointers.raw());
    d_hessian.resize(nvalues); d_prev_diag.resize(dim);
    // per factor and vertex pair: where its block lives (the role of setup_hessian_computation, hessian.hpp:178-208)
    for (auto *fd : graph->get_factor_descriptors()) fd->sparse_setup(d_keys.raw(), nblk, d_offsets.raw(), scalar_to_block_map.raw());
    sync();
    h_keys = d_keys.to_host(); h_col_pointers = d_col_pointers.to_host(); h_row_indices = d_row_indices.to_host(); h_offsets = d_offsets.to_host();
  }
  void update_values(Graph<T, S> *graph, StreamPool &) { // hessian.hpp:290-307
    d_hessian.zero();
    for (auto *fd : graph->get_factor_descriptors()) fd->sparse_hessian(d_hessian.raw());
    const size_t nb = graph->get_num_block_columns();
    if (nb) detail::k_hessian_diag<T, S, 0><<<detail::blocks(nb), detail::TPB>>>(nb, d_col_pointers.raw(), d_offsets.raw(), d_hessian_offsets.raw(), d_hessian.raw(), d_prev_diag.raw(), T(0), 0);
    detail::sync();
  }
  void apply_damping(Graph<T, S> *graph, T damping_factor, const bool use_identity, StreamPool &) { // hessian.hpp:136-176
    const size_t nb = graph->get_num_block_columns();
    if (nb) detail::k_hessian_diag<T, S, 1><<<detail::blocks(nb), detail::TPB>>>(nb, d_col_pointers.raw(), d_offsets.raw(), d_hessian_offsets.raw(), d_hessian.raw(), d_prev_diag.raw(), damping_factor, use_identity ? 1 : 0);
    detail::sync();
  }
  template <typename I> void build_csc_structure(Graph<T, S> *graph, CSCMatrix<S, I> &matrix) {
    detail::build_scalar_csc<S, I>(graph->get_hessian_dimension(), scalar_to_block_map, d_col_pointers, d_row_indices, d_offsets, d_hessian_offsets, matrix);
  }
  template <typename I> void update_csc_values(Graph<T, S> *graph, CSCMatrix<S, I> &matrix) {
    detail::update_scalar_csc<S, I>(graph->get_hessian_dimension(), d_hessian, scalar_to_block_map, d_col_pointers, d_row_indices, d_offsets, d_hessian_offsets, matrix);
  }
};

// =================================================================================================
// SchurComplement (schur.hpp:87-1120)
// =================================================================================================
namespace detail {
struct SchurMulOp { size_t dst, left, right, mid; uint32_t da, db, dl, pad; };       // S_dst -= L M R^T       (schur.hpp:484-585)
struct SchurHplOp { size_t blk, inv, prow, lrow; uint32_t da, dl; };                  // one Hpl block (pose rows prow.., landmark rows lrow..)
struct SchurCopyOp { size_t src, dst, count; };                                       // Hpp block -> S block  (:587)
struct SchurInvOp { size_t blk, inv; uint32_t d, pad; };                              // Hll block -> inverse  (:1067)
struct SchurVecOp { size_t blk, xoff, yoff; uint32_t rows, cols, transposed, pad; };  // y += A x or A^T x     (:307-393)

template <typename S> __global__ void k_schur_copy(const SchurCopyOp *ops, size_t nops, const S *H, S *Sv) {
  const size_t op = blockIdx.x;
  if (op >= nops) return;
  for (size_t i = threadIdx.x; i < ops[op].count; i += blockDim.x) Sv[ops[op].dst + i] = H[ops[op].src + i];
}
template <typename S> __global__ void __launch_bounds__(BLOCK_INV_THREADS) k_schur_invert(const SchurInvOp *ops, size_t nops, const S *H, S *inv, int max_d) {
  extern __shared__ double lds_inv[]; // [entry][thread] work matrices (core.hpp: lds_gauss_jordan), the role of cublas<t>matinvBatched (:1101)
  const int tid = threadIdx.x, nt = BLOCK_INV_THREADS;
  const size_t op = blockIdx.x * (size_t)nt + tid;
  if (op >= nops) return;
  const int d = (int)ops[op].d;
  double *A = lds_inv + tid, *R = lds_inv + (size_t)max_d * max_d * nt + tid;
  const S *B = H + ops[op].blk;
  for (int i = 0; i < d * d; ++i) { A[i * nt] = (double)B[i]; R[i * nt] = (i % d == i / d) ? 1.0 : 0.0; }
  lds_gauss_jordan(A, R, d, nt);
  for (int i = 0; i < d * d; ++i) inv[ops[op].inv + i] = (S)R[i * nt];
}
// S_dst -= sum over the products of that destination block of L M R^T (ops/schur.hpp:155-188 adds every product with one
// atomicAdd per output scalar).  The products are sorted by destination at build_structure (stable: landmark order) and cut
// into CHUNKS of <= SCHUR_MUL_CHUNK products of one destination; one workgroup per chunk: its threads form
// blockDim / (da db) slices, slice s takes products s, s + slices, ... with one thread per output scalar
// (value = sum_k L(row,k) sum_j M(k,j) R(col,j)), the slices are added through LDS in slice order.  A destination with a
// single chunk (almost all off-diagonal blocks) is finished there; the others (a camera's diagonal block collects one
// product per observation) leave one partial block per chunk and k_schur_mul_join adds them in chunk order — no atomics,
// no per-thread search, the same bits every run, and no workgroup walks more than SCHUR_MUL_CHUNK products.
constexpr int SCHUR_MUL_THREADS = 256, SCHUR_MUL_CHUNK = 64, SCHUR_MUL_STAGE = 1024;
struct SchurChunks { const unsigned *blk; const size_t *first_chunk; const size_t *first_product; double *partial; size_t stride; };
// max_dl: largest eliminated-vertex dimension of the graph (sizes the LDS share of a slice).  Every product's three blocks
// (L: da x dl, R: db x dl, M: dl x dl — contiguous in H / the inverse array) are staged through LDS by the slice that
// owns the product, one coalesced load per value, instead of every output thread fetching its 2 dl + dl^2 operands itself.
template <typename S> __global__ void __launch_bounds__(SCHUR_MUL_THREADS)
k_schur_mul(const SchurMulOp *ops, SchurChunks ch, const size_t *rowi, const size_t *bcol, const size_t *boff, const size_t *soff, const S *H, const S *inv, S *Sv, uint32_t max_dl) {
  __shared__ double part[SCHUR_MUL_THREADS];
  __shared__ S stage[SCHUR_MUL_STAGE];
  const size_t j = blockIdx.x, q = ch.blk[j], i = j - ch.first_chunk[q], nchunk = ch.first_chunk[q + 1] - ch.first_chunk[q];
  const size_t p0 = ch.first_product[q] + i * SCHUR_MUL_CHUNK, pe = ch.first_product[q + 1], p1 = p0 + SCHUR_MUL_CHUNK < pe ? p0 + SCHUR_MUL_CHUNK : pe;
  const size_t r = rowi[q], c = bcol[q];
  const uint32_t da = (uint32_t)(soff[r + 1] - soff[r]), db = (uint32_t)(soff[c + 1] - soff[c]), nout = da * db;
  const uint32_t cap = max_dl * (da + db + max_dl); // staged values of one product, at most
  // more outputs than threads (two pose blocks beyond 16 x 16): round trips
  for (uint32_t e0 = 0; e0 < nout; e0 += SCHUR_MUL_THREADS) {
    const uint32_t span = nout - e0 < SCHUR_MUL_THREADS ? nout - e0 : SCHUR_MUL_THREADS;
    uint32_t slices = SCHUR_MUL_THREADS / span;
    const bool staged = cap <= SCHUR_MUL_STAGE;
    if (staged && slices * cap > SCHUR_MUL_STAGE) slices = SCHUR_MUL_STAGE / cap;
    const uint32_t s = threadIdx.x / span, el = threadIdx.x % span, e = e0 + el, row = e % da, col = e / da;
    const uint32_t nit = (uint32_t)((p1 - p0 + slices - 1) / slices);
    double acc = 0;
    for (uint32_t it = 0; it < nit; ++it) { // block-uniform trip count: the barriers are reached by every thread
      const size_t p = p0 + (size_t)it * slices + s;
      const bool mine = s < slices && p < p1;
      SchurMulOp o{};
      if (mine) o = ops[p];
      const S *L = H + o.left, *R = H + o.right, *M = inv + o.mid;
      if (staged) {
        S *st = stage + s * cap;
        if (mine) {
          const uint32_t nl = da * o.dl, nr = db * o.dl, cnt = nl + nr + o.dl * o.dl;
          for (uint32_t v = el; v < cnt; v += span) st[v] = v < nl ? L[v] : v < nl + nr ? R[v - nl] : M[v - nl - nr];
          L = st; R = st + nl; M = st + nl + nr;
        }
        __syncthreads();
      }
      if (mine) {
        S value = 0;
        for (uint32_t k = 0; k < o.dl; ++k) {
          S mrt = 0;
          for (uint32_t jj = 0; jj < o.dl; ++jj) mrt += M[k + o.dl * jj] * R[col + db * jj];
          value += L[row + da * k] * mrt;
        }
        acc += (double)value;
      }
      if (staged) __syncthreads();
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (s == 0) {
      double tot = 0;
      for (uint32_t k = 0; k < slices; ++k) tot += part[k * span + threadIdx.x];
      if (nchunk == 1) Sv[boff[q] + e] -= (S)tot;
      else ch.partial[j * ch.stride + e] = tot;
    }
    __syncthreads();
  }
}
// The same reduction for graphs whose pose blocks all have dimension D and whose eliminated vertices all have dimension DL
// (bundle adjustment 9 / 3, SE(3) poses with 3-d landmarks 6 / 3, planar SLAM 3 / 2): compile-time dimensions, ONE WAVE per
// chunk, one lane per output COLUMN and G = 64 / D products side by side.  Lane (g, c) keeps column c of product group g's sum
// in D registers; the operands the D lanes of a group share (L: D x DL, M: DL x DL) are fetched once per group, the next
// round's while the current one is multiplied, and read back from the group's LDS strip as broadcasts (one wave: a wave-level fence,
// no workgroup barrier); R's column belongs to the lane.  The groups are then added in group order.  This is
// the engine's k_schur_products (graphite_amd/csrc/kernels.hpp) on the generic layer's operation records.
// Orders the LDS accesses of ONE wave whose lanes exchange data through LDS without a workgroup barrier: the hardware runs a
// wave's LDS instructions in issue order, but without this the compiler may move a lane's reads above another lane's writes
// (seen: garbage in the 3 / 2 instantiation below).
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <typename S, int D, int DL> __global__ void __launch_bounds__(64)
k_schur_mul_fixed(const SchurMulOp *ops, SchurChunks ch, const size_t *boff, const S *H, const S *inv, S *Sv) {
  constexpr int G = 64 / D, NL = D * DL, NS = NL + DL * DL, PER = (NS + D - 1) / D; // shared scalars per product, per lane
  __shared__ S strip[G][NS + 1];
  const size_t j = blockIdx.x, q = ch.blk[j], i = j - ch.first_chunk[q], nchunk = ch.first_chunk[q + 1] - ch.first_chunk[q];
  const size_t p0 = ch.first_product[q] + i * SCHUR_MUL_CHUNK, pe = ch.first_product[q + 1], p1 = p0 + SCHUR_MUL_CHUNK < pe ? p0 + SCHUR_MUL_CHUNK : pe;
  const int lane = threadIdx.x, g = lane / D, c = lane % D;
  double acc[D]; // products in S, their sum in double (as the any-dimension kernel)
#pragma unroll
  for (int r = 0; r < D; ++r) acc[r] = 0.0;
  if (g < G) {
    S *sg = strip[g];
    S sh_n[PER], rc_n[DL];
    auto fetch = [&](size_t p) {
      const SchurMulOp o = ops[p];
      const S *L = H + o.left, *R = H + o.right, *M = inv + o.mid;
#pragma unroll
      for (int u = 0; u < PER; ++u) { const int e = c + D * u; sh_n[u] = e < NL ? L[e] : (e < NS ? M[e - NL] : S(0)); }
#pragma unroll
      for (int jj = 0; jj < DL; ++jj) rc_n[jj] = R[c + D * jj];
    };
    size_t p = p0 + g;
    if (p < p1) fetch(p);
    for (; p < p1; p += G) {
#pragma unroll
      for (int u = 0; u < PER; ++u) { const int e = c + D * u; if (e < NS) sg[e] = sh_n[u]; }
      S rc[DL];
#pragma unroll
      for (int jj = 0; jj < DL; ++jj) rc[jj] = rc_n[jj];
      if (p + G < p1) fetch(p + G);
      wave_lds_fence(); // the strip is written and read by different lanes of this wave
      S u_k[DL]; // (M R^T)(k, c)
#pragma unroll
      for (int k = 0; k < DL; ++k) {
        S t = S(0);
#pragma unroll
        for (int jj = 0; jj < DL; ++jj) t += sg[NL + k + DL * jj] * rc[jj];
        u_k[k] = t;
      }
#pragma unroll
      for (int r = 0; r < D; ++r) {
        S t = S(0);
#pragma unroll
        for (int k = 0; k < DL; ++k) t += sg[r + D * k] * u_k[k];
        acc[r] += (double)t;
      }
      wave_lds_fence(); // ... and rewritten by the next round
    }
  }
#pragma unroll
  for (int r = 0; r < D; ++r) { // groups added in group order into lanes 0 .. D - 1
    double tot = acc[r];
#pragma unroll
    for (int gg = 1; gg < G; ++gg) tot += __shfl(acc[r], gg * D + c, 64);
    acc[r] = tot;
  }
  if (lane < D) {
#pragma unroll
    for (int r = 0; r < D; ++r) {
      if (nchunk == 1) Sv[boff[q] + r + D * c] -= (S)acc[r];
      else ch.partial[j * ch.stride + r + D * c] = acc[r];
    }
  }
}
template <typename S> __global__ void k_schur_mul_join(size_t nblk, SchurChunks ch, const size_t *rowi, const size_t *bcol, const size_t *boff, const size_t *soff, S *Sv) {
  const size_t q = blockIdx.x, c0 = ch.first_chunk[q], c1 = ch.first_chunk[q + 1];
  if (c1 - c0 < 2) return;
  const size_t r = rowi[q], c = bcol[q], nout = (soff[r + 1] - soff[r]) * (soff[c + 1] - soff[c]);
  for (size_t e = threadIdx.x; e < nout; e += blockDim.x) {
    double tot = 0;
    for (size_t j = c0; j < c1; ++j) tot += ch.partial[j * ch.stride + e];
    Sv[boff[q] + e] -= (S)tot;
  }
}
__global__ void k_schur_chunk_count(size_t nblk, const size_t *first_product, size_t *count) {
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q < nblk) count[q] = (first_product[q + 1] - first_product[q] + SCHUR_MUL_CHUNK - 1) / SCHUR_MUL_CHUNK;
}
__global__ void k_schur_chunk_fill(size_t nblk, const size_t *first_chunk, unsigned *blk) {
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= nblk) return;
  for (size_t j = first_chunk[q]; j < first_chunk[q + 1]; ++j) blk[j] = (unsigned)q;
}
// w_l = Hll^-1 v_l for every landmark block (v = b_l, or b_l - Hpl^T x_p)
template <typename T, typename S> __global__ void k_schur_apply_inverse(const SchurInvOp *ops, size_t nops, const size_t *lrow, const S *inv, const T *v, T *w) {
  const size_t op = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (op >= nops) return;
  const int d = (int)ops[op].d;
  const S *M = inv + ops[op].inv;
  const size_t r0 = lrow[op];
  for (int r = 0; r < d; ++r) {
    T s = 0;
    for (int c = 0; c < d; ++c) s += (T)M[r + d * c] * v[r0 + c];
    w[r0 + r] = s;
  }
}
// out_p[prow..] -= Hpl w_l (b_S, schur.hpp:901-920) and out_l[lrow..] -= Hpl^T x_p (back-substitution, :279-302).  The reference
// adds every block's product with atomics; here every output row has ONE writer and a fixed order of summation:
//   landmarks: the Hpl blocks of an eliminated vertex are consecutive records -> one thread per vertex walks them;
//   poses    : the records are sorted by pose block at build_structure (pose_idx, pose_first) -> one wave per pose block, lanes
//              stride over its records (a camera meets hundreds of points), then a butterfly.
// D, DL != 0: every pose block D x D and every eliminated vertex of dimension DL (loops unrolled at compile time); 0: run-time dimensions <= 16
template <typename T, typename S, int D = 0, int DL = 0> __global__ void k_schur_hpl_landmarks(size_t nl, const size_t *lm_first, const SchurHplOp *ops, const S *H, const T *xp, T *out_l) {
  const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (j >= nl) return;
  if constexpr (D != 0) {
    T s[DL];
#pragma unroll
    for (int c = 0; c < DL; ++c) s[c] = T(0);
    const size_t t0 = lm_first[j], t1 = lm_first[j + 1];
    if (t0 == t1) return;
    for (size_t t = t0; t < t1; ++t) {
      const SchurHplOp o = ops[t];
      const S *A = H + o.blk;
      T x[D];
#pragma unroll
      for (int r = 0; r < D; ++r) x[r] = xp[o.prow + r];
#pragma unroll
      for (int c = 0; c < DL; ++c) {
        T q = 0;
#pragma unroll
        for (int r = 0; r < D; ++r) q += (T)A[r + D * c] * x[r];
        s[c] += q;
      }
    }
    const size_t lrow = ops[t0].lrow;
#pragma unroll
    for (int c = 0; c < DL; ++c) out_l[lrow + c] -= s[c];
    return;
  }
  constexpr int MAXD = 16;
  T s[MAXD];
#pragma unroll
  for (int c = 0; c < MAXD; ++c) s[c] = T(0);
  uint32_t dl = 0;
  size_t lrow = 0;
  for (size_t t = lm_first[j]; t < lm_first[j + 1]; ++t) {
    const SchurHplOp o = ops[t];
    const S *A = H + o.blk; // da x dl, column-major
    dl = o.dl; lrow = o.lrow;
#pragma unroll
    for (int c = 0; c < MAXD; ++c)
      if ((uint32_t)c < o.dl) {
        T q = 0;
        for (uint32_t r = 0; r < o.da; ++r) q += (T)A[r + o.da * c] * xp[o.prow + r];
        s[c] += q;
      }
  }
#pragma unroll
  for (int c = 0; c < MAXD; ++c)
    if ((uint32_t)c < dl) out_l[lrow + c] -= s[c];
}
template <typename T, typename S, int D = 0, int DL = 0> __global__ void __launch_bounds__(64)
k_schur_hpl_poses(const size_t *pose_first, const unsigned *pose_idx, const SchurHplOp *ops, const S *H, const T *wl, T *out_p) {
  const size_t b = blockIdx.x, t0 = pose_first[b], t1 = pose_first[b + 1];
  if (t0 == t1) return;
  if constexpr (D != 0) {
    T s[D];
#pragma unroll
    for (int r = 0; r < D; ++r) s[r] = T(0);
    for (size_t t = t0 + threadIdx.x; t < t1; t += 64) {
      const SchurHplOp o = ops[pose_idx[t]];
      const S *A = H + o.blk;
      T w[DL];
#pragma unroll
      for (int c = 0; c < DL; ++c) w[c] = wl[o.lrow + c];
#pragma unroll
      for (int r = 0; r < D; ++r) {
        T q = 0;
#pragma unroll
        for (int c = 0; c < DL; ++c) q += (T)A[r + D * c] * w[c];
        s[r] += q;
      }
    }
    const size_t prow = ops[pose_idx[t0]].prow;
#pragma unroll
    for (int r = 0; r < D; ++r) {
      T v = s[r];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (threadIdx.x == 0) out_p[prow + r] -= v;
    }
    return;
  }
  constexpr int MAXD = 16;
  T s[MAXD];
#pragma unroll
  for (int r = 0; r < MAXD; ++r) s[r] = T(0);
  uint32_t da = 0;
  size_t prow = 0;
  for (size_t t = t0 + threadIdx.x; t < t1; t += 64) {
    const SchurHplOp o = ops[pose_idx[t]];
    const S *A = H + o.blk;
#pragma unroll
    for (int r = 0; r < MAXD; ++r)
      if ((uint32_t)r < o.da) {
        T q = 0;
        for (uint32_t c = 0; c < o.dl; ++c) q += (T)A[r + o.da * c] * wl[o.lrow + c];
        s[r] += q;
      }
  }
  { const SchurHplOp o = ops[pose_idx[t0]]; da = o.da; prow = o.prow; } // the same for every record of the block
#pragma unroll
  for (int r = 0; r < MAXD; ++r) {
    if ((uint32_t)r >= da) break; // block-uniform
    T v = s[r];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (threadIdx.x == 0) out_p[prow + r] -= v;
  }
}
// y = S x (schur.hpp:347-393 adds one atomic per block row).  The block records (upper blocks and their transposes) are sorted by
// OUTPUT block row at build_structure: one wave per block row, lanes stride over its records, butterfly, one plain store per
// row.  D != 0: every pose block D x D (compile-time loops); 0: run-time dimensions <= 16.
template <typename T, typename S, int D = 0> __global__ void __launch_bounds__(64)
k_schur_vec(const size_t *row_first, const unsigned *row_idx, const SchurVecOp *ops, const S *Sv, const T *x, T *y) {
  const size_t b = blockIdx.x, t0 = row_first[b], t1 = row_first[b + 1];
  if (t0 == t1) return;
  constexpr int MAXD = D != 0 ? D : 16;
  T s[MAXD];
#pragma unroll
  for (int r = 0; r < MAXD; ++r) s[r] = T(0);
  for (size_t t = t0 + threadIdx.x; t < t1; t += 64) {
    const SchurVecOp o = ops[row_idx[t]];
    const S *A = Sv + o.blk; // rows x cols, column-major
    if constexpr (D != 0) {
      T xv[D];
#pragma unroll
      for (int c = 0; c < D; ++c) xv[c] = x[o.xoff + c];
#pragma unroll
      for (int r = 0; r < D; ++r) {
        T q = 0;
#pragma unroll
        for (int c = 0; c < D; ++c) q += (T)(o.transposed ? A[c + D * r] : A[r + D * c]) * xv[c];
        s[r] += q;
      }
    } else {
      const uint32_t nout = o.transposed ? o.cols : o.rows, nin = o.transposed ? o.rows : o.cols;
#pragma unroll
      for (int r = 0; r < MAXD; ++r)
        if ((uint32_t)r < nout) {
          T q = 0;
          for (uint32_t c = 0; c < nin; ++c) q += (T)(o.transposed ? A[c + o.rows * r] : A[r + o.rows * c]) * x[o.xoff + c];
          s[r] += q;
        }
    }
  }
  const SchurVecOp o0 = ops[row_idx[t0]];
  const uint32_t nout = D != 0 ? (uint32_t)D : (o0.transposed ? o0.cols : o0.rows);
#pragma unroll
  for (int r = 0; r < MAXD; ++r) {
    if ((uint32_t)r >= nout) break; // block-uniform
    T v = s[r];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (threadIdx.x == 0) y[o0.yoff + r] = v;
  }
}
template <typename T> __global__ void k_sub(T *out, const T *a, const T *b, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] - b[i];
}
// dense symmetric image of the upper block-CSC (for the direct solve of the reduced system)
template <typename T, typename S> __global__ void k_blocks_to_dense(size_t nblk, const size_t *colp_of_blk, const size_t *rowi, const size_t *boff, const size_t *soff, const S *Sv, T *D, size_t ld) {
  const size_t b = blockIdx.x;
  if (b >= nblk) return;
  const size_t bc = colp_of_blk[b], br = rowi[b];
  const size_t nr = soff[br + 1] - soff[br], nc = soff[bc + 1] - soff[bc];
  for (size_t e = threadIdx.x; e < nr * nc; e += blockDim.x) {
    const size_t r = e % nr, c = e / nr;
    const T v = (T)Sv[boff[b] + e];
    D[(soff[br] + r) * ld + soff[bc] + c] = v;
    D[(soff[bc] + c) * ld + soff[br] + r] = v;
  }
}

// ---- symbolic phase of S on the device (schur.hpp:397-585 walks hash maps on the host) ----------------------------------
// View of H's block-CSC: colp / rowi / bcol / boff per block, soff = scalar start of every block column, L = first
// eliminated (landmark) column.  The blocks of a landmark column l are its pose rows (ascending) and, last, Hll.
struct HView { const size_t *colp, *rowi, *bcol, *boff, *soff; size_t L, nb, pose_dim; };
hd_fn inline size_t hv_dim(const HView &h, size_t b) { return h.soff[b + 1] - h.soff[b]; }
// per landmark column: number of (a <= bq) pose-row pairs; flags: 1 = an eliminated vertex shares a factor with another, 2 = dimension > 16
__global__ void k_schur_pair_counts(HView h, size_t *pairs, size_t *inv_size, int *flags) {
  const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x, l = h.L + j;
  if (l >= h.nb) return;
  const size_t k0 = h.colp[l], k1 = h.colp[l + 1] - 1, k = k1 - k0, dl = hv_dim(h, l);
  if (k && h.rowi[k1 - 1] >= h.L) atomicOr(flags, 1); // rows ascend: the last pose row is the largest
  if (dl > 16) atomicOr(flags, 2);
  pairs[j] = k * (k + 1) / 2;
  inv_size[j] = dl * dl;
}
// one thread per Hpl block q (pose row a of landmark column l): its Hpl record, and the products (a, bq >= a) it starts —
// S key and product record (dst filled in by k_schur_mul_dst once S is known)
__global__ void k_schur_emit(HView h, size_t nhpl, const size_t *pair_start, const size_t *inv_off, SchurHplOp *hpl, uint64_t *keys, SchurMulOp *mul, unsigned *hpl_pose) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= nhpl) return;
  // Hpl blocks and Hll blocks interleave in H's list: block q = colp[L] + t + (landmark columns before it); find the column by bisection
  size_t lo = h.L, hi = h.nb; // largest l with colp[l] - (l - L) <= colp[L] + t
  const size_t base = h.colp[h.L];
  while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (h.colp[mid] - (mid - h.L) <= base + t) lo = mid; else hi = mid; }
  const size_t l = lo, q = base + t + (l - h.L), k0 = h.colp[l], k1 = h.colp[l + 1] - 1, a = q - k0, k = k1 - k0, dl = hv_dim(h, l);
  const size_t ra = h.rowi[q], da = hv_dim(h, ra), io = inv_off[l - h.L];
  hpl[t] = SchurHplOp{h.boff[q], io, h.soff[ra], h.soff[l] - h.pose_dim, (uint32_t)da, (uint32_t)dl};
  hpl_pose[t] = (unsigned)ra;
  size_t p = pair_start[l - h.L] + a * k - (a * (a - 1)) / 2;
  for (size_t bq = q; bq < k1; ++bq, ++p) {
    const size_t rb = h.rowi[bq], db = hv_dim(h, rb);
    keys[p] = pack_block_key(ra, rb);
    mul[p] = SchurMulOp{0, h.boff[q], h.boff[bq], io, (uint32_t)da, (uint32_t)db, (uint32_t)dl, 0};
  }
}
__global__ void k_schur_mul_dst(size_t n, const uint64_t *pair_keys, const uint64_t *skeys, size_t ns, const size_t *soffs, SchurMulOp *mul, unsigned *blk, unsigned *perm) {
  const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t q = find_block_key(skeys, ns, pair_keys[p]);
  mul[p].dst = soffs[q];
  blk[p] = (unsigned)q; perm[p] = (unsigned)p;
}
__global__ void k_schur_permute(size_t n, const SchurMulOp *in, const unsigned *perm, SchurMulOp *out) {
  const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p < n) out[p] = in[perm[p]];
}
__global__ void k_iota(size_t n, unsigned *v) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) v[i] = (unsigned)i;
}
// first Hpl record of eliminated vertex j (records are emitted column by column, the Hll block of every column skipped)
__global__ void k_hpl_landmark_first(HView h, size_t nl, size_t *first) {
  const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (j <= nl) first[j] = h.colp[h.L + j] - h.colp[h.L] - j;
}
// first[q] = first sorted product whose block index is >= q (q = 0 .. nblk)
__global__ void k_first_of_block(size_t nblk, const unsigned *blk_sorted, size_t n, size_t *first) {
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q > nblk) return;
  size_t lo = 0, hi = n;
  while (lo < hi) { const size_t mid = (lo + hi) / 2; if (blk_sorted[mid] < q) lo = mid + 1; else hi = mid; }
  first[q] = lo;
}
__global__ void k_schur_copy_ops(size_t n, const uint64_t *hkeys, const size_t *hboff, HView h, const uint64_t *skeys, size_t ns, const size_t *soffs, SchurCopyOp *ops) {
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= n) return;
  ops[q] = SchurCopyOp{hboff[q], soffs[find_block_key(skeys, ns, hkeys[q])], hv_dim(h, h.rowi[q]) * hv_dim(h, h.bcol[q])};
}
__global__ void k_schur_inv_ops(HView h, const size_t *inv_off, SchurInvOp *ops, size_t *lrow) {
  const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x, l = h.L + j;
  if (l >= h.nb) return;
  ops[j] = SchurInvOp{h.boff[h.colp[l + 1] - 1], inv_off[j], (uint32_t)hv_dim(h, l), 0};
  lrow[j] = h.soff[l] - h.pose_dim;
}
// S x: one record per upper block and one for its transpose below the diagonal (schur.hpp:307-345); COUNT: records per block
template <bool COUNT> __global__ void k_schur_vec_ops(size_t ns, const size_t *rowi, const size_t *bcol, const size_t *boff, const size_t *soff, const size_t *first, size_t *count, SchurVecOp *ops, unsigned *out_blk) {
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= ns) return;
  const size_t r = rowi[q], c = bcol[q];
  if (COUNT) { count[q] = r == c ? 1 : 2; return; }
  const uint32_t dr = (uint32_t)(soff[r + 1] - soff[r]), dc = (uint32_t)(soff[c + 1] - soff[c]);
  ops[first[q]] = SchurVecOp{boff[q], soff[c], soff[r], dr, dc, 0, 0};
  out_blk[first[q]] = (unsigned)r;
  if (r != c) { ops[first[q] + 1] = SchurVecOp{boff[q], soff[r], soff[c], dr, dc, 1, 0}; out_blk[first[q] + 1] = (unsigned)c; }
}
__global__ void k_schur_diag_offsets(size_t L, const size_t *colp, const size_t *boff, size_t *diag) {
  const size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (b < L) diag[b] = colp[b + 1] > colp[b] ? boff[colp[b + 1] - 1] : 0; // upper triangle: the diagonal block closes its column
}
} // namespace detail

template <typename T, typename S> class SchurComplement {
  Hessian<T, S> &H;
  // structure of S: upper pose blocks, column-major sorted, blocks column-major (same layout as H)
  std::vector<size_t> h_col_pointers, h_row_indices, h_pose_offsets, h_diag_offsets;
  device_vector<uint64_t> d_keys;
  device_vector<size_t> d_col_pointers, d_row_indices, d_offsets, d_block_col, d_schur_offsets, scalar_to_block_map, d_mul_first, d_inv_lrow;
  device_vector<S> d_schur, d_hll_inv;
  device_vector<T> b_Schur, l_workspace, l_rhs;
  device_vector<detail::SchurMulOp> d_mul_ops;
  device_vector<detail::SchurHplOp> d_hpl_ops;
  device_vector<detail::SchurCopyOp> d_copy_ops;
  device_vector<detail::SchurInvOp> d_inv_ops;
  device_vector<detail::SchurVecOp> d_vec_ops;
  device_vector<size_t> d_vec_first;   // S x: first (sorted) block record of every output block row (+ end)
  device_vector<unsigned> d_vec_idx;   // ... and the records sorted by output block row
  device_vector<size_t> d_hpl_lm_first, d_hpl_pose_first; // first Hpl record of every eliminated vertex / (sorted) of every pose block (+ end)
  device_vector<unsigned> d_hpl_pose_idx;                 // Hpl records sorted by pose block
  device_vector<unsigned> d_chunk_blk;   // destination block of every product chunk
  device_vector<size_t> d_chunk_first;   // first chunk of every S block (+ end)
  device_vector<double> d_mul_partial;   // [chunk][chunk_stride]: partial blocks of the destinations with several chunks
  size_t num_chunks = 0, chunk_stride = 0, max_landmark_dim = 1;
  size_t uniform_pose_dim = 0, uniform_landmark_dim = 0; // the dimension every pose / eliminated block shares, 0 = mixed
  size_t landmark_col_start = 0, num_block_columns = 0, pose_dim = 0, landmark_dim = 0, num_blocks = 0;
public:
  explicit SchurComplement(Hessian<T, S> &H_) : H(H_) {}
  size_t get_pose_dimension() const { return pose_dim; }
  size_t get_landmark_dimension() const { return landmark_dim; }
  size_t num_pose_blocks() const { return landmark_col_start; }
  const std::vector<size_t> &host_pose_offsets() const { return h_pose_offsets; }   // scalar start of every pose block (+ end)
  const std::vector<size_t> &host_diag_offsets() const { return h_diag_offsets; }   // value offset of S(b, b)
  const std::vector<size_t> &host_col_pointers() const { return h_col_pointers; }
  const std::vector<size_t> &host_row_indices() const { return h_row_indices; }
  const device_vector<S> &get_values() const { return d_schur; }
  S *get_values_ptr() { return d_schur.raw(); }
  device_vector<T> &get_b_Schur() { return b_Schur; }

  void build_structure(Graph<T, S> *graph, StreamPool &) { // schur.hpp:194-225
    using namespace detail;
    num_block_columns = graph->get_num_block_columns();
    landmark_col_start = graph->get_elimination_block_column();
    const size_t L = landmark_col_start, nb = num_block_columns, nl = nb - L;
    const auto &soff = H.host_scalar_offsets();
    const auto dim_of = [&](size_t b) { return soff[b + 1] - soff[b]; };
    pose_dim = soff[L];
    landmark_dim = soff[nb] - pose_dim;
    const HView hv{H.get_block_col_pointers().raw(), H.get_block_row_indices().raw(), H.device_block_columns().raw(), H.get_block_value_offsets().raw(),
                   H.device_scalar_offsets().raw(), L, nb, pose_dim};
    const size_t n_hpp = H.host_col_pointers()[L];                                    // blocks of the pose columns
    const size_t n_hpl = H.host_col_pointers()[nb] - n_hpp - nl;                      // landmark columns minus their Hll blocks
    // symbolic S: Hpp pattern U {(i, j): i <= j both meet a landmark} (schur.hpp:397-476)
    device_vector<size_t> pairs(nl), pair_start, inv_size(nl), inv_off;
    device_vector<int> flags(1);
    flags.zero();
    if (nl) k_schur_pair_counts<<<blocks(nl), TPB>>>(hv, pairs.raw(), inv_size.raw(), flags.raw());
    const size_t npairs = exclusive_scan(pairs, pair_start), inv_values = exclusive_scan(inv_size, inv_off);
    const int bad = flags[0];
    if (bad & 1) throw std::invalid_argument("SchurComplement: eliminated vertices share a factor (Hll must be block diagonal)");
    if (bad & 2) throw std::invalid_argument("SchurComplement: eliminated vertex dimension > 16");
    d_keys.resize(n_hpp + npairs);
    if (n_hpp) GRAPHITE_HIP(hipMemcpy(d_keys.raw(), H.device_block_keys().raw(), n_hpp * sizeof(uint64_t), hipMemcpyDeviceToDevice));
    device_vector<uint64_t> pair_keys(npairs);
    d_hpl_ops.resize(n_hpl); d_mul_ops.resize(npairs);
    device_vector<unsigned> hpl_pose(n_hpl);
    if (n_hpl) k_schur_emit<<<blocks(n_hpl), TPB>>>(hv, n_hpl, pair_start.raw(), inv_off.raw(), d_hpl_ops.raw(), pair_keys.raw(), d_mul_ops.raw(), hpl_pose.raw());
    { // Hpl records by eliminated vertex (consecutive as emitted) and by pose block (stable sort): the two walks of k_schur_hpl_*
      d_hpl_lm_first.resize(nl + 1);
      k_hpl_landmark_first<<<blocks(nl + 1), TPB>>>(hv, nl, d_hpl_lm_first.raw());
      device_vector<unsigned> idx(n_hpl), pose_sorted(n_hpl);
      d_hpl_pose_idx.resize(n_hpl);
      if (n_hpl) {
        k_iota<<<blocks(n_hpl), TPB>>>(n_hpl, idx.raw());
        ScratchBytes scratch;
        size_t bytes = 0;
        GRAPHITE_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, hpl_pose.raw(), pose_sorted.raw(), idx.raw(), d_hpl_pose_idx.raw(), (int)n_hpl));
        GRAPHITE_HIP(hipcub::DeviceRadixSort::SortPairs(scratch.get(bytes), bytes, hpl_pose.raw(), pose_sorted.raw(), idx.raw(), d_hpl_pose_idx.raw(), (int)n_hpl));
      }
      d_hpl_pose_first.resize(L + 1);
      k_first_of_block<<<blocks(L + 1), TPB>>>(L, pose_sorted.raw(), n_hpl, d_hpl_pose_first.raw());
    }
    if (npairs) GRAPHITE_HIP(hipMemcpy(d_keys.raw() + n_hpp, pair_keys.raw(), npairs * sizeof(uint64_t), hipMemcpyDeviceToDevice));
    num_blocks = sort_unique_keys(d_keys);
    d_schur_offsets.resize(L + 1);
    if (L + 1) GRAPHITE_HIP(hipMemcpy(d_schur_offsets.raw(), H.device_scalar_offsets().raw(), (L + 1) * sizeof(size_t), hipMemcpyDeviceToDevice));
    device_vector<size_t> counts(num_blocks);
    d_row_indices.resize(num_blocks); d_block_col.resize(num_blocks);
    if (num_blocks) k_unpack_keys<<<blocks(num_blocks), TPB>>>(num_blocks, d_keys.raw(), d_schur_offsets.raw(), d_row_indices.raw(), d_block_col.raw(), counts.raw());
    const size_t nvalues = exclusive_scan(counts, d_offsets);
    d_offsets.resize(num_blocks);
    d_col_pointers.resize(L + 1);
    k_col_pointers<<<blocks(L + 1), TPB>>>(L, d_keys.raw(), num_blocks, d_col_pointers.raw());
    scalar_to_block_map.resize(pose_dim);
    if (L) k_scalar_to_block<<<blocks(L), TPB>>>(L, d_schur_offsets.raw(), scalar_to_block_map.raw());
    d_schur.resize(nvalues);
    // operation lists
    d_copy_ops.resize(n_hpp);
    if (n_hpp) k_schur_copy_ops<<<blocks(n_hpp), TPB>>>(n_hpp, H.device_block_keys().raw(), H.get_block_value_offsets().raw(), hv, d_keys.raw(), num_blocks, d_offsets.raw(), d_copy_ops.raw()); // Hpp copy (:587)
    d_inv_ops.resize(nl); d_inv_lrow.resize(nl);
    if (nl) k_schur_inv_ops<<<blocks(nl), TPB>>>(hv, inv_off.raw(), d_inv_ops.raw(), d_inv_lrow.raw());
    // products grouped by destination block (stable sort on the block index: landmark order inside a block), d_mul_first[q] = first product of S block q
    {
      device_vector<unsigned> blk(npairs), blk_sorted(npairs), perm(npairs), perm_sorted(npairs);
      device_vector<SchurMulOp> unsorted(npairs);
      if (npairs) {
        k_schur_mul_dst<<<blocks(npairs), TPB>>>(npairs, pair_keys.raw(), d_keys.raw(), num_blocks, d_offsets.raw(), d_mul_ops.raw(), blk.raw(), perm.raw());
        ScratchBytes scratch;
        size_t bytes = 0;
        GRAPHITE_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, blk.raw(), blk_sorted.raw(), perm.raw(), perm_sorted.raw(), (int)npairs));
        GRAPHITE_HIP(hipcub::DeviceRadixSort::SortPairs(scratch.get(bytes), bytes, blk.raw(), blk_sorted.raw(), perm.raw(), perm_sorted.raw(), (int)npairs));
        GRAPHITE_HIP(hipMemcpy(unsorted.raw(), d_mul_ops.raw(), npairs * sizeof(SchurMulOp), hipMemcpyDeviceToDevice));
        k_schur_permute<<<blocks(npairs), TPB>>>(npairs, unsorted.raw(), perm_sorted.raw(), d_mul_ops.raw());
      }
      d_mul_first.resize(num_blocks + 1);
      k_first_of_block<<<blocks(num_blocks + 1), TPB>>>(num_blocks, blk_sorted.raw(), npairs, d_mul_first.raw());
      // chunks of <= SCHUR_MUL_CHUNK products of one destination: the workgroups of k_schur_mul
      device_vector<size_t> nchunk(num_blocks);
      if (num_blocks) k_schur_chunk_count<<<blocks(num_blocks), TPB>>>(num_blocks, d_mul_first.raw(), nchunk.raw());
      num_chunks = exclusive_scan(nchunk, d_chunk_first);
      d_chunk_blk.resize(num_chunks);
      if (num_blocks) k_schur_chunk_fill<<<blocks(num_blocks), TPB>>>(num_blocks, d_chunk_first.raw(), d_chunk_blk.raw());
      chunk_stride = 0;
      for (size_t b = 0; b < L; ++b) chunk_stride = std::max(chunk_stride, dim_of(b));
      chunk_stride *= chunk_stride; // the largest S block
      max_landmark_dim = 1;
      for (size_t b = L; b < nb; ++b) max_landmark_dim = std::max(max_landmark_dim, dim_of(b));
      uniform_pose_dim = L ? dim_of(0) : 0; uniform_landmark_dim = nb > L ? dim_of(L) : 0;
      for (size_t b = 0; b < L; ++b) if (dim_of(b) != uniform_pose_dim) uniform_pose_dim = 0;
      for (size_t b = L; b < nb; ++b) if (dim_of(b) != uniform_landmark_dim) uniform_landmark_dim = 0;
      if (getenv("GRAPHITE_SCHUR_MUL_GENERIC") && atoi(getenv("GRAPHITE_SCHUR_MUL_GENERIC")) != 0) uniform_pose_dim = 0; // A/B: the any-dimension kernel
      d_mul_partial.resize(num_chunks * chunk_stride);
    }
    device_vector<size_t> vec_count(num_blocks), vec_first;
    if (num_blocks) k_schur_vec_ops<true><<<blocks(num_blocks), TPB>>>(num_blocks, d_row_indices.raw(), d_block_col.raw(), d_offsets.raw(), d_schur_offsets.raw(), nullptr, vec_count.raw(), nullptr, nullptr);
    const size_t nvec = exclusive_scan(vec_count, vec_first);
    d_vec_ops.resize(nvec);
    {
      device_vector<unsigned> out_blk(nvec), out_sorted(nvec), idx(nvec);
      if (num_blocks) k_schur_vec_ops<false><<<blocks(num_blocks), TPB>>>(num_blocks, d_row_indices.raw(), d_block_col.raw(), d_offsets.raw(), d_schur_offsets.raw(), vec_first.raw(), nullptr, d_vec_ops.raw(), out_blk.raw());
      d_vec_idx.resize(nvec);
      if (nvec) { // records by output block row (stable: column order inside a row)
        k_iota<<<blocks(nvec), TPB>>>(nvec, idx.raw());
        ScratchBytes scratch;
        size_t bytes = 0;
        GRAPHITE_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, out_blk.raw(), out_sorted.raw(), idx.raw(), d_vec_idx.raw(), (int)nvec));
        GRAPHITE_HIP(hipcub::DeviceRadixSort::SortPairs(scratch.get(bytes), bytes, out_blk.raw(), out_sorted.raw(), idx.raw(), d_vec_idx.raw(), (int)nvec));
      }
      d_vec_first.resize(L + 1);
      k_first_of_block<<<blocks(L + 1), TPB>>>(L, out_sorted.raw(), nvec, d_vec_first.raw());
    }
    device_vector<size_t> diag(L);
    if (L) k_schur_diag_offsets<<<blocks(L), TPB>>>(L, d_col_pointers.raw(), d_offsets.raw(), diag.raw());
    sync();
    h_col_pointers = d_col_pointers.to_host(); h_row_indices = d_row_indices.to_host(); h_diag_offsets = diag.to_host();
    h_pose_offsets.assign(soff.begin(), soff.begin() + L + 1);
    d_hll_inv.resize(inv_values);
    b_Schur.resize(pose_dim); l_workspace.resize(landmark_dim); l_rhs.resize(landmark_dim);
  }

  void update_values(Graph<T, S> *graph, StreamPool &) { // schur.hpp:227-235
    using namespace detail;
    d_schur.zero();
    if (d_copy_ops.size()) k_schur_copy<S><<<(unsigned)d_copy_ops.size(), 64>>>(d_copy_ops.raw(), d_copy_ops.size(), H.get_values_ptr(), d_schur.raw());
    if (d_inv_ops.size()) {
      allow_block_inverse_lds(reinterpret_cast<const void *>(&k_schur_invert<S>));
      k_schur_invert<S><<<(unsigned)((d_inv_ops.size() + BLOCK_INV_THREADS - 1) / BLOCK_INV_THREADS), BLOCK_INV_THREADS, block_inverse_lds_bytes(max_landmark_dim)>>>(
          d_inv_ops.raw(), d_inv_ops.size(), H.get_values_ptr(), d_hll_inv.raw(), (int)max_landmark_dim);
      launch_check();
    }
    if (num_chunks) {
      const SchurChunks ch{d_chunk_blk.raw(), d_chunk_first.raw(), d_mul_first.raw(), d_mul_partial.raw(), chunk_stride};
      // uniform dimensions with a compiled fast form: 9 / 3 (bundle adjustment), 6 / 3 (SE(3) + 3-d landmarks), 3 / 2 (planar SLAM)
      const unsigned nc = (unsigned)num_chunks;
      if (uniform_pose_dim == 9 && uniform_landmark_dim == 3) k_schur_mul_fixed<S, 9, 3><<<nc, 64>>>(d_mul_ops.raw(), ch, d_offsets.raw(), H.get_values_ptr(), d_hll_inv.raw(), d_schur.raw());
      else if (uniform_pose_dim == 6 && uniform_landmark_dim == 3) k_schur_mul_fixed<S, 6, 3><<<nc, 64>>>(d_mul_ops.raw(), ch, d_offsets.raw(), H.get_values_ptr(), d_hll_inv.raw(), d_schur.raw());
      else if (uniform_pose_dim == 3 && uniform_landmark_dim == 2) k_schur_mul_fixed<S, 3, 2><<<nc, 64>>>(d_mul_ops.raw(), ch, d_offsets.raw(), H.get_values_ptr(), d_hll_inv.raw(), d_schur.raw());
      else
      k_schur_mul<S><<<nc, SCHUR_MUL_THREADS>>>(d_mul_ops.raw(), ch, d_row_indices.raw(), d_block_col.raw(), d_offsets.raw(), d_schur_offsets.raw(), H.get_values_ptr(), d_hll_inv.raw(), d_schur.raw(), (uint32_t)max_landmark_dim);
      k_schur_mul_join<S><<<(unsigned)num_blocks, 128>>>(num_blocks, ch, d_row_indices.raw(), d_block_col.raw(), d_offsets.raw(), d_schur_offsets.raw(), d_schur.raw());
    }
    // b_S = b_p - Hpl Hll^-1 b_l (:901-920)
    const T *b = graph->get_b().raw();
    GRAPHITE_HIP(hipMemcpy(b_Schur.raw(), b, pose_dim * sizeof(T), hipMemcpyDefault));
    if (d_inv_ops.size()) {
      k_schur_apply_inverse<T, S><<<blocks(d_inv_ops.size()), TPB>>>(d_inv_ops.raw(), d_inv_ops.size(), d_inv_lrow.raw(), d_hll_inv.raw(), b + pose_dim, l_workspace.raw());
      if (d_hpl_ops.size()) {
        const unsigned np = (unsigned)landmark_col_start;
#define GRAPHITE_HPL_POSES(D, DL) k_schur_hpl_poses<T, S, D, DL><<<np, 64>>>(d_hpl_pose_first.raw(), d_hpl_pose_idx.raw(), d_hpl_ops.raw(), H.get_values_ptr(), l_workspace.raw(), b_Schur.raw())
        if (uniform_pose_dim == 9 && uniform_landmark_dim == 3) GRAPHITE_HPL_POSES(9, 3);
        else if (uniform_pose_dim == 6 && uniform_landmark_dim == 3) GRAPHITE_HPL_POSES(6, 3);
        else if (uniform_pose_dim == 3 && uniform_landmark_dim == 2) GRAPHITE_HPL_POSES(3, 2);
        else GRAPHITE_HPL_POSES(0, 0);
#undef GRAPHITE_HPL_POSES
      }
    }
    sync();
  }
  template <typename I> void build_csc_structure(Graph<T, S> *, CSCMatrix<S, I> &matrix) {
    detail::build_scalar_csc<S, I>(pose_dim, scalar_to_block_map, d_col_pointers, d_row_indices, d_offsets, d_schur_offsets, matrix);
  }
  template <typename I> void update_csc_values(Graph<T, S> *, CSCMatrix<S, I> &matrix) {
    detail::update_scalar_csc<S, I>(pose_dim, d_schur, scalar_to_block_map, d_col_pointers, d_row_indices, d_offsets, d_schur_offsets, matrix);
  }
  // x_l = Hll^-1 (b_l - Hpl^T x_p)  (schur.hpp:279-302); xl, xp: device pointers
  void compute_landmark_update(Graph<T, S> *graph, StreamPool &, T *xl, const T *xp) {
    using namespace detail;
    if (landmark_col_start >= num_block_columns) return;
    const T *b = graph->get_b().raw();
    GRAPHITE_HIP(hipMemcpy(l_rhs.raw(), b + pose_dim, landmark_dim * sizeof(T), hipMemcpyDefault));
    if (d_hpl_ops.size()) {
#define GRAPHITE_HPL_LM(D, DL) k_schur_hpl_landmarks<T, S, D, DL><<<blocks(d_inv_ops.size()), TPB>>>(d_inv_ops.size(), d_hpl_lm_first.raw(), d_hpl_ops.raw(), H.get_values_ptr(), xp, l_rhs.raw())
      if (uniform_pose_dim == 9 && uniform_landmark_dim == 3) GRAPHITE_HPL_LM(9, 3);
      else if (uniform_pose_dim == 6 && uniform_landmark_dim == 3) GRAPHITE_HPL_LM(6, 3);
      else if (uniform_pose_dim == 3 && uniform_landmark_dim == 2) GRAPHITE_HPL_LM(3, 2);
      else GRAPHITE_HPL_LM(0, 0);
#undef GRAPHITE_HPL_LM
    }
    k_schur_apply_inverse<T, S><<<blocks(d_inv_ops.size()), TPB>>>(d_inv_ops.raw(), d_inv_ops.size(), d_inv_lrow.raw(), d_hll_inv.raw(), l_rhs.raw(), xl);
    sync();
  }
  // vec_out = S vec_in (schur.hpp:347-393)
  void execute_schur_vector_multiply(Graph<T, S> *, StreamPool &, T *vec_out, const T *vec_in) {
    using namespace detail;
    fill<T>(vec_out, pose_dim, T(0)); // a block row without records keeps 0
    if (d_vec_ops.size()) {
      const unsigned np = (unsigned)landmark_col_start;
#define GRAPHITE_SVEC(D) k_schur_vec<T, S, D><<<np, 64>>>(d_vec_first.raw(), d_vec_idx.raw(), d_vec_ops.raw(), d_schur.raw(), vec_in, vec_out)
      if (uniform_pose_dim == 9) GRAPHITE_SVEC(9);
      else if (uniform_pose_dim == 6) GRAPHITE_SVEC(6);
      else if (uniform_pose_dim == 3) GRAPHITE_SVEC(3);
      else GRAPHITE_SVEC(0);
#undef GRAPHITE_SVEC
    }
    sync();
  }
  // dense row-major image of S (both triangles), leading dimension pose_dim: input of the direct reduced solve
  void to_dense(T *D) {
    using namespace detail;
    fill<T>(D, pose_dim * pose_dim, T(0));
    if (num_blocks) k_blocks_to_dense<T, S><<<(unsigned)num_blocks, 64>>>(num_blocks, d_block_col.raw(), d_row_indices.raw(), d_offsets.raw(), d_schur_offsets.raw(), d_schur.raw(), D, pose_dim);
    sync();
  }
};

} // namespace graphite
