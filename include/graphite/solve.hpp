// graphite/solve.hpp — solvers, preconditioners and the Levenberg–Marquardt driver of the generic layer
// (see core.hpp for scope).  Mirrors solver/solver.hpp:12-25, solver/pcg.hpp:35-232,
// preconditioner/{preconditioner,identity,block_jacobi}.hpp, solver/eigen.hpp:16-100 and
// optimizer/levenberg_marquardt.hpp:20-242 of the reference.
#pragma once
#include "core.hpp"
#include "sparse.hpp"
#include "../graphite_mi355x.h"

namespace graphite {

namespace detail {
template <typename T> __global__ void k_axpy(T *y, T a, const T *x, size_t n) { // y = a x + y
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i] + y[i];
}
template <typename T> __global__ void k_xpby(T *y, const T *x, T b, size_t n) { // y = x + b y
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] + b * y[i];
}
template <typename T> __global__ void k_scale_copy(T *y, T a, const T *x, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i];
}
template <typename T> __global__ void k_damp(T *v2, const T *p, const T *diag, T mu, int identity, size_t n) { // ops/vector.hpp:25-41
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) v2[i] += identity ? mu * p[i] : mu * diag[i] * p[i];
}
template <typename T> __global__ void k_clamp(T *d, T lo, T hi, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) d[i] = d[i] < lo ? lo : (d[i] > hi ? hi : d[i]);
}
template <typename T> __global__ void k_dot(const T *a, const T *b, size_t n, T *out) {
  __shared__ T red[TPB];
  T s = 0;
  for (size_t i = threadIdx.x; i < n; i += TPB) s += a[i] * b[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = TPB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *out = red[0];
}
// in-place inverse of `count` d x d blocks (column-major) after damping the diagonal
// (block_jacobi.hpp:120-172; the reference uses cuBLAS matinvBatched = Gauss-Jordan with pivoting)
// One thread per block, as the batched inverse of the reference; work matrices in LDS (core.hpp: lds_gauss_jordan).
template <typename T> __global__ void __launch_bounds__(BLOCK_INV_THREADS) k_block_inverse(const T *blocks, T *inv, size_t count, int d, T mu, int identity, const uint8_t *state) {
  extern __shared__ double lds_inv[];
  const int tid = threadIdx.x, nt = BLOCK_INV_THREADS;
  const size_t v = blockIdx.x * (size_t)nt + tid;
  if (v >= count || !is_vertex_active(state, v)) return;
  const T *B = blocks + v * d * d;
  T *X = inv + v * d * d;
  double *A = lds_inv + tid, *R = lds_inv + (size_t)d * d * nt + tid; // entry e of this thread: A[e * nt]
  for (int c = 0; c < d; ++c)
    for (int r = 0; r < d; ++r) {
      double val = (double)B[r + c * d];
      if (r == c) { const double cl = val < 1.0e-6 ? 1.0e-6 : (val > 1.0e32 ? 1.0e32 : val); val += identity ? (double)mu : (double)mu * cl; }
      A[(r + c * d) * nt] = val;
      R[(r + c * d) * nt] = r == c ? 1.0 : 0.0;
    }
  lds_gauss_jordan(A, R, d, nt);
  for (int i = 0; i < d * d; ++i) X[i] = (T)R[i * nt];
}
// (k_block_apply: core.hpp, with the VertexDescriptor member that also launches it)
template <typename T> __global__ void k_damp_dense(T *H, size_t n, T mu, int identity) { // hessian.hpp:136-176
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = (double)H[i * n + i];
  const double cl = d < 1.0e-6 ? 1.0e-6 : (d > 1.0e32 ? 1.0e32 : d);
  H[i * n + i] = (T)(identity ? d + (double)mu : d + (double)mu * cl);
}
// two stages, fixed partition (up to DOT_BLOCKS block partials in scratch[1..], summed in order into scratch[0]): the
// same bits every run, and the whole chip reads the vectors instead of one workgroup (183 k doubles: 180 us -> 6 us)
constexpr int DOT_BLOCKS = 256;
constexpr size_t DOT_SCRATCH = DOT_BLOCKS + 1;
template <typename T, bool RHO> __global__ void k_dot_partial(const T *a, const T *b, size_t n, T mu, T *partial) {
  __shared__ T red[TPB];
  const size_t per = (n + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  T s = 0;
  for (size_t i = i0 + threadIdx.x; i < i1; i += TPB) s += RHO ? a[i] * (mu * a[i] + b[i]) : a[i] * b[i]; // RHO: levenberg_marquardt.hpp:20-47
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = TPB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// one wave: lane L adds partials L, L + 64, ... in order, then a fixed butterfly (a single thread adding 256 partials took 12 us)
template <typename T> __global__ void k_dot_final(const T *partial, int nb, T *out) {
  T s = 0;
  for (int b = threadIdx.x; b < nb; b += 64) s += partial[b];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) *out = s;
}
template <typename T, bool RHO = false> inline T dot(const T *a, const T *b, size_t n, T *scratch /* DOT_SCRATCH elements */, T mu = T(0)) {
  const int nb = (int)std::max<size_t>(1, std::min<size_t>(DOT_BLOCKS, (n + 4 * TPB - 1) / (4 * TPB)));
  k_dot_partial<T, RHO><<<nb, TPB>>>(a, b, n, mu, scratch + 1);
  k_dot_final<T><<<1, 64>>>(scratch + 1, nb, scratch);
  T result;
  GRAPHITE_HIP(hipMemcpy(&result, scratch, sizeof(T), hipMemcpyDeviceToHost)); // ordered behind the kernels, no host mapping of HBM needed
  return result;
}

// ---- device-resident PCG control (solver/pcg.hpp:61-232, solver/pcg_schur.hpp:79-168) ---------------------------------------
// The reference's loops read three dot products per inner iteration back to the host (and so did rounds 1-3 here: three
// synchronising copies, ~25 us each, plus a D2D copy that blocks).  Here the scalars never leave the GPU: every dot product is
// a set of per-workgroup partials in fixed order, the kernels that need alpha / beta / the loop decision re-derive them from
// the partials and a small control record, and the host only watches one pinned word per iteration to know when to stop
// enqueueing (with one iteration of look-ahead while the previous solve ran that long).  Same iterates, same stopping and
// rejection rules; the preconditioner is applied to r and the 1 / ||r|| of pcg.hpp:118,190 is folded into the scalars (it is
// linear).
template <typename T> struct PcgCtl { T rz, rz0, beta, rinv; int done, iters, reject, pad; };
template <typename T> __device__ inline T sum_partials(const T *partial, int nb, T *lds /* [TPB] */) { // every thread gets the sum, fixed order
  T s = 0;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) s += partial[b];
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) lds[threadIdx.x] += lds[threadIdx.x + o]; __syncthreads(); }
  const T tot = lds[0];
  __syncthreads();
  return tot;
}
// partial[b] = sum over block b's range of a[i] * b[i]; DAMP: first v2 += mu (diag | 1) p (ops/vector.hpp:25-41), then p . v2
template <typename T, bool DAMP> __global__ void k_pcg_dot(const PcgCtl<T> *ctl, T *a, const T *b, const T *diag, T mu, int identity, size_t n, T *partial) {
  __shared__ T red[TPB];
  if (ctl->done) return;
  const size_t per = (n + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  T s = 0;
  for (size_t i = i0 + threadIdx.x; i < i1; i += TPB) {
    T av = a[i];
    if (DAMP) { av += identity ? mu * b[i] : mu * diag[i] * b[i]; a[i] = av; }
    s += av * b[i];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = TPB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// start of a solve: rz = r . z (z = Minv r, times 1 / ||r|| with NORM), p = z; hflag[*] cleared by the host
template <typename T, bool NORM> __global__ void k_pcg_start(PcgCtl<T> *ctl, const T *part_rz, const T *part_rr, int nb, volatile int *hflag0) {
  __shared__ T lds[TPB];
  const T rzp = sum_partials(part_rz, nb, lds);
  T rinv = T(1);
  if (NORM) rinv = (T)(1.0 / std::sqrt((double)sum_partials(part_rr, nb, lds)));
  if (threadIdx.x == 0) {
    ctl->rz = rzp * rinv; ctl->rz0 = std::numeric_limits<T>::infinity(); ctl->beta = T(0); ctl->rinv = rinv;
    ctl->done = (rzp * rinv == T(0)) ? 1 : 0; ctl->iters = 0; ctl->reject = 0;
    if (ctl->done) { *hflag0 = 2; __threadfence_system(); } // pcg.hpp:133: rz == 0 ends the loop before it starts
  }
}
// x_backup = x; x += alpha p; r -= alpha v2 with alpha = rz / (p . v2) re-derived by every workgroup; partial r . r
template <typename T> __global__ void k_pcg_xr(const PcgCtl<T> *ctl, const T *part_pap, int nb, T *x, T *xb, T *r, const T *p, const T *v2, size_t n, T *part_rr, int nb_out) {
  __shared__ T lds[TPB];
  if (ctl->done) return;
  const T denom = sum_partials(part_pap, nb, lds);
  if (denom == T(0) || denom != denom) return; // pcg_schur.hpp:122: the decision kernel closes the loop
  const T alpha = ctl->rz / denom;
  const size_t per = (n + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  T s = 0;
  for (size_t i = i0 + threadIdx.x; i < i1; i += TPB) {
    const T xo = x[i];
    xb[i] = xo;
    x[i] = alpha * p[i] + xo;
    const T rn = -alpha * v2[i] + r[i];
    r[i] = rn;
    s += rn * rn;
  }
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int o = TPB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) lds[threadIdx.x] += lds[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0 && (int)blockIdx.x < nb_out) part_rr[blockIdx.x] = lds[0];
}
// the loop decision after iteration k (pcg.hpp:196-229): rejection, tolerance, beta; one workgroup; the host sees hflag[k]
template <typename T, bool NORM> __global__ void k_pcg_decide(PcgCtl<T> *ctl, const T *part_pap, const T *part_rz, const T *part_rr, int nb, T tol, T rej, int k, volatile int *hflag) {
  __shared__ T lds[TPB];
  if (ctl->done) { if (threadIdx.x == 0) { hflag[k] = 2; __threadfence_system(); } return; }
  const T denom = sum_partials(part_pap, nb, lds);
  const T rzp = sum_partials(part_rz, nb, lds);
  T rinv = T(1);
  if (NORM) rinv = (T)(1.0 / std::sqrt((double)sum_partials(part_rr, nb, lds)));
  if (threadIdx.x != 0) return;
  int flag = 1;
  if (denom == T(0) || denom != denom) { ctl->done = 1; flag = 2; }
  else {
    const T rz_new = rzp * rinv;
    ctl->iters = k + 1;
    const T arz = rz_new < T(0) ? -rz_new : rz_new;
    if (arz > rej * ctl->rz0 || rz_new != rz_new) { ctl->done = 1; ctl->reject = 1; flag = 2; }
    else {
      ctl->rz0 = ctl->rz0 < arz ? ctl->rz0 : arz;
      ctl->beta = rz_new / ctl->rz;
      ctl->rz = rz_new;
      ctl->rinv = rinv;
      if (arz < tol) { ctl->done = 1; flag = 2; }
      else if (rz_new == T(0)) { ctl->done = 1; flag = 2; } // the next trip would leave at pcg.hpp:133
    }
  }
  hflag[k] = flag;
  __threadfence_system();
}
// p = z / ||r|| + beta p (first: p = z / ||r||); a rejected step restores x from its backup (pcg.hpp:203-207)
template <typename T> __global__ void k_pcg_dir(PcgCtl<T> *ctl, T *p, const T *z, T *x, const T *xb, size_t n, int first) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (ctl->reject) { if (i < n) x[i] = xb[i]; return; } // stays set for the launches enqueued ahead: they restore the same values again
  if (ctl->done || i >= n) return;
  p[i] = first ? z[i] * ctl->rinv : z[i] * ctl->rinv + ctl->beta * p[i];
}
template <typename T> struct DevicePcg {
  hbm_vector<PcgCtl<T>> ctl;
  hbm_vector<T> part_pap, part_rz, part_rr;
  volatile int *hflag = nullptr;
  size_t cap = 0;
  int predicted = 1 << 30;
  DevicePcg() { ctl.resize(1); part_pap.resize(DOT_BLOCKS); part_rz.resize(DOT_BLOCKS); part_rr.resize(DOT_BLOCKS); }
  DevicePcg(const DevicePcg &) = delete;
  ~DevicePcg() { if (hflag) (void)hipHostFree(const_cast<int *>(hflag)); }
  static int nblocks(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>(DOT_BLOCKS, (n + 4 * TPB - 1) / (4 * TPB))); }
  void begin(size_t max_iter) {
    if (max_iter + 1 > cap) {
      if (hflag) (void)hipHostFree(const_cast<int *>(hflag));
      void *q = nullptr;
      GRAPHITE_HIP(hipHostMalloc(&q, (max_iter + 1) * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
      hflag = static_cast<volatile int *>(q);
      cap = max_iter + 1;
    }
    for (size_t k = 0; k <= max_iter; ++k) hflag[k] = 0;
  }
  // runs the loop: enqueue(k) puts iteration k's kernels into the (null) stream; returns the number of iterations executed
  template <typename Enqueue> size_t run(size_t max_iter, Enqueue &&enqueue) {
    size_t enq = 0, ran = 0;
    if (max_iter > 0 && hflag[0] == 0) { enqueue(0); enq = 1; }
    for (size_t k = 0; k < max_iter; ++k) {
      if (k + 1 < max_iter && (int)(k + 1) < predicted && enq == k + 1) { enqueue(k + 1); ++enq; } // look-ahead
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned it = 0; __atomic_load_n(const_cast<const int *>(&hflag[k]), __ATOMIC_ACQUIRE) == 0; ++it)
        if ((it & 0xFFFF) == 0xFFFF && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) {
          GRAPHITE_HIP(hipDeviceSynchronize()); // surfaces a kernel fault instead of hanging
          if (hflag[k] == 0) throw std::runtime_error("PCG: timeout waiting for the device's loop decision");
        }
      ran = k + 1;
      if (hflag[k] == 2) break;
      if (k + 1 < max_iter && enq == k + 1) { enqueue(k + 1); ++enq; }
    }
    sync();
    PcgCtl<T> h;
    GRAPHITE_HIP(hipMemcpy(&h, ctl.raw(), sizeof(h), hipMemcpyDeviceToHost));
    predicted = std::max(1, h.iters);
    (void)ran;
    return (size_t)h.iters;
  }
};
} // namespace detail

// ---- preconditioner/preconditioner.hpp:8-20 ------------------------------------------------------
template <typename T, typename S> class Preconditioner {
public:
  virtual ~Preconditioner() = default;
  virtual void update_structure(Graph<T, S> *graph, StreamPool &streams) = 0;
  virtual void update_values(Graph<T, S> *graph, StreamPool &streams) = 0;
  virtual void set_damping_factor(Graph<T, S> *graph, T damping_factor, const bool use_identity, StreamPool &streams) = 0;
  virtual void apply(Graph<T, S> *graph, T *z, const T *r, StreamPool &streams) = 0;
  virtual int engine_kind() const { return -1; } // gr_solver this preconditioner turns PCGSolver into, -1 = none
};

template <typename T, typename S> class IdentityPreconditioner : public Preconditioner<T, S> {
public:
  int engine_kind() const override { return GR_SOLVER_PCG_IDENTITY; }
  void update_structure(Graph<T, S> *, StreamPool &) override {}
  void update_values(Graph<T, S> *, StreamPool &) override {}
  void set_damping_factor(Graph<T, S> *, T, const bool, StreamPool &) override {}
  void apply(Graph<T, S> *graph, T *z, const T *r, StreamPool &) override {
    GRAPHITE_HIP(hipMemcpyAsync(z, r, graph->get_hessian_dimension() * sizeof(T), hipMemcpyDeviceToDevice, nullptr));
  }
};

// preconditioner/block_jacobi.hpp:79-186: one d x d block of J^T rho' P J per vertex, damped and inverted
template <typename T, typename S> class BlockJacobiPreconditioner : public Preconditioner<T, S> {
  std::vector<std::unique_ptr<hbm_vector<T>>> blocks, inverses;
public:
  int engine_kind() const override { return GR_SOLVER_PCG; }
  void update_structure(Graph<T, S> *graph, StreamPool &) override {
    auto &vds = graph->get_vertex_descriptors();
    blocks.clear(); inverses.clear();
    for (auto *vd : vds) {
      if (vd->dimension() > 16) throw std::invalid_argument("BlockJacobiPreconditioner: vertex dimension > 16");
      blocks.emplace_back(new hbm_vector<T>(vd->count() * vd->dimension() * vd->dimension()));
      inverses.emplace_back(new hbm_vector<T>(vd->count() * vd->dimension() * vd->dimension()));
    }
  }
  void update_values(Graph<T, S> *graph, StreamPool &) override {
    auto &vds = graph->get_vertex_descriptors();
    for (auto &b : blocks) detail::fill<T>(b->raw(), b->size(), T(0));
    for (auto *fd : graph->get_factor_descriptors())
      for (size_t s = 0; s < fd->num_slots(); ++s) {
        const size_t k = std::find(vds.begin(), vds.end(), fd->slot_descriptor(s)) - vds.begin();
        if (k < vds.size()) fd->block_diagonal(s, blocks[k]->raw());
      }
    detail::sync();
  }
  void set_damping_factor(Graph<T, S> *graph, T mu, const bool use_identity, StreamPool &) override {
    auto &vds = graph->get_vertex_descriptors();
    detail::allow_block_inverse_lds(reinterpret_cast<const void *>(&detail::k_block_inverse<T>));
    for (size_t k = 0; k < vds.size(); ++k)
      if (vds[k]->count())
        detail::k_block_inverse<T><<<(unsigned)((vds[k]->count() + detail::BLOCK_INV_THREADS - 1) / detail::BLOCK_INV_THREADS), detail::BLOCK_INV_THREADS,
                                     2 * vds[k]->dimension() * vds[k]->dimension() * detail::BLOCK_INV_THREADS * sizeof(double)>>>(
            blocks[k]->raw(), inverses[k]->raw(), vds[k]->count(), (int)vds[k]->dimension(), mu, use_identity ? 1 : 0, vds[k]->device_active_state());
    detail::launch_check();
    detail::sync();
  }
  void apply(Graph<T, S> *graph, T *z, const T *r, StreamPool &) override {
    auto &vds = graph->get_vertex_descriptors();
    for (size_t k = 0; k < vds.size(); ++k)
      if (vds[k]->count())
        detail::k_block_apply<T, T><<<detail::blocks(vds[k]->count() * vds[k]->dimension()), detail::TPB>>>(inverses[k]->raw(), vds[k]->device_hessian_ids(), vds[k]->device_active_state(), vds[k]->count(), (int)vds[k]->dimension(), z, r);
  }
};

// ---- solver/solver.hpp:12-25 ---------------------------------------------------------------------
template <typename T, typename S> class Solver {
public:
  virtual ~Solver() = default;
  virtual void update_structure(Graph<T, S> *graph, StreamPool &streams) = 0;
  virtual void update_values(Graph<T, S> *graph, StreamPool &streams) = 0;
  virtual void set_damping_factor(Graph<T, S> *graph, T damping_factor, const bool use_identity, StreamPool &streams) = 0;
  virtual bool solve(Graph<T, S> *graph, T *delta_x, StreamPool &streams) = 0;
  // the gr_solver of libgraphite_mi355x.so that runs the same algorithm on a BAL-model graph (-1 = none), and its
  // PCG parameters: used by the optimizer to hand tagged bundle-adjustment graphs to the specialised engine
  virtual int engine_kind(size_t /*num_cameras*/) const { return -1; }
  virtual void engine_pcg_parameters(int &max_iter, double &tol, double &rejection_ratio) const { max_iter = 10; tol = 1.0; rejection_ratio = 5.0; }
};

// solver/pcg.hpp:35-232: matrix-free PCG on (J^T rho' P J + mu D) x = b; same iterates, stopping and
// rejection rules (the preconditioner is applied to r / ||r||)
template <typename T, typename S> class PCGSolver : public Solver<T, S> {
  size_t max_iter;
  T tol, rejection_ratio;
  Preconditioner<T, S> *preconditioner;
  hbm_vector<T> r, p, z, v2, diag, y, xb, scratch;
  detail::DevicePcg<T> dev;
  T damping = 0;
  bool damping_identity = false;
  size_t iterations_ = 0;
public:
  PCGSolver(size_t max_iter_, T tol_, T rejection_ratio_, Preconditioner<T, S> *preconditioner_)
      : max_iter(max_iter_), tol(tol_), rejection_ratio(rejection_ratio_), preconditioner(preconditioner_) { scratch.resize(detail::DOT_SCRATCH); }
  size_t last_iterations() const { return iterations_; }
  int engine_kind(size_t) const override { return preconditioner->engine_kind(); }
  void engine_pcg_parameters(int &m, double &t, double &rj) const override { m = (int)max_iter; t = (double)tol; rj = (double)rejection_ratio; }
  void update_structure(Graph<T, S> *graph, StreamPool &streams) override {
    const size_t n = graph->get_hessian_dimension();
    r.resize(n); p.resize(n); z.resize(n); v2.resize(n); diag.resize(n); y.resize(n); xb.resize(n);
    preconditioner->update_structure(graph, streams);
  }
  void update_values(Graph<T, S> *graph, StreamPool &streams) override {
    const size_t n = graph->get_hessian_dimension();
    detail::fill<T>(diag.raw(), n, T(0)); // pcg.hpp:93-103: clamped scalar diagonal of the (scaled) J^T rho' P J
    for (auto *fd : graph->get_factor_descriptors()) fd->scalar_diagonal(diag.raw());
    if (n) detail::k_clamp<T><<<detail::blocks(n), detail::TPB>>>(diag.raw(), T(1.0e-6), T(1.0e32), n);
    preconditioner->update_values(graph, streams);
  }
  void set_damping_factor(Graph<T, S> *graph, T mu, const bool use_identity, StreamPool &streams) override {
    damping = mu; damping_identity = use_identity;
    preconditioner->set_damping_factor(graph, mu, use_identity, streams);
  }
  bool solve(Graph<T, S> *graph, T *x, StreamPool &streams) override {
    using namespace detail;
    const size_t n = graph->get_hessian_dimension();
    if (!n) return true;
    const int nb = blocks(n), nd = DevicePcg<T>::nblocks(n);
    PcgCtl<T> *c = dev.ctl.raw();
    dev.begin(max_iter);
    fill<T>(x, n, T(0));
    GRAPHITE_HIP(hipMemcpyAsync(r.raw(), graph->get_b().raw(), n * sizeof(T), hipMemcpyDeviceToDevice, nullptr));
    // z' = Minv r; rz = r . z' / ||r||; p = z' / ||r||  (pcg.hpp:114-131 with the normalisation folded into the scalars)
    fill<int>(&c->done, 1, 0); // the dot kernels look at it
    k_pcg_dot<T, false><<<nd, TPB>>>(c, r.raw(), r.raw(), nullptr, T(0), 0, n, dev.part_rr.raw());
    preconditioner->apply(graph, z.raw(), r.raw(), streams);
    k_pcg_dot<T, false><<<nd, TPB>>>(c, r.raw(), z.raw(), nullptr, T(0), 0, n, dev.part_rz.raw());
    k_pcg_start<T, true><<<1, TPB>>>(c, dev.part_rz.raw(), dev.part_rr.raw(), nd, dev.hflag);
    k_pcg_dir<T><<<nb, TPB>>>(c, p.raw(), z.raw(), x, xb.raw(), n, 1);
    iterations_ = dev.run(max_iter, [&](size_t k) {
      graph->hessian_matvec(v2.raw(), p.raw());
      k_pcg_dot<T, true><<<nd, TPB>>>(c, v2.raw(), p.raw(), diag.raw(), damping, damping_identity ? 1 : 0, n, dev.part_pap.raw());
      k_pcg_xr<T><<<nd, TPB>>>(c, dev.part_pap.raw(), nd, x, xb.raw(), r.raw(), p.raw(), v2.raw(), n, dev.part_rr.raw(), nd);
      preconditioner->apply(graph, z.raw(), r.raw(), streams);
      k_pcg_dot<T, false><<<nd, TPB>>>(c, r.raw(), z.raw(), nullptr, T(0), 0, n, dev.part_rz.raw());
      k_pcg_decide<T, true><<<1, TPB>>>(c, dev.part_pap.raw(), dev.part_rz.raw(), dev.part_rr.raw(), nd, tol, rejection_ratio, (int)k, dev.hflag);
      k_pcg_dir<T><<<nb, TPB>>>(c, p.raw(), z.raw(), x, xb.raw(), n, 0);
    });
    return true;
  }
};

// ---- Schur elimination for the generic layer ------------------------------------------------------
// H = [Hpp Hpl; Hlp Hll] is reduced over the vertices marked set_eliminate exactly as the reference does it
// (schur.hpp:194-302): block-sparse Hessian<T,S> (upper block-CSC), SchurComplement<T,S> built from its structure
// (sparse.hpp).  Memory is O(blocks), not n^2: pose graphs and bundle-adjustment graphs of any size fit; eliminated
// vertices must not share a factor (Hll block diagonal), as in the reference.
namespace detail {
// d x d diagonal blocks of the sparse S, one per pose block (block_jacobi_schur.hpp:114-150): copy + invert
template <typename T, typename S> __global__ void __launch_bounds__(BLOCK_INV_THREADS)
k_schur_diag_inverse(size_t nblocks, const size_t *diag_off, const size_t *soff, const S *Sv, T *inv, const size_t *inv_off, int max_d) {
  extern __shared__ double lds_inv[];
  const int tid = threadIdx.x, nt = BLOCK_INV_THREADS;
  const size_t b = blockIdx.x * (size_t)nt + tid;
  if (b >= nblocks) return;
  const int d = (int)(soff[b + 1] - soff[b]);
  double *A = lds_inv + tid, *R = lds_inv + (size_t)max_d * max_d * nt + tid;
  const S *B = Sv + diag_off[b];
  for (int i = 0; i < d * d; ++i) { A[i * nt] = (double)B[i]; R[i * nt] = (i % d == i / d) ? 1.0 : 0.0; }
  lds_gauss_jordan(A, R, d, nt);
  for (int i = 0; i < d * d; ++i) inv[inv_off[b] + i] = (T)R[i * nt];
}
template <typename T> __global__ void k_schur_diag_apply(size_t pose_dim, const size_t *s2b_start, const size_t *soff, const T *inv, const size_t *inv_off, const size_t *s2b, T *z, const T *r) {
  const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (row >= pose_dim) return;
  const size_t b = s2b[row], r0 = soff[b], d = soff[b + 1] - r0, i = row - r0;
  T s = 0;
  for (size_t c = 0; c < d; ++c) s += inv[inv_off[b] + i + d * c] * r[r0 + c];
  z[row] = s;
}
} // namespace detail

// preconditioner/schur_preconditioner.hpp:10-28
template <typename T, typename S> class SchurPreconditioner {
public:
  virtual ~SchurPreconditioner() = default;
  virtual void update_structure(Graph<T, S> *graph, SchurComplement<T, S> *schur, StreamPool &streams) = 0;
  virtual void update_values(Graph<T, S> *graph, SchurComplement<T, S> *schur, StreamPool &streams) = 0;
  virtual void set_damping_factor(Graph<T, S> *graph, SchurComplement<T, S> *schur, T damping_factor, const bool use_identity, StreamPool &streams) = 0;
  virtual void apply(Graph<T, S> *graph, SchurComplement<T, S> *schur, T *z, const T *r, StreamPool &streams) = 0;
  virtual bool is_block_jacobi() const { return false; }
};
// preconditioner/block_jacobi_schur.hpp:114-178: inverse diagonal blocks of S
template <typename T, typename S> class BlockJacobiSchurPreconditioner : public SchurPreconditioner<T, S> {
  device_vector<T> inverses;
  device_vector<size_t> d_diag_off, d_soff, d_inv_off, d_s2b;
  size_t nblocks = 0, pose_dim = 0, max_dim = 1;
public:
  bool is_block_jacobi() const override { return true; }
  void update_structure(Graph<T, S> *, SchurComplement<T, S> *schur, StreamPool &) override {
    const auto &soff = schur->host_pose_offsets();
    nblocks = schur->num_pose_blocks(); pose_dim = schur->get_pose_dimension();
    std::vector<size_t> inv_off(nblocks), s2b(pose_dim);
    size_t total = 0;
    for (size_t b = 0; b < nblocks; ++b) {
      const size_t d = soff[b + 1] - soff[b];
      if (d > 16) throw std::invalid_argument("BlockJacobiSchurPreconditioner: vertex dimension > 16");
      inv_off[b] = total; total += d * d;
      max_dim = std::max(max_dim, d);
      for (size_t c = soff[b]; c < soff[b + 1]; ++c) s2b[c] = b;
    }
    inverses.resize(total);
    d_diag_off = schur->host_diag_offsets(); d_soff = soff; d_inv_off = inv_off; d_s2b = s2b;
  }
  // S already carries the damping (it is reduced from the damped H), so the blocks are inverted as they are
  void update_values(Graph<T, S> *, SchurComplement<T, S> *schur, StreamPool &) override {
    if (!nblocks) return;
    detail::allow_block_inverse_lds(reinterpret_cast<const void *>(&detail::k_schur_diag_inverse<T, S>));
    detail::k_schur_diag_inverse<T, S><<<(unsigned)((nblocks + detail::BLOCK_INV_THREADS - 1) / detail::BLOCK_INV_THREADS), detail::BLOCK_INV_THREADS, detail::block_inverse_lds_bytes(max_dim)>>>(
        nblocks, d_diag_off.raw(), d_soff.raw(), schur->get_values_ptr(), inverses.raw(), d_inv_off.raw(), (int)max_dim);
    detail::launch_check();
  }
  void set_damping_factor(Graph<T, S> *, SchurComplement<T, S> *, T, const bool, StreamPool &) override {}
  void apply(Graph<T, S> *, SchurComplement<T, S> *, T *z, const T *r, StreamPool &) override {
    if (pose_dim) detail::k_schur_diag_apply<T><<<detail::blocks(pose_dim), detail::TPB>>>(pose_dim, nullptr, d_soff.raw(), inverses.raw(), d_inv_off.raw(), d_s2b.raw(), z, r);
  }
};

// solver/pcg_schur.hpp:42-168: PCG on the reduced system, then the landmark back-substitution
template <typename T, typename S> class PCGSchurSolver : public Solver<T, S> {
  Hessian<T, S> H;
  SchurComplement<T, S> schur;
  SchurPreconditioner<T, S> *preconditioner;
  hbm_vector<T> r, p, z, Ap, xb, scratch;
  detail::DevicePcg<T> dev;
  size_t max_iter, iterations_ = 0;
  T tol, rejection_ratio;
public:
  PCGSchurSolver(size_t max_iter_, T tol_, T rejection_ratio_, SchurPreconditioner<T, S> *preconditioner_)
      : schur(H), preconditioner(preconditioner_), max_iter(max_iter_), tol(tol_), rejection_ratio(rejection_ratio_) { scratch.resize(detail::DOT_SCRATCH); }
  size_t last_iterations() const { return iterations_; }
  // explicit S while its dense block map fits the engine (16384 cameras), the implicit form (same iterates) beyond
  int engine_kind(size_t num_cameras) const override {
    return !preconditioner->is_block_jacobi() ? -1 : num_cameras <= 16384 ? GR_SOLVER_PCG_SCHUR : GR_SOLVER_PCG_SCHUR_IMPLICIT;
  }
  void engine_pcg_parameters(int &m, double &t, double &rj) const override { m = (int)max_iter; t = (double)tol; rj = (double)rejection_ratio; }
  void update_structure(Graph<T, S> *graph, StreamPool &streams) override {
    if (graph->get_pose_dimension() == 0 || graph->get_pose_dimension() > graph->get_hessian_dimension())
      throw std::invalid_argument("Schur solver: no vertex descriptor left after elimination");
    H.build_structure(graph, streams);
    schur.build_structure(graph, streams);
    const size_t pd = schur.get_pose_dimension();
    r.resize(pd); p.resize(pd); z.resize(pd); Ap.resize(pd); xb.resize(pd);
    preconditioner->update_structure(graph, &schur, streams);
  }
  void update_values(Graph<T, S> *graph, StreamPool &streams) override { H.update_values(graph, streams); }
  void set_damping_factor(Graph<T, S> *graph, T mu, const bool use_identity, StreamPool &streams) override {
    H.apply_damping(graph, mu, use_identity, streams);
    preconditioner->set_damping_factor(graph, &schur, mu, use_identity, streams);
  }
  bool solve(Graph<T, S> *graph, T *x, StreamPool &streams) override {
    using namespace detail;
    const size_t n = graph->get_hessian_dimension(), pd = schur.get_pose_dimension();
    if (!n) return true;
    schur.update_values(graph, streams); // S depends on the damping: reduced here, as pcg_schur.hpp:84 does
    preconditioner->update_values(graph, &schur, streams);
    const int nb = blocks(pd), nd = DevicePcg<T>::nblocks(pd);
    PcgCtl<T> *c = dev.ctl.raw();
    dev.begin(max_iter);
    fill<T>(x, n, T(0));
    GRAPHITE_HIP(hipMemcpyAsync(r.raw(), schur.get_b_Schur().raw(), pd * sizeof(T), hipMemcpyDeviceToDevice, nullptr));
    fill<int>(&c->done, 1, 0);
    preconditioner->apply(graph, &schur, z.raw(), r.raw(), streams);
    k_pcg_dot<T, false><<<nd, TPB>>>(c, r.raw(), z.raw(), nullptr, T(0), 0, pd, dev.part_rz.raw());
    k_pcg_start<T, false><<<1, TPB>>>(c, dev.part_rz.raw(), nullptr, nd, dev.hflag);
    k_pcg_dir<T><<<nb, TPB>>>(c, p.raw(), z.raw(), x, xb.raw(), pd, 1);
    iterations_ = dev.run(max_iter, [&](size_t k) {
      schur.execute_schur_vector_multiply(graph, streams, Ap.raw(), p.raw());
      k_pcg_dot<T, false><<<nd, TPB>>>(c, Ap.raw(), p.raw(), nullptr, T(0), 0, pd, dev.part_pap.raw());
      k_pcg_xr<T><<<nd, TPB>>>(c, dev.part_pap.raw(), nd, x, xb.raw(), r.raw(), p.raw(), Ap.raw(), pd, dev.part_rr.raw(), nd);
      preconditioner->apply(graph, &schur, z.raw(), r.raw(), streams);
      k_pcg_dot<T, false><<<nd, TPB>>>(c, r.raw(), z.raw(), nullptr, T(0), 0, pd, dev.part_rz.raw());
      k_pcg_decide<T, false><<<1, TPB>>>(c, dev.part_pap.raw(), dev.part_rz.raw(), nullptr, nd, tol, rejection_ratio, (int)k, dev.hflag);
      k_pcg_dir<T><<<nb, TPB>>>(c, p.raw(), z.raw(), x, xb.raw(), pd, 0);
    });
    schur.compute_landmark_update(graph, streams, x + pd, x);
    sync();
    return true;
  }
};

// solver/eigen_schur.hpp:20-108: direct solve of the reduced system (here: the MFMA Cholesky), then the back-substitution
template <typename T, typename S> class EigenSchurLDLTSolver : public Solver<T, S> {
  Hessian<T, S> H;
  SchurComplement<T, S> schur;
  device_vector<T> dense;
public:
  EigenSchurLDLTSolver() : schur(H) {}
  int engine_kind(size_t) const override { return GR_SOLVER_DENSE_SCHUR; }
  void update_structure(Graph<T, S> *graph, StreamPool &streams) override {
    if (graph->get_pose_dimension() == 0 || graph->get_pose_dimension() > graph->get_hessian_dimension())
      throw std::invalid_argument("Schur solver: no vertex descriptor left after elimination");
    H.build_structure(graph, streams);
    schur.build_structure(graph, streams);
    dense.resize(schur.get_pose_dimension() * schur.get_pose_dimension());
  }
  void update_values(Graph<T, S> *graph, StreamPool &streams) override { H.update_values(graph, streams); }
  void set_damping_factor(Graph<T, S> *graph, T mu, const bool use_identity, StreamPool &streams) override { H.apply_damping(graph, mu, use_identity, streams); }
  bool solve(Graph<T, S> *graph, T *x, StreamPool &streams) override {
    const size_t n = graph->get_hessian_dimension(), pd = schur.get_pose_dimension();
    if (!n) return true;
    schur.update_values(graph, streams);
    schur.to_dense(dense.raw());
    int dev = 0;
    GRAPHITE_HIP(hipGetDevice(&dev));
    detail::fill<T>(x, n, T(0));
    const gr_status st = gr_dense_cholesky_solve(sizeof(T) == 8 ? GR_F64 : GR_F32, (int64_t)pd, dense.raw(), (int64_t)pd, schur.get_b_Schur().raw(), x, dev, nullptr, nullptr);
    if (st != GR_OK) return false;
    schur.compute_landmark_update(graph, streams, x + pd, x);
    detail::sync();
    return true;
  }
};

// solver/eigen.hpp:16-100 (EigenLDLTSolver) / solver/cudss.hpp:183-256 (cudssSolver): direct solve of the FULL system
// (J^T rho' P J damped) x = b.  A sparse direct factorisation is free to choose its elimination order, and for a graph whose
// set_eliminate descriptors form a block-diagonal trailing part (bundle adjustment: the points; examples/bal.cu:156 marks them
// for every --solver) the order "those vertices first" IS the Schur reduction followed by the back-substitution: the
// block-diagonal pivots are inverted per vertex, the remaining system is S, and x_l = Hll^-1 (b_l - Hpl^T x_p).  That is what
// runs here — the block-sparse Hessian / SchurComplement of sparse.hpp and the tile Cholesky of libgraphite_mi355x.so on S —
// so --solver eigen | cudss works at bundle-adjustment scale with O(blocks) memory; the step equals the reference's
// (tests/schur.cu:242-289 holds the two to 1e-8 there).  Graphs without eliminated descriptors, or with a storage type other
// than the graph's, are assembled densely (n^2 memory: small graphs).  On the gr_bal engine: GR_SOLVER_DENSE_SCHUR.
template <typename T, typename S> class EigenLDLTSolver : public Solver<T, S> {
  hbm_vector<T> Hd, Hdd;
  Hessian<T, S> H;
  SchurComplement<T, S> schur;
  device_vector<T> dense;
  T damping = 0;
  bool damping_identity = false, eliminate_first = false;
  static constexpr bool same_type = std::is_same<T, S>::value;
  // no elimination order, every block column of the same dimension, 512 columns or more (a pose graph): the block-sparse
  // Hessian goes to the nested-dissection tile Cholesky (gr_spchol) instead of a dense n x n array
  gr_spchol *sparse_chol = nullptr;
  bool sparse_direct = false;
  size_t sp_nb = 0, sp_bs = 0;
  int sp_dev = -1;
  std::vector<int64_t> sp_row, sp_col; // the block structure sparse_chol was analysed for
public:
  EigenLDLTSolver() : schur(H) {}
  EigenLDLTSolver(const EigenLDLTSolver &) = delete;
  ~EigenLDLTSolver() { if (sparse_chol) gr_spchol_destroy(sparse_chol); }
  bool uses_sparse_factorisation() const { return sparse_direct; }
  int engine_kind(size_t) const override { return GR_SOLVER_DENSE_SCHUR; }
  bool uses_elimination_order() const { return eliminate_first; }
  void update_structure(Graph<T, S> *graph, StreamPool &streams) override {
    const size_t n = graph->get_hessian_dimension(), pd = graph->get_pose_dimension();
    eliminate_first = false;
    if constexpr (same_type) {
      if (pd > 0 && pd < n) {
        try {
          H.build_structure(graph, streams);
          schur.build_structure(graph, streams); // throws when the eliminated part is not block diagonal
          dense.resize(schur.get_pose_dimension() * schur.get_pose_dimension());
          eliminate_first = true;
        } catch (const std::invalid_argument &) { eliminate_first = false; }
      }
    }
    sparse_direct = false;
    if constexpr (same_type && std::is_floating_point<T>::value) {
      const size_t nb = graph->get_num_block_columns();
      const size_t sparse_min = getenv("GRAPHITE_LDLT_SPARSE_MIN") ? (size_t)std::max(1, atoi(getenv("GRAPHITE_LDLT_SPARSE_MIN"))) : 512; // (measured on SE(2) pose graphs: 897 columns 0.61 ms sparse / 1.56 dense per LM iteration; smaller graphs do not dissect)
      if (!eliminate_first && nb && n % nb == 0 && n >= sparse_min) {
        const size_t bs = n / nb;
        bool uniform = bs <= 128;
        for (size_t j = 0; j < nb && uniform; ++j) uniform = graph->get_variable_dimension(j) == bs;
        if (uniform) {
          H.build_structure(graph, streams);
          const auto &cp = H.host_col_pointers(), &ri = H.host_row_indices();
          std::vector<int64_t> brow(ri.size()), bcol(ri.size());
          for (size_t j = 0; j < nb; ++j)
            for (size_t q = cp[j]; q < cp[j + 1]; ++q) { brow[q] = (int64_t)ri[q]; bcol[q] = (int64_t)j; }
          int dev = 0;
          GRAPHITE_HIP(hipGetDevice(&dev));
          // the analysis (nested dissection, tile symbolic factorisation, allocation: 10 k poses ~ 15 ms) is kept between optimiser calls
          // while the block structure is the one it was made for
          if (sparse_chol && sp_nb == nb && sp_bs == bs && sp_dev == dev && sp_row == brow && sp_col == bcol) { sparse_direct = true; return; }
          if (sparse_chol) { gr_spchol_destroy(sparse_chol); sparse_chol = nullptr; }
          const gr_status st = gr_spchol_create(&sparse_chol, sizeof(T) == 8 ? GR_F64 : GR_F32, (int64_t)nb, (int32_t)bs, (int64_t)ri.size(), brow.data(), bcol.data(), dev, nullptr);
          if (st == GR_OK) { sparse_direct = true; sp_nb = nb; sp_bs = bs; sp_dev = dev; sp_row = std::move(brow); sp_col = std::move(bcol); }
          else {
            sparse_chol = nullptr;
            if (st != GR_ERR_SOLVE_FAILED) throw std::runtime_error(std::string("graphite: EigenLDLTSolver: gr_spchol_create: ") + gr_last_error_string());
            if (getenv("GR_VERBOSE")) std::cerr << "[graphite] EigenLDLTSolver: " << gr_last_error_string() << "; dense factorisation" << std::endl;
          }
        }
      }
    }
    if (sparse_direct) return;
    if (sparse_chol) { gr_spchol_destroy(sparse_chol); sparse_chol = nullptr; } // (the graph is no longer one for the sparse form)
    if (!eliminate_first) {
      // no elimination order: the whole damped Hessian as ONE dense matrix for the MFMA Cholesky (the reference hands a sparse one to
      // SimplicialLDLT on the host, solver/eigen.hpp:49-98).  Fine for the graphs that solver is used on here (hundreds to a few thousand
      // columns); a graph whose n^2 does not fit says so instead of failing inside an allocation
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && 2.0 * (double)n * (double)n * sizeof(T) > 0.9 * (double)free_b + (double)(Hd.size() + Hdd.size()) * sizeof(T))
        throw std::runtime_error("graphite: EigenLDLTSolver without an elimination order factorises the dense " + std::to_string(n) + " x " + std::to_string(n) +
                                 " Hessian (" + std::to_string(2.0 * (double)n * (double)n * sizeof(T) / 1e9) + " GB): it does not fit this device — mark the landmark descriptor with set_eliminate(true), or use PCGSolver");
      Hd.resize(n * n); Hdd.resize(n * n);
    }
  }
  void update_values(Graph<T, S> *graph, StreamPool &streams) override {
    if (eliminate_first || sparse_direct) { H.update_values(graph, streams); return; }
    const size_t n = graph->get_hessian_dimension();
    detail::fill<T>(Hd.raw(), n * n, T(0));
    for (auto *fd : graph->get_factor_descriptors()) fd->dense_hessian(Hd.raw(), n);
    detail::sync();
  }
  void set_damping_factor(Graph<T, S> *graph, T mu, const bool use_identity, StreamPool &streams) override {
    damping = mu; damping_identity = use_identity;
    if (eliminate_first || sparse_direct) H.apply_damping(graph, mu, use_identity, streams);
  }
  bool solve(Graph<T, S> *graph, T *x, StreamPool &streams) override {
    const size_t n = graph->get_hessian_dimension();
    if (!n) return true;
    if constexpr (same_type) {
      if (sparse_direct) { // block-sparse damped Hessian (upper blocks, column-major) -> tile Cholesky -> x
        const bool ok = gr_spchol_factor_solve(sparse_chol, H.get_values_ptr(), graph->get_b().raw(), x) == GR_OK;
        detail::sync();
        return ok;
      }
    }
    int dev = 0;
    GRAPHITE_HIP(hipGetDevice(&dev));
    const gr_dtype dt = sizeof(T) == 8 ? GR_F64 : GR_F32;
    if (eliminate_first) {
      const size_t pd = schur.get_pose_dimension();
      schur.update_values(graph, streams);
      schur.to_dense(dense.raw());
      detail::fill<T>(x, n, T(0));
      if (gr_dense_cholesky_solve(dt, (int64_t)pd, dense.raw(), (int64_t)pd, schur.get_b_Schur().raw(), x, dev, nullptr, nullptr) != GR_OK) return false;
      schur.compute_landmark_update(graph, streams, x + pd, x);
      detail::sync();
      return true;
    }
    GRAPHITE_HIP(hipMemcpy(Hdd.raw(), Hd.raw(), n * n * sizeof(T), hipMemcpyDefault));
    detail::k_damp_dense<T><<<detail::blocks(n), detail::TPB>>>(Hdd.raw(), n, damping, damping_identity ? 1 : 0);
    detail::sync();
    return gr_dense_cholesky_solve(dt, (int64_t)n, Hdd.raw(), (int64_t)n, graph->get_b().raw(), x, dev, nullptr, nullptr) == GR_OK;
  }
};

// ---- optimizer/levenberg_marquardt.hpp -----------------------------------------------------------
namespace optimizer {

template <typename T, typename S> class LevenbergMarquardtOptions {
public:
  Solver<T, S> *solver = nullptr;
  size_t iterations = 10;
  double initial_damping = 1e-4;
  uint8_t optimization_level = 0;
  bool verbose = false;
  bool *stop_flag = nullptr;
  bool use_identity = false;
  StreamPool *streams = nullptr;
  bool validate() const {
    if (!solver) { if (verbose) std::cerr << "Levenberg-Marquardt options invalid: solver is null" << std::endl; return false; }
    if (!streams) { if (verbose) std::cerr << "Levenberg-Marquardt options invalid: streams is null" << std::endl; return false; }
    return true;
  }
};

// :20-47
template <typename T, typename S> T compute_rho(Graph<T, S> *graph, const T *delta_x, T chi2, T new_chi2, T mu, bool step_is_good) {
  T num = chi2 - new_chi2, denom = 1;
  if (step_is_good) {
    const size_t n = graph->get_hessian_dimension();
    static hbm_vector<T> *scratch = new hbm_vector<T>(detail::DOT_SCRATCH); // lives as long as the process (no hipFree at exit)
    denom = detail::dot<T, true>(delta_x, graph->get_b().raw(), n, scratch->raw(), mu);
    denom += T(1.0e-3);
  }
  return num / denom;
}

// :110-242 (EARLY = false) and :255-418 (EARLY = true: levenberg_marquardt2)
namespace detail {
// Bundle-adjustment graphs built with the generic descriptors — one camera descriptor (9), one point descriptor (3,
// set_eliminate or not), one factor descriptor whose error()/jacobian() ARE the engine's reprojection model (verified by
// value on sampled factors, below; the `bal_reprojection_model` tag is optional) — are optimised by the
// specialised engine of libgraphite_mi355x.so (gr_bal_*: matrix-free kernels, device-resident loop) instead of
// the generic stored-Jacobian kernels: same algorithm, same trace.  Returns false when the graph or the solver
// is anything else (unused vertices, inactive factors, precision matrices, a solver without an engine
// counterpart, GRAPHITE_GENERIC_ONLY=1): the caller then runs the generic loop.
// number of optimiser calls of this process that ran on the gr_bal engine (tests assert which path ran)
inline size_t &engine_handovers() { static size_t n = 0; return n; }
// ... and how many of them found their engine problem already built (EngineCache hit: parameters uploaded, nothing else)
inline size_t &engine_cache_hits() { static size_t n = 0; return n; }
// ... and how many ran the engine's kernels instantiated on the user's traits (engine_model.hpp) rather than the built-in model
inline size_t &engine_model_handovers() { static size_t n = 0; return n; }
// host seconds the last hand-over spent outside gr_bal_levenberg_marquardt (checks, export, probe, create / cache look-up,
// parameter transfer both ways, residual refresh)
inline double &engine_last_setup_seconds() { static double s = 0; return s; }
// ... and inside the engine's LM loop (gr_lm_stats.loop_seconds) / its iteration count
inline double &engine_last_loop_seconds() { static double s = 0; return s; }
inline int &engine_last_iterations() { static int n = 0; return n; }

// The engine problem of one graph, kept on the Graph between optimiser calls (Graph::engine_cache).  Valid while the three
// descriptors are the same objects, none of their structure epochs moved (add / remove / replace / set_fixed / set_active
// / set_eliminate / clear all bump one), the optimisation level is the same and the digests of their public arrays (vertex
// addresses and state bytes; factor ids, observations, activity, precision matrices, losses) are unchanged.
template <typename T> struct EngineCache {
  gr_bal_problem *prob = nullptr;
  const void *cd = nullptr, *pd = nullptr, *fd = nullptr;
  size_t epoch_c = 0, epoch_p = 0, epoch_f = 0;
  uint64_t digest_c = 0, digest_p = 0, digest_f = 0;
  uint8_t level = 0;
  int device = 0;
  size_t n_factors = 0;
  // vertices no active factor touches are not part of the engine problem: used[k] = local vertex id of engine vertex k
  // (empty = every vertex is used, identity)
  std::vector<uint32_t> cam_used, pt_used;
  managed_vector<T> cams, pts;      // all vertices of the descriptors, local order (gather / scatter staging)
  std::vector<T> ecams, epts;       // the engine's subset when it is not the identity
  // USER-TRAITS engine (engine_model.hpp): the kernels instantiated on the user's traits + their engine-order vertex copies and
  // factor streams; null = the library's built-in camera model was verified and runs (the parameter arrays above travel)
  std::shared_ptr<graphite::detail::EngineModelBase> model;
  ~EngineCache() { if (prob) gr_bal_destroy(prob); } // before `model`: the problem calls into it
};

// optimiser calls that ran on the pose-graph engine (engine_pose.hpp)
inline size_t &pose_engine_handovers() { static size_t n = 0; return n; }
// A graph of ONE vertex descriptor and ONE binary factor descriptor between its vertices, solved by PCGSolver with the block-Jacobi
// or identity preconditioner: the pose-graph engine (engine_pose.hpp, kernels instantiated on the user's traits).  false: the graph
// stays on the generic kernels (the reason under GR_VERBOSE).
template <typename T, typename S>
bool pose_engine_levenberg_marquardt(Graph<T, S> *graph, LevenbergMarquardtOptions<T, S> *options, bool early_stop, bool &result) {
  if (getenv("GRAPHITE_POSE_ENGINE") && atoi(getenv("GRAPHITE_POSE_ENGINE")) == 0) return false;
  auto *vd = graph->get_vertex_descriptors()[0];
  const bool verbose_why = getenv("GR_VERBOSE") != nullptr;
  auto why = [&](const std::string &w) { if (verbose_why) std::cerr << "[graphite] pose-graph engine: " << w << "; using the generic kernels" << std::endl; return false; };
  for (auto *fd : graph->get_factor_descriptors()) {
    if (fd->num_slots() > 2) return why("a factor descriptor with more than two slots");
    for (size_t sl = 0; sl < fd->num_slots(); ++sl) if (fd->slot_descriptor(sl) != vd) return why("a factor slot that is not the graph's vertex descriptor");
  }
  if (vd->eliminate) return why("the vertex descriptor is marked for elimination");
  const int kind = options->solver->engine_kind(vd->count());
  if (kind != GR_SOLVER_PCG && kind != GR_SOLVER_PCG_IDENTITY) return why("the solver is not PCGSolver with the block-Jacobi or identity preconditioner");
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  if (!graph->initialize_optimization(options->optimization_level, /*light=*/true)) return false;
  graphite::detail::PoseEngineOptions o;
  o.iterations = options->iterations; o.initial_damping = (double)options->initial_damping; o.use_identity = options->use_identity;
  int m = 0; double tol = 0, rej = 0;
  options->solver->engine_pcg_parameters(m, tol, rej);
  o.pcg_max_iter = m; o.pcg_tol = tol; o.pcg_rej = rej; o.identity_precond = kind == GR_SOLVER_PCG_IDENTITY;
  o.early_stop = early_stop; o.scale_system = graph->scales_system(); o.stop_flag = options->stop_flag;
  graphite::detail::PoseEngineResult r;
  int rc;
  {
    typename Graph<T, S>::DeviceMirrorScope mirror_scope(graph); // written back into the user's vertices on the way out
    rc = graphite::detail::pose_engine_run(graph, o, r);
    if (rc == 0) { graph->compute_error(); graphite::detail::sync(); } // the residuals of the optimised vertices, as the generic loop leaves them
  }
  if (rc < 0) return why(r.declined);
  if (verbose_why) std::cerr << "[graphite] pose-graph engine: set-up " << 1e3 * r.setup_seconds << " ms (of which " << r.detail << "), loop " << 1e3 * r.loop_seconds << " ms, "
                             << r.iterations_run << " LM iterations, " << r.pcg_iterations << " PCG iterations" << std::endl;
  if (rc > 0) { std::cerr << "graphite: pose-graph engine: " << r.declined << "; the graph's vertices are unchanged, using the generic kernels" << std::endl; return false; }
  ++pose_engine_handovers();
  ++engine_model_handovers(); // (kernels instantiated on the user's traits: counted there too)
  const double total = std::chrono::duration<double>(clk::now() - t0).count();
  engine_last_setup_seconds() = total - r.loop_seconds;
  engine_last_loop_seconds() = r.loop_seconds;
  engine_last_iterations() = r.iterations_run;
  if (options->verbose) { // levenberg_marquardt.hpp:216-221; the set-up is in the first row's total, as the reference's is
    const int prec = early_stop ? 4 : 12, w0 = early_stop ? 10 : 18, w = early_stop ? 16 : 24;
    std::cout << std::setprecision(12) << std::setw(18) << "Iteration" << std::setw(24) << "Initial Chi2" << std::setw(24)
              << "Current Chi2" << std::setw(24) << "Lambda" << std::setw(24) << "Time" << std::setw(24) << "Total Time" << std::endl;
    std::cout << std::string(138, '-') << std::endl;
    double run = 0;
    for (double sec : r.seconds) run += sec;
    run = total - run;
    for (int i = 0; i < r.iterations_run; ++i) {
      run += r.seconds[i];
      std::cout << std::setprecision(prec) << std::setw(w0) << i << std::setw(w) << (T)r.chi2[i] << std::setw(w) << (T)r.chi2[i + 1] << std::setw(w)
                << (T)r.lambda[i + 1] << std::setw(w) << r.seconds[i] << std::setw(w) << run << std::endl;
    }
  }
  if (r.stop_bits & 1) std::cout << "Damping factor is infinite, terminating optimization" << std::endl;
  if (r.stop_bits & 2) std::cout << "Rho is zero, terminating optimization" << std::endl;
  if (r.stop_bits & 8) std::cout << "Stopping optimization due to stop flag" << std::endl;
  result = r.ok;
  return true;
}

template <typename T, typename S>
bool engine_levenberg_marquardt(Graph<T, S> *graph, LevenbergMarquardtOptions<T, S> *options, bool early_stop, bool &result) {
  if (getenv("GRAPHITE_GENERIC_ONLY") && atoi(getenv("GRAPHITE_GENERIC_ONLY")) != 0) return false;
  if (!(std::is_same<T, S>::value || (std::is_same<T, double>::value && std::is_same<S, float>::value))) {
    // Graph<T, bf16 / half>: the engine keeps no Jacobians (it recomputes them from the camera pack), so there is nothing to hold
    // in a 16-bit storage type; such graphs run on the generic kernels, which DO store S-typed Jacobians (types.hpp:8-43)
    if (getenv("GR_VERBOSE")) std::cerr << "[graphite] Jacobian storage type of " << sizeof(S) << " bytes under a " << sizeof(T)
                                        << "-byte graph: not an engine configuration (no stored Jacobians there); using the generic kernels" << std::endl;
    return false;
  }
  auto &vds = graph->get_vertex_descriptors();
  auto &fds = graph->get_factor_descriptors();
  if (vds.size() == 1 && !fds.empty()) return pose_engine_levenberg_marquardt(graph, options, early_stop, result); // (no BAL shape has one vertex descriptor)
  if (vds.size() != 2 || fds.size() != 1 || fds[0]->num_slots() != 2) return false;
  auto *cd = fds[0]->slot_descriptor(0), *pd = fds[0]->slot_descriptor(1);
  if (cd == pd || cd->dimension() > 9 || pd->dimension() > 3 || fds[0]->error_dimension() > 2 || cd->eliminate) return false;
  const bool bal_dims = cd->dimension() == 9 && pd->dimension() == 3 && fds[0]->error_dimension() == 2;
  // GRAPHITE_ENGINE=model: run even a graph whose traits ARE the built-in model on the kernels instantiated from them (tests, A/B)
  const bool force_model = getenv("GRAPHITE_ENGINE") && std::string(getenv("GRAPHITE_ENGINE")) == "model";
  if (!((vds[0] == cd && vds[1] == pd) || (vds[0] == pd && vds[1] == cd))) return false;
  const int kind = options->solver->engine_kind(cd->count());
  if (kind < 0) return false;
  using clk = std::chrono::steady_clock;
  const bool timing = getenv("GR_VERBOSE") != nullptr;
  const auto tt0 = clk::now();
  auto lap = [&, last = tt0](const char *what) mutable {
    if (!timing) return;
    graphite::detail::sync();
    const auto now = clk::now();
    std::cerr << "[graphite] hand-over: " << what << " " << std::chrono::duration<double, std::milli>(now - last).count() << " ms" << std::endl;
    last = now;
  };
  int dev = 0;
  GRAPHITE_HIP(hipGetDevice(&dev));
  const gr_dtype dt = sizeof(T) == 8 ? GR_F64 : GR_F32;

  // ---- the cached engine problem of this graph, if the graph is still the one it was built from -----------------------
  using Cache = EngineCache<T>;
  std::shared_ptr<Cache> cache = std::static_pointer_cast<Cache>(graph->engine_cache);
  const uint64_t dg_c = cd->content_fingerprint(), dg_p = pd->content_fingerprint(), dg_f = fds[0]->content_fingerprint();
  bool hit = cache && cache->prob && cache->cd == cd && cache->pd == pd && cache->fd == fds[0] && cache->epoch_c == cd->structure_epoch &&
             cache->epoch_p == pd->structure_epoch && cache->epoch_f == fds[0]->structure_epoch && cache->level == options->optimization_level &&
             cache->device == dev && cache->digest_c == dg_c && cache->digest_p == dg_p && cache->digest_f == dg_f &&
             graph->last_init_level == options->optimization_level;
  lap(hit ? "cache look-up (epochs + digests): HIT" : "cache look-up (epochs + digests): miss");
  auto fail = [&](const char *what) {
    std::cerr << "graphite: engine hand-over failed in " << what << ": " << gr_last_error_string() << "; using the generic kernels" << std::endl;
    graph->engine_cache.reset();
    return false;
  };
  if (!hit) {
    graph->engine_cache.reset();
    cache.reset();
    // light: local ids, active list, vertex states and Hessian columns — what the checks, the probe and the export below read;
    // the Jacobian storage and gather lists of the generic kernels are only built when the graph stays with them (lm_loop)
    if (!graph->initialize_optimization(options->optimization_level, /*light=*/true)) return false;
    lap("initialize_optimization (light)");
    // fixed vertices (bit 0) go to the engine as masks (gr_bal_set_fixed); a vertex no active factor touches (bit 7: outlier
    // rejection or a level mask took its factors away, active.hpp:18-21) is simply not part of the engine problem
    auto fresh = std::make_shared<Cache>();
    std::vector<unsigned char> cam_fixed, pt_fixed;
    bool any_fixed = false;
    std::vector<int32_t> cam_new(cd->count(), -1), pt_new(pd->count(), -1);
    for (auto *vd : {cd, pd}) {
      const uint8_t *st = vd->get_active_state();
      std::vector<unsigned char> &mask = vd == cd ? cam_fixed : pt_fixed;
      std::vector<uint32_t> &used = vd == cd ? fresh->cam_used : fresh->pt_used;
      std::vector<int32_t> &renum = vd == cd ? cam_new : pt_new;
      for (size_t v = 0; v < vd->count(); ++v) {
        if (st[v] & ~uint8_t(0x81)) return false; // a state this layer does not know
        if (st[v] & 0x80) continue;
        renum[v] = (int32_t)used.size();
        used.push_back((uint32_t)v);
        mask.push_back(st[v] & 0x1);
        any_fixed = any_fixed || (st[v] & 0x1);
      }
      if (used.empty()) return false;
      if (used.size() == vd->count()) used.clear(); // identity
    }
    std::vector<int32_t> ci, pi;
    std::vector<T> obs;
    int loss_kind = 0; double loss_delta = 0;
    const size_t nc = fresh->cam_used.empty() ? cd->count() : fresh->cam_used.size(), np = fresh->pt_used.empty() ? pd->count() : fresh->pt_used.size();
    // (which path a graph takes is reported once both attempts below have decided — ADVICE r5: a step that fails here says what failed,
    // not where the graph ends up)
    std::string declined;
    auto bad = [&](const char *what) { declined = std::string(what) + " failed: " + gr_last_error_string(); std::cerr << "graphite: engine hand-over: " << declined << std::endl; return false; };
    // the library's built-in camera model, when the user's traits are verified to BE it (by value, below)
    auto try_builtin = [&]() -> bool {
    if (!bal_dims || force_model) return false;
    if (!fds[0]->export_bal(ci, pi, obs, loss_kind, loss_delta)) return false; // the ACTIVE factors
    if (!fresh->cam_used.empty() || !fresh->pt_used.empty())
      for (size_t f = 0; f < ci.size(); ++f) { ci[f] = cam_new[ci[f]]; pi[f] = pt_new[pi[f]]; }
    lap("fixed masks + export_bal");
    // VERIFIED hand-over: the engine is specialised for one function (gr_bal_model_evaluate).  The user's own
    // parameters() / error() / jacobian() (or dual-number Jacobian) / update() are evaluated on up to 256 of the graph's
    // factors AND on synthetic triples that reach the branches a sample does not (theta == 0, the point behind / in front of
    // the camera, a large radial term, a large rotation) and must reproduce it — 1e-10 (fp64) / 1e-4 (fp32) of the block's
    // magnitude; update() must be plain addition.  No tag is needed; a factor that carries the bal_reprojection_model tag
    // but computes something else is reported and stays on the generic kernels.  Verified once per cached problem: the
    // functions are properties of the traits TYPE.
    {
      std::vector<T> pc, pp, po, ur, uJc, uJp;
      double update_dev = 0;
      size_t nsyn = 0;
      if (!fds[0]->probe_bal(256, pc, pp, po, ur, uJc, uJp, update_dev, &nsyn)) return false;
      const size_t ns = pc.size() / 9;
      std::vector<T> er(2 * ns), eJc(18 * ns), eJp(6 * ns);
      if (gr_bal_model_evaluate(dt, (int64_t)ns, pc.data(), pp.data(), po.data(), er.data(), eJc.data(), eJp.data(), dev, nullptr) != GR_OK) {
        return bad("gr_bal_model_evaluate");
      }
      const double tol = sizeof(T) == 8 ? 1e-10 : 1e-4;
      double worst = 0, worst_syn = 0;
      auto block_dev = [&](const T *u, const T *e, size_t len, double floor_mag) {
        double mag = floor_mag, d = 0;
        for (size_t k = 0; k < len; ++k) mag = std::max(mag, std::abs((double)e[k]));
        for (size_t k = 0; k < len; ++k) { const double x = std::abs((double)u[k] - (double)e[k]); d = (x > d || x != x) ? x : d; }
        return d / mag;
      };
      for (size_t a = 0; a < ns; ++a) {
        const double omag = std::max({1.0, std::abs((double)po[2 * a]), std::abs((double)po[2 * a + 1])});
        double &w = a + nsyn < ns ? worst : worst_syn;
        for (double d : {block_dev(&ur[2 * a], &er[2 * a], 2, omag), block_dev(&uJc[18 * a], &eJc[18 * a], 18, 1e-300), block_dev(&uJp[6 * a], &eJp[6 * a], 6, 1e-300)})
          w = (d > w || d != d) ? d : w;
      }
      const bool model_ok = worst <= tol && worst_syn <= tol, update_ok = update_dev <= (sizeof(T) == 8 ? 1e-14 : 1e-6);
      if (getenv("GR_VERBOSE"))
        std::cerr << "[graphite] engine hand-over probe: " << ns - nsyn << " factors + " << nsyn << " synthetic branch triples, max deviation of error()/jacobian() from the engine's model "
                  << worst << " / " << worst_syn << " (bar " << tol << "), of update() from addition " << update_dev << std::endl;
      if (!model_ok || !update_ok) {
        if (fds[0]->declares_bal_model())
          std::cerr << "graphite: the factor traits declare bal_reprojection_model, but " << (model_ok ? "update()" : "error()/jacobian()")
                    << " differ from the engine's model (relative deviation " << (model_ok ? update_dev : std::max(worst, worst_syn))
                    << (worst <= tol && !(worst_syn <= tol) ? ", on the synthetic branch triples only" : "")
                    << "); not the library's built-in camera model: trying the engine's kernels instantiated on these traits (engine_model.hpp)" << std::endl;
        return false;
      }
    }
    lap("probe");
    fresh->cams.resize(9 * cd->count()); fresh->pts.resize(3 * pd->count());
    cd->gather_parameters(fresh->cams.raw()); pd->gather_parameters(fresh->pts.raw());
    graphite::detail::sync();
    const T *c_src = fresh->cams.raw(), *p_src = fresh->pts.raw();
    if (!fresh->cam_used.empty()) {
      fresh->ecams.resize(9 * nc);
      for (size_t k = 0; k < nc; ++k) std::copy(c_src + 9 * (size_t)fresh->cam_used[k], c_src + 9 * (size_t)fresh->cam_used[k] + 9, fresh->ecams.begin() + 9 * k);
      c_src = fresh->ecams.data();
    }
    if (!fresh->pt_used.empty()) {
      fresh->epts.resize(3 * np);
      for (size_t k = 0; k < np; ++k) std::copy(p_src + 3 * (size_t)fresh->pt_used[k], p_src + 3 * (size_t)fresh->pt_used[k] + 3, fresh->epts.begin() + 3 * k);
      p_src = fresh->epts.data();
    }
    lap("gather_parameters");
    if (gr_bal_create(&fresh->prob, dt, (int64_t)nc, (int64_t)np, (int64_t)ci.size(), c_src, p_src, obs.data(), ci.data(), pi.data(), dev, nullptr) != GR_OK) return bad("gr_bal_create");
    if (gr_bal_set_loss(fresh->prob, loss_kind ? GR_LOSS_HUBER : GR_LOSS_DEFAULT, loss_delta) != GR_OK) return bad("gr_bal_set_loss");
    return true;
    };
    // ... otherwise the engine's kernels instantiated on the user's traits (engine_model.hpp): per-factor precision matrices and
    // losses, constraint data, any (<= 9, <= 3) -> <= 2 error / jacobian / update
    auto try_model = [&]() -> bool {
      if (fresh->prob) { gr_bal_destroy(fresh->prob); fresh->prob = nullptr; }
      if (kind == GR_SOLVER_PCG_SCHUR_IMPLICIT) { declined = "the implicit Schur form recomputes the built-in Jacobian"; return false; } // beyond the explicit form's camera count: generic kernels
      ci.clear(); pi.clear();
      fresh->model = fds[0]->make_engine_model(ci, pi, nc, np);
      if (!fresh->model) { declined = "traits outside the engine's model (dimensions beyond (9, 3) -> 2, non-plain-data types, or a precision matrix that is not symmetric positive semi-definite)"; return false; }
      if (!fresh->cam_used.empty() || !fresh->pt_used.empty())
        for (size_t f = 0; f < ci.size(); ++f) { ci[f] = cam_new[ci[f]]; pi[f] = pt_new[pi[f]]; }
      if (gr_bal_create_model(&fresh->prob, dt, (int64_t)nc, (int64_t)np, (int64_t)ci.size(), ci.data(), pi.data(), fresh->model->ops(), dev, nullptr) != GR_OK) return bad("gr_bal_create_model");
      std::vector<int32_t> obs_order(ci.size()), lm_order(np);
      if (gr_bal_model_orders(fresh->prob, obs_order.data(), lm_order.data()) != GR_OK) return bad("gr_bal_model_orders");
      fresh->model->bind(fresh->cam_used, fresh->pt_used, obs_order, lm_order);
      lap("user-traits engine: export + gr_bal_create_model + factor streams");
      return true;
    };
    if (!try_builtin()) {
      fresh->model.reset();
      if (!try_model()) {
        if (fresh->prob) { gr_bal_destroy(fresh->prob); fresh->prob = nullptr; } // (it holds the model's launcher table: before the model goes)
        fresh->model.reset();
        if (getenv("GR_VERBOSE") || !declined.empty())
          std::cerr << "graphite: engine hand-over declined" << (declined.empty() ? "" : " (") << declined << (declined.empty() ? "" : ")") << ": this graph runs on the generic kernels" << std::endl;
        return false;
      }
      if (getenv("GR_VERBOSE")) std::cerr << "[graphite] engine hand-over: the engine's kernels instantiated on the user's traits (engine_model.hpp)" << std::endl;
    } else if (getenv("GR_VERBOSE")) std::cerr << "[graphite] engine hand-over: the library's built-in camera model" << std::endl;
    if (any_fixed && gr_bal_set_fixed(fresh->prob, cam_fixed.data(), pt_fixed.data()) != GR_OK) return bad("gr_bal_set_fixed");
    fresh->cd = cd; fresh->pd = pd; fresh->fd = fds[0];
    fresh->epoch_c = cd->structure_epoch; fresh->epoch_p = pd->structure_epoch; fresh->epoch_f = fds[0]->structure_epoch;
    // initialize_optimization rewrote the bit-7 state bytes: the digests are taken of what the NEXT call will see
    fresh->digest_c = cd->content_fingerprint(); fresh->digest_p = pd->content_fingerprint(); fresh->digest_f = fds[0]->content_fingerprint();
    fresh->level = options->optimization_level; fresh->device = dev; fresh->n_factors = ci.size();
    cache = fresh;
    graph->engine_cache = cache;
    if (cache->model) cache->model->upload_vertices();
    lap("gr_bal_create + settings");
  } else if (cache->model) {
    cache->model->upload_vertices(); // unchanged structure: only the vertex VALUES travel
    ++engine_cache_hits();
    lap("user-traits engine: vertex upload");
  } else {
    // unchanged structure: only the vertex VALUES travel
    cd->gather_parameters(cache->cams.raw()); pd->gather_parameters(cache->pts.raw());
    graphite::detail::sync();
    const T *c_src = cache->cams.raw(), *p_src = cache->pts.raw();
    if (!cache->cam_used.empty()) {
      for (size_t k = 0; k < cache->cam_used.size(); ++k) std::copy(c_src + 9 * (size_t)cache->cam_used[k], c_src + 9 * (size_t)cache->cam_used[k] + 9, cache->ecams.begin() + 9 * k);
      c_src = cache->ecams.data();
    }
    if (!cache->pt_used.empty()) {
      for (size_t k = 0; k < cache->pt_used.size(); ++k) std::copy(p_src + 3 * (size_t)cache->pt_used[k], p_src + 3 * (size_t)cache->pt_used[k] + 3, cache->epts.begin() + 3 * k);
      p_src = cache->epts.data();
    }
    if (gr_bal_set_params(cache->prob, c_src, p_src) != GR_OK) return fail("gr_bal_set_params");
    ++engine_cache_hits();
    lap("gather_parameters + gr_bal_set_params");
  }
  gr_bal_problem *prob = cache->prob;
  if (gr_bal_set_scale_system(prob, graph->scales_system() ? 1 : 0) != GR_OK) return fail("gr_bal_set_scale_system");
  if (!cache->model && gr_bal_set_jacobian_precision(prob, std::is_same<T, S>::value ? dt : GR_F32) != GR_OK) return fail("gr_bal_set_jacobian_precision");
  gr_lm_options o{};
  o.solver = kind; o.iterations = (int32_t)options->iterations; o.initial_damping = options->initial_damping;
  o.use_identity = options->use_identity ? 1 : 0; o.early_stop = early_stop ? 1 : 0;
  o.stop_flag = reinterpret_cast<const volatile unsigned char *>(options->stop_flag); // polled per iteration by the engine
  int m; double tl, rj;
  options->solver->engine_pcg_parameters(m, tl, rj);
  o.pcg_max_iter = m; o.pcg_tol = tl; o.pcg_rejection_ratio = rj;
  // GR_PROFILE_KERNELS=1 (measurement: bench.py's user-traits entry, tools/): per-kernel dispatch times of this call, printed below
  const bool profile_kernels = getenv("GR_PROFILE_KERNELS") && atoi(getenv("GR_PROFILE_KERNELS")) != 0;
  o.profile = profile_kernels ? 1 : 0;
  gr_lm_stats st{};
  std::vector<double> chi2(options->iterations + 1), lambda(options->iterations + 1);
  if (getenv("GR_VERBOSE")) std::cerr << "[graphite] bundle-adjustment graph (" << cd->count() << " cameras, " << pd->count() << " points, " << cache->n_factors
                                      << " active factors) handed to the gr_bal engine" << (cache->model ? " (kernels instantiated on the user's traits)" : " (built-in camera model)")
                                      << ", gr_solver " << kind << (hit ? " (cached problem)" : "") << std::endl;
  if (cache->model) ++engine_model_handovers();
  const auto t_lm0 = clk::now();
  if (gr_bal_levenberg_marquardt(prob, &o, &st, chi2.data(), lambda.data()) != GR_OK) return fail("gr_bal_levenberg_marquardt");
  const double lm_seconds = std::chrono::duration<double>(clk::now() - t_lm0).count();
  ++engine_handovers();
  engine_last_loop_seconds() = st.loop_seconds; engine_last_iterations() = st.iterations_run;
  if (profile_kernels) {
    gr_kernel_stat ks[64];
    int nk = 0;
    if (gr_bal_kernel_stats(prob, ks, 64, &nk) == GR_OK)
      for (int k = 0; k < std::min(nk, 64); ++k)
        std::cout << "KERNEL " << ks[k].name << " launches " << ks[k].launches << " active " << ks[k].active_launches << " total_ms " << std::setprecision(9) << ks[k].total_ms
                  << " bytes_per_launch " << ks[k].bytes_per_launch << " flops_per_launch " << ks[k].flops_per_launch << std::endl;
    std::cout << "LM_LOOP_SECONDS " << st.loop_seconds << " ITERATIONS " << st.iterations_run << " PCG_ITERATIONS " << st.pcg_iterations << std::endl;
  }
  lap("gr_bal_levenberg_marquardt");
  if (cache->model) cache->model->download_vertices();
  else {
    T *c_dst = cache->cam_used.empty() ? cache->cams.raw() : cache->ecams.data(), *p_dst = cache->pt_used.empty() ? cache->pts.raw() : cache->epts.data();
    if (gr_bal_get_params(prob, c_dst, p_dst) != GR_OK) return fail("gr_bal_get_params");
    // vertices outside the engine problem keep the values gathered above
    for (size_t k = 0; k < cache->cam_used.size(); ++k) std::copy(cache->ecams.begin() + 9 * k, cache->ecams.begin() + 9 * k + 9, cache->cams.raw() + 9 * (size_t)cache->cam_used[k]);
    for (size_t k = 0; k < cache->pt_used.size(); ++k) std::copy(cache->epts.begin() + 3 * k, cache->epts.begin() + 3 * k + 3, cache->pts.raw() + 3 * (size_t)cache->pt_used[k]);
  }
  if (!cache->model) { cd->scatter_parameters(cache->cams.raw()); pd->scatter_parameters(cache->pts.raw()); }
  lap("get_params + scatter_parameters");
  {
    // leave the residuals of the optimised vertices behind, as the generic loop does (graph->chi2() is valid).  Through the HBM
    // mirror of the vertex values: dereferencing the user's (pinned host) vertices per factor is 65 MB of PCIe reads on
    // Ladybug-1723 (1.7 ms); mirrored in and out they are 4 MB each way
    typename Graph<T, S>::DeviceMirrorScope mirror_scope(graph);
    graph->compute_error();
    graphite::detail::sync();
  }
  lap("compute_error");
  const double total = std::chrono::duration<double>(clk::now() - tt0).count();
  engine_last_setup_seconds() = total - lm_seconds;
  if (options->verbose) {
    // the reference's table (levenberg_marquardt.hpp:216-221): every row carries ITS iteration's time; what the hand-over
    // cost before the loop is in the first row's total, as the reference's set-up is
    std::vector<double> it_s(std::max(1, st.iterations_run), 0.0);
    int nit = 0;
    (void)gr_bal_lm_iteration_seconds(prob, it_s.data(), (int)it_s.size(), &nit);
    double run = total - st.loop_seconds;
    const int prec = early_stop ? 4 : 12, w0 = early_stop ? 10 : 18, w = early_stop ? 16 : 24;
    std::cout << std::setprecision(12) << std::setw(18) << "Iteration" << std::setw(24) << "Initial Chi2" << std::setw(24)
              << "Current Chi2" << std::setw(24) << "Lambda" << std::setw(24) << "Time" << std::setw(24) << "Total Time" << std::endl;
    std::cout << std::string(138, '-') << std::endl;
    for (int i = 0; i < st.iterations_run; ++i) {
      run += it_s[i];
      std::cout << std::setprecision(prec) << std::setw(w0) << i << std::setw(w) << chi2[i] << std::setw(w) << chi2[i + 1] << std::setw(w)
                << lambda[i + 1] << std::setw(w) << it_s[i] << std::setw(w) << run << std::endl;
    }
  }
  result = st.ok != 0;
  return true;
}

template <bool EARLY, typename T, typename S> bool lm_loop(Graph<T, S> *graph, LevenbergMarquardtOptions<T, S> *options) {
  if (!options->validate()) return false;
  {
    bool result = false;
    if (engine_levenberg_marquardt(graph, options, EARLY, result)) return result;
  }
  using clk = std::chrono::steady_clock;
  auto start = clk::now();
  if (!graph->initialize_optimization(options->optimization_level)) return false;
  typename Graph<T, S>::DeviceMirrorScope mirror_scope(graph); // written back into the user's vertices on every way out
  graph->build_structure();
  StreamPool &streams = *options->streams;
  Solver<T, S> *solver = options->solver;
  T mu = static_cast<T>(options->initial_damping), nu = 2;
  solver->update_structure(graph, streams);
  graph->linearize(streams);
  solver->update_values(graph, streams);
  T chi2 = graph->chi2();
  hbm_vector<T> delta_x(graph->get_hessian_dimension());
  bool run = true;
  int num_bad = 0;
  constexpr int prec = EARLY ? 4 : 12, w0 = EARLY ? 10 : 18, w = EARLY ? 16 : 24; // table rows of :216-221 / :382-387
  if (options->verbose) {
    std::cout << std::setprecision(12) << std::setw(18) << "Iteration" << std::setw(24) << "Initial Chi2" << std::setw(24)
              << "Current Chi2" << std::setw(24) << "Lambda" << std::setw(24) << "Time" << std::setw(24) << "Total Time" << std::endl;
    std::cout << std::string(138, '-') << std::endl;
  }
  double time = std::chrono::duration<double>(clk::now() - start).count();
  for (size_t i = 0; i < options->iterations && run; ++i) {
    auto t0 = clk::now();
    solver->set_damping_factor(graph, mu, options->use_identity, streams);
    const bool solve_ok = solver->solve(graph, delta_x.raw(), streams);
    graph->backup_parameters();
    graph->apply_update(delta_x.raw(), streams);
    graph->compute_error();
    T new_chi2 = graph->chi2();
    if (!solve_ok) new_chi2 = std::numeric_limits<T>::max();
    const T rho = compute_rho(graph, delta_x.raw(), chi2, new_chi2, mu, solve_ok);
    const T chi2_before = chi2;
    bool step_accepted = false;
    if (solve_ok && std::isfinite(new_chi2) && rho > 0) {
      step_accepted = true;
      double alpha = 1.0 - std::pow(2.0 * rho - 1.0, 3);
      alpha = std::max(std::min(alpha, 2.0 / 3.0), 1.0 / 3.0);
      mu *= static_cast<T>(alpha);
      nu = 2;
      chi2 = new_chi2;
      graph->linearize(streams);
      solver->update_values(graph, streams);
    } else {
      graph->revert_parameters();
      graph->compute_error();
      (void)graph->chi2();
      mu *= nu;
      nu *= 2;
      new_chi2 = chi2;
    }
    const double it_time = std::chrono::duration<double>(clk::now() - t0).count();
    time += it_time;
    if (options->verbose)
      std::cout << std::setprecision(prec) << std::setw(w0) << i << std::setw(w) << chi2_before << std::setw(w) << new_chi2 << std::setw(w)
                << mu << std::setw(w) << it_time << std::setw(w) << time << std::endl;
    if (!std::isfinite(mu)) { std::cout << "Damping factor is infinite, terminating optimization" << std::endl; run = false; }
    if (rho == 0) { std::cout << "Rho is zero, terminating optimization" << std::endl; break; }
    if (options->stop_flag && *options->stop_flag) { std::cout << "Stopping optimization due to stop flag" << std::endl; break; }
    if (EARLY && step_accepted) { // :404-414
      if (((chi2_before - new_chi2) * 1.0e3) < chi2_before) num_bad++;
      else num_bad = 0;
      if (num_bad >= 3) break;
    }
  }
  return run;
}
} // namespace detail
// optimiser calls of this process that ran on the hand-written gr_bal engine (the others ran on the generic kernels)
inline size_t engine_handover_count() { return detail::engine_handovers(); }
// ... of which: calls that found the graph's engine problem cached (structure unchanged since the previous call)
inline size_t engine_cache_hit_count() { return detail::engine_cache_hits(); }
// ... of which: calls whose per-observation kernels were instantiated on the user's traits (user-traits engine)
inline size_t engine_model_handover_count() { return detail::engine_model_handovers(); }
// optimiser calls of this process that ran on the pose-graph engine (engine_pose.hpp: one vertex descriptor, binary factors between its vertices)
inline size_t pose_engine_handover_count() { return detail::pose_engine_handovers(); }
// host seconds the last hand-over spent around gr_bal_levenberg_marquardt (checks, export, probe, create or cache look-up, transfers)
inline double engine_last_setup_seconds() { return detail::engine_last_setup_seconds(); }
// seconds inside the engine's LM loop of the last hand-over, and the LM iterations it ran
inline double engine_last_loop_seconds() { return detail::engine_last_loop_seconds(); }
inline int engine_last_iterations() { return detail::engine_last_iterations(); }
template <typename T, typename S> bool levenberg_marquardt(Graph<T, S> *graph, LevenbergMarquardtOptions<T, S> *options) {
  return detail::lm_loop<false>(graph, options);
}
template <typename T, typename S> bool levenberg_marquardt2(Graph<T, S> *graph, LevenbergMarquardtOptions<T, S> *options) {
  return detail::lm_loop<true>(graph, options);
}

} // namespace optimizer
} // namespace graphite
