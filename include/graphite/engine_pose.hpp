// graphite/engine_pose.hpp — POSE-GRAPH ENGINE: levenberg_marquardt + PCGSolver on a graph whose factors are unary or BINARY between vertices
// of ONE descriptor (pose–pose "between" factors and priors: the SLAM back-end shape of the reference's README.md:27), device-resident, gfx950.
//
// The generic kernels (core.hpp / solve.hpp) follow the reference launch by launch: per LM iteration ~25 launches and several
// host round trips for the linearisation, ~9 launches per PCG iteration (1.0 ms per LM iteration on 10 k poses / 48.6 k factors,
// 45 us per PCG iteration).  This engine runs the SAME algorithm — optimizer/levenberg_marquardt.hpp:110-242 (and :255-418),
// solver/pcg.hpp:61-232, preconditioner/block_jacobi.hpp:79-186 or identity.hpp, Graph::linearize graph.hpp:236-290 — as
// TWO launches per LM iteration and no host round trip at all:
//
//   k_pe_solve  (whole grid resident; 1, 2, 4 or 8 lanes per vertex, four waves per workgroup)
//       assemble (after an accepted step): every entry of a vertex gathers its side of its factor's record; per-vertex sums in fixed
//       order -> the vertex's Hessian block, gradient, column scales 1 / (eps + sqrt(h_cc)), b = -s.g, clamped scaled diagonal
//       damp + invert the block (block-Jacobi), start the PCG, run ALL its iterations with two grid rendezvous each, then apply
//       the trial step (backup or restore-then-update of the vertex through Traits::update) and leave the rho-denominator partials
//   k_pe_factor  (one factor per lane) Traits::error at the trial point, chi2 = rho(r^T P r), rho', Traits::jacobian or dual numbers,
//       W = rho' P and the factor's RECORD — per side  J_v^T W J_v | J_v^T W r | J_v^T W J_other — staged through LDS and written
//       lane-consecutive into the trial point's record set; the LAST workgroup to finish adds the chi2 partials in fixed order and
//       takes the LM decision (accept: the record sets swap; reject; mu, nu, stopping rules, trace)
//
// The PCG operator is applied block-sparse:  y_v = S_v H_vv S_v p_v + s_v . sum_e B_e (s.p)_nbr(e) + mu D p_v  with
// B_e = J_v^T rho' P J_nbr of factor e — the same matrix J^T rho' P J the reference applies as J^T (rho' P (J p))
// (ops/product.hpp:195,405), one gather phase instead of two.  Entries sit in a wave-sliced ELL layout (slice = the 64 vertices
// of a wave, element-major inside a group of 64): every load of a phase is coalesced except the neighbour's direction record.
// What crosses workgroups inside the launch: those records [s.z' | s.p] (48 bytes for SE(2), fp64) and the dot-product partials,
// stored write-through and read with sc1 loads (cdna_hip_programming.md Guideline 16, form R1).  The recurrence is the LAZY form
// of kernels_rp.hpp: p_k = sigma_k z'_k + beta_k p_{k-1} is formed where it is used, so beta needs no pass of its own.
//
// Fits: ONE vertex descriptor and any number of factor descriptors that are unary or binary on it (between-factors, priors; the
// reference's circle example is such a graph too), T == S, vertex / state types trivially copyable, tangent and error dimension <= 7,
// symmetric precision matrices, PCGSolver with the block-Jacobi or identity preconditioner.  Anything else stays on the generic kernels.
// GRAPHITE_POSE_ENGINE=0 switches it off.  The typed pieces (kernels on a descriptor's traits) sit behind virtuals of the base
// descriptors (core.hpp); detail::pose_engine_run below drives them.
#pragma once
#include "core.hpp"

namespace graphite {
namespace detail {

struct PoseEngineOptions {
  size_t iterations = 0;
  double initial_damping = 0;
  bool use_identity = false;      // damping form: mu I instead of mu clamp(diag)
  int pcg_max_iter = 0;
  double pcg_tol = 0, pcg_rej = 0;
  bool identity_precond = false;
  bool early_stop = false;        // levenberg_marquardt2
  bool scale_system = true;
  const volatile bool *stop_flag = nullptr;
};
struct PoseEngineResult {
  std::vector<double> chi2, lambda, seconds; // chi2[0 .. run], lambda[0 .. run], seconds[0 .. run) per iteration
  int iterations_run = 0, accepted = 0;
  long long pcg_iterations = 0;
  int stop_bits = 0;              // 1 mu not finite, 2 rho == 0, 4 early stop, 8 stop flag, 16 barrier failure
  bool ok = true;
  double setup_seconds = 0, loop_seconds = 0;
  std::string declined;           // why the descriptor is not an engine configuration (return value -1)
  std::string detail;             // set-up breakdown (GR_VERBOSE)
};

namespace pe {
constexpr int W = 64;
constexpr int WPB = 4;                 // waves per workgroup of the solve
constexpr int MAX_GRID = 1024;
constexpr int SC1 = 16;

struct Ctl {
  double mu, nu, chi2;
  int fresh;       // 1: the linearisation point moved (start / accepted step): the solve assembles first and BACKS the vertices UP;
                   // 0: the last trial was rejected: the solve restores the vertices before it applies the new step
  int stop;        // bit 0 mu not finite, bit 1 rho == 0, bit 2 early stop, bit 4 barrier failure
  int it, num_bad, accepted, cur;
  unsigned ticket;
  int last_pcg_its;
  long long pcg_its;
  double rho;
};

template <typename T> __device__ __forceinline__ T ld_x(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_x(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// NW scalars of the record at byte offset `off`, sc1 (written by another workgroup in this launch)
template <typename T, int NW> __device__ __forceinline__ void ld_rec(__amdgpu_buffer_rsrc_t r, int off, T (&v)[NW]) {
  constexpr int bytes = NW * (int)sizeof(T);
  if constexpr (bytes % 16 == 0) {
#pragma unroll
    for (int q = 0; q < bytes / 16; ++q) {
      const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16 * q, 0, SC1);
      if constexpr (sizeof(T) == 8) { const double2 d = __builtin_bit_cast(double2, a); v[2 * q] = (T)d.x; v[2 * q + 1] = (T)d.y; }
      else { const float4 f = __builtin_bit_cast(float4, a); v[4 * q] = (T)f.x; v[4 * q + 1] = (T)f.y; v[4 * q + 2] = (T)f.z; v[4 * q + 3] = (T)f.w; }
    }
  } else {
    static_assert(bytes % 8 == 0, "records are an even number of scalars");
#pragma unroll
    for (int q = 0; q < bytes / 8; ++q) {
      const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, off + 8 * q, 0, SC1);
      if constexpr (sizeof(T) == 8) v[q] = (T)__builtin_bit_cast(double, a);
      else { const float2 f = __builtin_bit_cast(float2, a); v[2 * q] = (T)f.x; v[2 * q + 1] = (T)f.y; }
    }
  }
}
template <typename T, int NW> __device__ __forceinline__ void st_rec(__amdgpu_buffer_rsrc_t r, int off, const T (&v)[NW]) {
  constexpr int bytes = NW * (int)sizeof(T);
  if constexpr (bytes % 16 == 0) {
#pragma unroll
    for (int q = 0; q < bytes / 16; ++q) {
      u32x4 a;
      if constexpr (sizeof(T) == 8) { double2 d; d.x = (double)v[2 * q]; d.y = (double)v[2 * q + 1]; a = __builtin_bit_cast(u32x4, d); }
      else { float4 f; f.x = (float)v[4 * q]; f.y = (float)v[4 * q + 1]; f.z = (float)v[4 * q + 2]; f.w = (float)v[4 * q + 3]; a = __builtin_bit_cast(u32x4, f); }
      __builtin_amdgcn_raw_buffer_store_b128(a, r, off + 16 * q, 0, SC1);
    }
  } else {
#pragma unroll
    for (int q = 0; q < bytes / 8; ++q) {
      u32x2 a;
      if constexpr (sizeof(T) == 8) a = __builtin_bit_cast(u32x2, (double)v[q]);
      else { float2 f; f.x = (float)v[2 * q]; f.y = (float)v[2 * q + 1]; a = __builtin_bit_cast(u32x2, f); }
      __builtin_amdgcn_raw_buffer_store_b64(a, r, off + 8 * q, 0, SC1);
    }
  }
}

// Sums over the whole grid, and the grid-wide rendezvous that goes with them, as ONE exchange of flagged records: every workgroup
// (one wave) stores {its partial, tag} as a single 16-byte write-through store per value — after its other exchange stores of the
// phase have been acknowledged — and then polls the records of ALL workgroups (sc1 loads, a few per lane) until each carries the
// tag.  Seeing a workgroup's record means its phase is complete and visible; the sum of the values in fixed order is the same in
// every workgroup.  One store-to-load hop instead of the store -> counter -> generation word -> partials chain of a counter barrier
// (4.5 -> 2.5 us per rendezvous at 157 workgroups).  tag = (launch << 32) | epoch: unique per rendezvous of an optimiser call (the
// record buffer is cleared at set-up); two record sets alternate by epoch parity (a workgroup overwrites its epoch-e record only
// after every workgroup has announced e + 1, i.e. finished reading e).  A grid that is not fully resident times out (2 s).
__device__ __forceinline__ void st_sum(__amdgpu_buffer_rsrc_t r, int off, double v, unsigned long long tag) {
  // the second half is tag XOR the value's bits: a reader that saw the two halves of different stores (should a 16-byte access ever be
  // split) finds no valid tag and polls again
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v), mark = tag ^ bits;
  u32x4 a;
  a.x = (unsigned)bits; a.y = (unsigned)(bits >> 32); a.z = (unsigned)mark; a.w = (unsigned)(mark >> 32);
  __builtin_amdgcn_raw_buffer_store_b128(a, r, off, 0, SC1);
}
// DRAIN: the phase published exchange records that must be acknowledged before the workgroup's record announces them.
template <int NR, bool DRAIN> __device__ __forceinline__ bool grid_sums(double (&v)[NR], __amdgpu_buffer_rsrc_t r_sum, unsigned long long launch_tag, unsigned &epoch, int *fail, double *s_red /* [WPB * 2 + 4] */,
                                                                        long long timeout, bool mute = false) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, G = (int)gridDim.x;
  ++epoch;
  const unsigned long long tag = launch_tag | epoch;
  const int set = (int)(epoch & 1u) * MAX_GRID;
  if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NR; ++i) { v[i] = wave_sum(v[i]); if (lane == 0) s_red[wv * 2 + i] = v[i]; }
  __syncthreads(); // every wave's exchange stores are acknowledged, its partials are in LDS
  if (wv == 0) {
    double a[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      double t = 0;
#pragma unroll
      for (int q = 0; q < WPB; ++q) t += s_red[q * 2 + i];
      if (lane == i && !mute) st_sum(r_sum, ((set + (int)blockIdx.x) * 2 + i) * 16, t, tag);
      a[i] = 0;
    }
    const int nper = (G + W - 1) / W;
    bool ok = true;
    for (int c0 = 0; c0 < nper; c0 += 4) {
      u32x4 rec[4][NR];
      unsigned pend = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) if (c0 + u < nper && lane + W * (c0 + u) < G) pend |= ((1u << NR) - 1u) << (NR * u);
      const long long t0 = wall_clock64();
      while (true) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < NR; ++i)
            if (pend & (1u << (NR * u + i))) rec[u][i] = __builtin_amdgcn_raw_buffer_load_b128(r_sum, ((set + lane + W * (c0 + u)) * 2 + i) * 16, 0, SC1);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < NR; ++i)
            if ((pend & (1u << (NR * u + i))) && ((((unsigned long long)rec[u][i].w << 32) | rec[u][i].z) ^ (((unsigned long long)rec[u][i].y << 32) | rec[u][i].x)) == tag) pend &= ~(1u << (NR * u + i));
        if (!__any(pend != 0)) break;
        if (wall_clock64() - t0 > timeout) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) break;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NR; ++i)
          if (c0 + u < nper && lane + W * (c0 + u) < G) a[i] += __builtin_bit_cast(double, ((unsigned long long)rec[u][i].y << 32) | rec[u][i].x);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) { a[i] = wave_sum(a[i]); if (lane == 0) s_red[WPB * 2 + i] = a[i]; }
    if (lane == 0) {
      s_red[WPB * 2 + 2] = ok ? 1.0 : 0.0;
      if (!ok) __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NR; ++i) v[i] = s_red[WPB * 2 + i];
  const bool ok = s_red[WPB * 2 + 2] != 0.0;
  __syncthreads(); // (s_red is rewritten by the next rendezvous)
  return ok;
}

// unpivoted Gauss-Jordan of a D x D block held in registers (damped symmetric positive definite blocks)
template <int D> __device__ __forceinline__ void invert(double (&A)[D * D], double (&R)[D * D]) {
#pragma unroll
  for (int i = 0; i < D * D; ++i) R[i] = (i / D == i % D) ? 1.0 : 0.0;
#pragma unroll
  for (int p = 0; p < D; ++p) {
    const double ip = 1.0 / A[p * D + p];
#pragma unroll
    for (int c = 0; c < D; ++c) { A[p * D + c] *= ip; R[p * D + c] *= ip; }
#pragma unroll
    for (int r = 0; r < D; ++r) {
      if (r == p) continue;
      const double f = A[r * D + p];
#pragma unroll
      for (int c = 0; c < D; ++c) { A[r * D + c] -= f * A[p * D + c]; R[r * D + c] -= f * R[p * D + c]; }
    }
  }
}

template <typename T> struct SolveArgs {
  int NV, NVp;                 // active vertices, padded
  int lg, nslices;             // 1 << lg lanes per vertex; slices of 64 >> lg vertices
  const int *sbase;            // [nslices + 1] first entry group of a slice
  const int *enbr;             // [groups][64] neighbour vertex of an entry (-1: none / padding / fixed)
  const int *k2l;              // engine vertex -> descriptor-local vertex
  const int *efac;             // [groups][64] active factor of an entry * 2 + its side (-1: padding)
  const T *frec;               // [2][na][2 (2 D D + D)] the factor kernel's records, per side J_v^T W J_v | J_v^T W r | J_v^T W J_other; set ctl->cur is the accepted point's
  size_t na;
  T *B;                        // per entry group, element-major [(group * D D + e) * 64 + lane]: J_v^T W J_nbr, formed by the assemble phase
  size_t zero_block;           // offset in B of D D x 64 zeros behind the last group (what padding entries multiply)
  T *Hs, *Minv, *s, *b, *dg;   // per vertex, element-major [e * NVp + k]
  T *x, *xb, *r, *t, *p, *y;
  T *ex;                       // [NVp][2 D] s.z' | s.p
  void *sums;                  // [2][MAX_GRID][2] flagged 16-byte records {partial, tag} (grid_sums)
  double *part_rho;            // [MAX_GRID] rho-denominator partials for k_pe_factor
  int *fail;
  Ctl *ctl;
  int max_iter, identity_precond, use_identity, scale_system;
  double tol, rej;
  int var;                     // GRAPHITE_POSE_VAR 256 (tests): workgroup 0 never announces itself, every rendezvous times out
  long long timeout;           // rendezvous time-out in 10 ns ticks (2 s; 20 ms with var 256)
  long long *dbg;              // GRAPHITE_POSE_DEBUG: wall-clock stamps of workgroup 0 at its phase boundaries, [64]; nullptr off
  T *graph_b, *graph_scales, *dx; // Graph::get_b / get_jacobian_scales / the step, column order (what the generic loop leaves behind)
};

// ---- the solve ------------------------------------------------------------------------------------------------------------------
// LPV = 1 << lg lanes share a vertex: each gathers every LPV-th entry of it (their partial sums meet through lane shuffles), all of them
// carry the vertex's own small products redundantly, sub-lane 0 stores.  A wave's SLICE is its 64 >> lg vertices.
// ONE_PASS (the grid has a wave for every slice — graphs up to 64 k lanes of vertices x LPV — and D <= 4): the vertex's own state
// (x, its backup, r, z', p, y, scales, b, clamped diagonal, the scaled Hessian block, the block-Jacobi inverse) stays in REGISTERS
// for the whole launch: the update phase touches no memory but the record it publishes (2.0 -> 0.5 us per PCG iteration on 10 k
// poses), the operator phase reads only neighbour records and entry blocks.  Otherwise every phase reloads it (grid-stride over slices).
constexpr int LDS_GROUPS = 16;  // entry groups of a wave's slice whose neighbour ids stay in LDS for the whole launch (4 KB per wave)
template <typename T, typename VTr, int D, bool ONE_PASS>
__global__ void __launch_bounds__(W * WPB) k_pe_solve(const SolveArgs<T> A, typename VTr::Vertex *verts /* the HBM mirror, by local vertex id */, typename state_of<VTr>::type *backup) {
  Ctl *const ctl = A.ctl;
  if (ctl->stop) return;
  constexpr int DD = D * D;
  constexpr int CH = DD <= 9 ? 4 : DD <= 16 ? 2 : 1; // entry groups gathered at once: their neighbour records and blocks are in flight together
  __shared__ int s_nbr[WPB][LDS_GROUPS * W];
  __shared__ double s_red[WPB * 2 + 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gwave = (int)blockIdx.x * WPB + wv, nwaves = (int)gridDim.x * WPB;
  const int lg = A.lg, LPV = 1 << lg, VPW = W >> lg, sub = lane & (LPV - 1), vl = lane >> lg;
  const int NV = A.NV, NVp = A.NVp, nslices = A.nslices;
  const bool first_thread = blockIdx.x == 0 && threadIdx.x == 0;
  const bool fresh = ctl->fresh != 0;
  const double mu = ctl->mu;
  const unsigned long long launch_tag = (unsigned long long)(unsigned)(ctl->it + 1) << 32;
  unsigned epoch = 0;
  int n_stamp = 0;
  auto stamp = [&]() __attribute__((always_inline)) { if (A.dbg && first_thread && n_stamp < 63) A.dbg[n_stamp++] = wall_clock64(); };
  stamp();
  long long op_t0 = 0; // (debug: every workgroup's time in the operator phase of the launch's LAST PCG iteration, dbg[64 + blockIdx])
  const __amdgpu_buffer_rsrc_t r_ex = rsrc(A.ex, (size_t)NVp * 2 * D * sizeof(T));
  const __amdgpu_buffer_rsrc_t r_sum = rsrc(A.sums, (size_t)2 * MAX_GRID * 2 * 16);
  constexpr int REC = 2 * D * (int)sizeof(T);
  // the LPV partial sums of a vertex meet in every one of its lanes, fixed order; the shuffles of one step are issued together
  auto over_lanes = [&](auto &arr) __attribute__((always_inline)) {
    constexpr int NE = (int)(sizeof(arr) / sizeof(arr[0]));
    for (int m = 1; m < LPV; m <<= 1) {
      T o[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) o[e] = __shfl_xor(arr[e], m, 64);
#pragma unroll
      for (int e = 0; e < NE; ++e) arr[e] += o[e];
    }
  };
  // one slice per wave (the grid covers every vertex in one pass) and a short one: its neighbour ids are read once
  bool nbr_in_lds = false;
  if (nwaves >= nslices && gwave < nslices) {
    const int g0 = A.sbase[gwave], ng = A.sbase[gwave + 1] - g0;
    if (ng <= LDS_GROUPS) {
      nbr_in_lds = true;
      for (int j = 0; j < ng; ++j) s_nbr[wv][j * W + lane] = A.enbr[(size_t)(g0 + j) * W + lane];
    }
  }
  // the vertex's own state: registers for the whole launch (ONE_PASS) or reloaded by every phase
  T ox[D], oxb[D], orr[D], ot[D], op[D], oy[D], os[D], ob[D], odg[D], oH[DD], oMi[DD];
  const int last = ONE_PASS ? gwave + 1 : nslices, step = ONE_PASS ? 1 : nwaves; // (ONE_PASS: a wave's loop over slices is its own slice, once)
#define PE_AT(arr, e, k) A.arr[(size_t)(e) * NVp + (k)]

  // ---- assemble (new linearisation), damp, invert, start --------------------------------------------------------------------------
  double acc[2] = {0, 0};
  for (int ws = gwave; ws < last && ws < nslices; ws += step) {
    const int k = ws * VPW + vl, g0 = A.sbase[ws], g1 = A.sbase[ws + 1];
    const bool valid = k < NV, owner = valid && sub == 0;
    const int kk = valid ? k : 0;
    if (fresh) {
      T H[DD], g[D];
#pragma unroll
      for (int e = 0; e < DD; ++e) H[e] = T(0);
#pragma unroll
      for (int e = 0; e < D; ++e) g[e] = T(0);
      // Every entry gathers its side of the factor's record (contiguous: J_v^T W J_v | J_v^T W r | J_v^T W J_nbr) and keeps the
      // off-diagonal block for the PCG iterations, lane-consecutive.  Padding entries read factor 0 with weight zero: no branch
      // between the loads.
      constexpr int RS = 2 * DD + D, RF = 2 * RS;
      const T *const frec_cur = A.frec + (size_t)ctl->cur * A.na * RF;
#pragma unroll 2
      for (int grp = g0; grp < g1; ++grp) {
        const int code = A.efac[(size_t)grp * W + lane];
        const T on = code >= 0 ? T(1) : T(0);
        const int cc = code >= 0 ? code : 0;
        const T *rec = frec_cur + (size_t)(cc >> 1) * RF + (cc & 1) * RS;
        T v[RS];
#pragma unroll
        for (int q = 0; q < RS; ++q) v[q] = rec[q];
#pragma unroll
        for (int e = 0; e < DD; ++e) H[e] += on * v[e];
#pragma unroll
        for (int e = 0; e < D; ++e) g[e] += on * v[DD + e];
        if (code >= 0) {
#pragma unroll
          for (int e = 0; e < DD; ++e) A.B[((size_t)grp * DD + e) * W + lane] = v[DD + D + e];
        }
      }
      over_lanes(H);
      over_lanes(g);
#pragma unroll
      for (int c = 0; c < D; ++c) {
        // graph.hpp:253-262: scales 1 / (eps + sqrt(diag)) in double
        os[c] = A.scale_system ? (T)(1.0 / (std::numeric_limits<double>::epsilon() + sqrt((double)H[c * D + c]))) : T(1);
        ob[c] = -(os[c] * g[c]);
      }
#pragma unroll
      for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) oH[r * D + c] = os[r] * H[r * D + c] * os[c];
#pragma unroll
      for (int c = 0; c < D; ++c) {
        const double d = (double)oH[c * D + c];
        odg[c] = (T)(d < 1.0e-6 ? 1.0e-6 : (d > 1.0e32 ? 1.0e32 : d)); // pcg.hpp:93-103
      }
      if (owner) { // (kept in memory too: a rejected step's next launch and the caller's Graph::get_b / get_jacobian_scales read them)
#pragma unroll
        for (int c = 0; c < D; ++c) {
          PE_AT(s, c, k) = os[c]; PE_AT(b, c, k) = ob[c]; PE_AT(dg, c, k) = odg[c];
          A.graph_scales[(size_t)k * D + c] = os[c]; A.graph_b[(size_t)k * D + c] = ob[c];
        }
#pragma unroll
        for (int e = 0; e < DD; ++e) PE_AT(Hs, e, k) = oH[e];
      }
    } else {
#pragma unroll
      for (int c = 0; c < D; ++c) { os[c] = PE_AT(s, c, kk); ob[c] = PE_AT(b, c, kk); odg[c] = PE_AT(dg, c, kk); }
#pragma unroll
      for (int e = 0; e < DD; ++e) oH[e] = PE_AT(Hs, e, kk);
    }
    // block-Jacobi: diagonal <- d + mu clamp(d, 1e-6, 1e32) (or d + mu), ops/hessian.hpp:80-112; inverse; identity.hpp: z = r
    if (!A.identity_precond) {
      double M[DD], R[DD];
#pragma unroll
      for (int e = 0; e < DD; ++e) M[e] = (double)oH[e];
#pragma unroll
      for (int c = 0; c < D; ++c) {
        const double d = M[c * D + c], cl = d < 1.0e-6 ? 1.0e-6 : (d > 1.0e32 ? 1.0e32 : d);
        M[c * D + c] = A.use_identity ? d + mu : d + mu * cl;
      }
      invert<D>(M, R);
#pragma unroll
      for (int e = 0; e < DD; ++e) oMi[e] = (T)R[e];
    }
    T rec[2 * D];
#pragma unroll
    for (int q = 0; q < D; ++q) {
      orr[q] = ob[q];
      T z = ob[q];
      if (!A.identity_precond) {
        z = T(0);
#pragma unroll
        for (int c = 0; c < D; ++c) z += oMi[q * D + c] * ob[c];
      }
      ot[q] = z; ox[q] = T(0); oxb[q] = T(0); op[q] = T(0);
      rec[q] = os[q] * z; rec[D + q] = T(0);
    }
    if (owner) {
#pragma unroll
      for (int c = 0; c < D; ++c) { acc[0] += (double)orr[c] * (double)orr[c]; acc[1] += (double)orr[c] * (double)ot[c]; }
      st_rec<T, 2 * D>(r_ex, k * REC, rec);
      if (!ONE_PASS) {
#pragma unroll
        for (int c = 0; c < D; ++c) { PE_AT(r, c, k) = orr[c]; PE_AT(t, c, k) = ot[c]; PE_AT(x, c, k) = T(0); PE_AT(xb, c, k) = T(0); PE_AT(p, c, k) = T(0); }
        if (!A.identity_precond) {
#pragma unroll
          for (int e = 0; e < DD; ++e) PE_AT(Minv, e, k) = oMi[e];
        }
      }
    }
  }
  stamp();
  if (!grid_sums<2, true>(acc, r_sum, launch_tag, epoch, A.fail, s_red, A.timeout, (A.var & 256) && blockIdx.x == 0)) { if (threadIdx.x == 0) ctl->stop = 16; return; }
  stamp();
  // pcg.hpp:114-131: z = Minv (r / |r|), p = z, rz = r . z
  double sigma = 1.0 / sqrt(acc[0]), rz = acc[1] * sigma, beta = 0, rz0 = INFINITY;
  int its = 0, reject = 0;
  for (int it = 0; it < A.max_iter; ++it) {
    if (rz == 0) break; // pcg.hpp:133
    // ---- operator: p = sigma z' + beta p;  y = (S J^T rho' P J S + mu D) p;  p . y ------------------------------------------------
    double den[1] = {0};
    if (A.dbg && threadIdx.x == 0) op_t0 = wall_clock64();
    for (int ws = gwave; ws < last && ws < nslices; ws += step) {
      const int k = ws * VPW + vl, g0 = A.sbase[ws], ng = A.sbase[ws + 1] - g0;
      const bool valid = k < NV, owner = valid && sub == 0;
      const int kk = valid ? k : 0;
      // everything the phase reads is requested up front (no branch between the loads): the vertex's own state and blocks (not
      // ONE_PASS), then per chunk the neighbours' records (the only sc1 round trip) and the entry blocks; padding entries read the
      // vertex's own record against a zero block
      T a[D];
      if (!ONE_PASS) {
#pragma unroll
        for (int c = 0; c < D; ++c) { ot[c] = PE_AT(t, c, kk); op[c] = PE_AT(p, c, kk); os[c] = PE_AT(s, c, kk); odg[c] = PE_AT(dg, c, kk); }
#pragma unroll
        for (int e = 0; e < DD; ++e) oH[e] = PE_AT(Hs, e, kk);
      }
#pragma unroll
      for (int c = 0; c < D; ++c) { op[c] = (T)sigma * ot[c] + (T)beta * op[c]; a[c] = T(0); }
      // (straight-line per chunk — no branch between the loads of a chunk, or the compiler waits for each group's loads in turn)
      auto gather = [&](auto in_lds) __attribute__((always_inline)) {
        for (int j0 = 0; j0 < ng; j0 += CH) {
          int nb[CH];
          T rec[CH][2 * D], Bv[CH][DD];
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const int j = j0 + u < ng ? j0 + u : ng - 1; // (a short last chunk reads its last group again, weight zero below)
            int n;
            if constexpr (decltype(in_lds)::value) n = s_nbr[wv][j * W + lane];
            else n = A.enbr[(size_t)(g0 + j) * W + lane];
            nb[u] = j0 + u < ng ? n : -2;
          }
#pragma unroll
          for (int u = 0; u < CH; ++u) ld_rec<T, 2 * D>(r_ex, (nb[u] >= 0 ? nb[u] : kk) * REC, rec[u]);
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            // (a padding entry reads the ONE zero block behind the last group — the same line for every such lane — instead of its own
            // slot: on large graphs, where this phase streams the blocks from HBM, a third of the slots are padding)
            const int j = j0 + u < ng ? j0 + u : ng - 1;
            const size_t base = nb[u] >= 0 ? (size_t)(g0 + j) * DD * W + lane : A.zero_block;
#pragma unroll
            for (int e = 0; e < DD; ++e) Bv[u][e] = A.B[base + (size_t)e * W];
          }
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const T on = nb[u] >= 0 ? T(1) : T(0);
            T pn[D];
#pragma unroll
            for (int c = 0; c < D; ++c) pn[c] = on * ((T)sigma * rec[u][c] + (T)beta * rec[u][D + c]);
#pragma unroll
            for (int q = 0; q < D; ++q)
#pragma unroll
              for (int c = 0; c < D; ++c) a[q] += Bv[u][q * D + c] * pn[c];
          }
        }
      };
      if (nbr_in_lds) gather(std::true_type{}); else gather(std::false_type{});
      over_lanes(a);
#pragma unroll
      for (int q = 0; q < D; ++q) {
        T yq = os[q] * a[q];
#pragma unroll
        for (int c = 0; c < D; ++c) yq += oH[q * D + c] * op[c];
        yq += (T)mu * (A.use_identity ? T(1) : odg[q]) * op[q]; // ops/vector.hpp:25-41
        oy[q] = yq;
        if (owner) den[0] += (double)op[q] * (double)yq;
      }
      if (!ONE_PASS && owner) {
#pragma unroll
        for (int q = 0; q < D; ++q) { PE_AT(y, q, k) = oy[q]; PE_AT(p, q, k) = op[q]; }
      }
    }
    if (A.dbg && threadIdx.x == 0) A.dbg[64 + blockIdx.x] = wall_clock64() - op_t0;
    stamp();
    if (!grid_sums<1, false>(den, r_sum, launch_tag, epoch, A.fail, s_red, A.timeout)) { if (threadIdx.x == 0) ctl->stop = 16; return; }
    stamp();
    if (den[0] == 0 || den[0] != den[0]) break; // pcg_schur.hpp:122 / the decision kernel of the generic loop
    const T alpha = (T)(rz / den[0]);
    // ---- update: x += alpha p (backup first), r -= alpha y, z' = Minv r, the two dots, publish [s.z' | s.p] -----------------------
    double dots[2] = {0, 0};
    for (int ws = gwave; ws < last && ws < nslices; ws += step) {
      const int k = ws * VPW + vl;
      const bool valid = k < NV, owner = valid && sub == 0;
      if (!ONE_PASS) {
        if (!owner) continue;
#pragma unroll
        for (int c = 0; c < D; ++c) { ox[c] = PE_AT(x, c, k); op[c] = PE_AT(p, c, k); oy[c] = PE_AT(y, c, k); orr[c] = PE_AT(r, c, k); os[c] = PE_AT(s, c, k); }
        if (!A.identity_precond) {
#pragma unroll
          for (int e = 0; e < DD; ++e) oMi[e] = PE_AT(Minv, e, k);
        }
      }
      T rec[2 * D];
#pragma unroll
      for (int c = 0; c < D; ++c) {
        oxb[c] = ox[c];
        ox[c] = alpha * op[c] + ox[c];
        orr[c] = -alpha * oy[c] + orr[c];
      }
#pragma unroll
      for (int q = 0; q < D; ++q) {
        T z = orr[q];
        if (!A.identity_precond) {
          z = T(0);
#pragma unroll
          for (int c = 0; c < D; ++c) z += oMi[q * D + c] * orr[c];
        }
        ot[q] = z;
        rec[q] = os[q] * z; rec[D + q] = os[q] * op[q];
      }
      if (owner) {
#pragma unroll
        for (int q = 0; q < D; ++q) { dots[0] += (double)orr[q] * (double)orr[q]; dots[1] += (double)orr[q] * (double)ot[q]; }
        st_rec<T, 2 * D>(r_ex, k * REC, rec);
        if (!ONE_PASS) {
#pragma unroll
          for (int c = 0; c < D; ++c) { PE_AT(xb, c, k) = oxb[c]; PE_AT(x, c, k) = ox[c]; PE_AT(r, c, k) = orr[c]; PE_AT(t, c, k) = ot[c]; }
        }
      }
    }
    stamp();
    if (!grid_sums<2, true>(dots, r_sum, launch_tag, epoch, A.fail, s_red, A.timeout)) { if (threadIdx.x == 0) ctl->stop = 16; return; }
    stamp();
    const double rinv = 1.0 / sqrt(dots[0]), rz_new = dots[1] * rinv, arz = rz_new < 0 ? -rz_new : rz_new;
    its = it + 1;
    if (arz > A.rej * rz0 || rz_new != rz_new) { reject = 1; break; } // pcg.hpp:203-207
    rz0 = rz0 < arz ? rz0 : arz;
    beta = rz_new / rz;
    rz = rz_new;
    sigma = rinv;
    if (arz < A.tol) break;
  }
  // ---- trial step: Graph::backup_parameters / apply_update (graph.hpp:292-309), rho-denominator partials (levenberg_marquardt.hpp:20-47)
  double rden = 0;
  for (int ws = gwave; ws < last && ws < nslices; ws += step) {
    const int k = ws * VPW + vl;
    if (!(k < NV && sub == 0)) continue;
    if (!ONE_PASS) {
#pragma unroll
      for (int c = 0; c < D; ++c) { ox[c] = PE_AT(x, c, k); oxb[c] = PE_AT(xb, c, k); ob[c] = PE_AT(b, c, k); os[c] = PE_AT(s, c, k); }
    }
    T d[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const T xc = reject ? oxb[c] : ox[c];
      A.dx[(size_t)k * D + c] = xc;
      rden += (double)xc * ((double)(T)mu * (double)xc + (double)ob[c]);
      d[c] = xc * os[c]; // ops/update.hpp:26
    }
    const int l = A.k2l[k];
    if (fresh) {
      if constexpr (state_of<VTr>::custom) backup[l] = VTr::get_state(verts[l]);
      else backup[l] = verts[l];
    } else {
      if constexpr (state_of<VTr>::custom) VTr::set_state(verts[l], backup[l]);
      else verts[l] = backup[l];
    }
    VTr::update(verts[l], d);
  }
#undef PE_AT
  rden = wave_sum(rden);
  if (lane == 0) A.part_rho[gwave] = rden;
  if (first_thread) { ctl->last_pcg_its = its; ctl->pcg_its += its; }
  stamp();
  if (A.dbg && first_thread) A.dbg[63] = n_stamp;
}

template <typename VTr> __global__ void k_pe_finish(const Ctl *ctl, const int *k2l, int NV, typename VTr::Vertex *verts, const typename state_of<VTr>::type *backup) {
  if (ctl->fresh) return; // the last trial was accepted (or there was none): the vertices are where they belong
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= NV) return;
  const int l = k2l[k];
  if constexpr (state_of<VTr>::custom) VTr::set_state(verts[l], backup[l]);
  else verts[l] = backup[l];
}

template <typename T> struct FactorArgs {
  Ctl *ctl;
  double *part;             // chi2 partials, one per workgroup
  const double *part_rho;   // the solve's rho-denominator partials
  int n_rho;
  size_t na;                // active factors of THIS descriptor
  size_t a0, na_all;        // ... which are factors [a0, a0 + na) of the na_all active factors of the graph's descriptors (records, pos, lij)
  int block0, total_blocks; // this launch's first chi2 partial; workgroups of all the descriptors' launches (the last of them decides)
  double *tr_chi2, *tr_mu;  // [iterations + 1]
  long long *tr_clock;      // [iterations + 1]
  const int *pos;           // [na_all][2] entry slot of (factor, side): group * 64 + lane, -1: that vertex has no column
  const int *lij;           // [na_all][2] descriptor-local vertex ids of the active factor, -1 for the missing slot of a unary factor (the vertex objects are read from the mirror
                            // directly: active list -> id table -> pointer table -> vertex is two dependent loads longer)
  T *frec;                  // [2][na_all][2 (2 D D + D)] per side J_v^T W J_v | J_v^T W r | J_v^T W J_other: the accepted point's set (ctl->cur) and the trial point's
  int early;
};

// threads per workgroup of the factor kernel (measured on 10 k poses, us per LM iteration: 64 threads 142.6, 128 132.8, 256 124.9 — fewer
// partials and tickets, longer staged runs — although 256 threads leave a quarter of the CUs without a workgroup there)
#ifndef GRAPHITE_POSE_FACTOR_THREADS
#define GRAPHITE_POSE_FACTOR_THREADS 256
#endif
constexpr int FTPB = GRAPHITE_POSE_FACTOR_THREADS;

// Jacobian block of slot I at the vertices v: the user's jacobian<> or one dual-number evaluation per column (ops/linearize.hpp:43-138)
template <typename F, size_t I, typename VT, size_t... Is>
__device__ inline void pe_jacobian(const FactorView<F> &fv, size_t f, const VT &v, typename F::Scalar *J, std::index_sequence<Is...> seq) {
  using T = typename F::Scalar;
  constexpr size_t d = slot_dim<F, I>(), E = F::E;
  if constexpr (std::is_same<typename F::Traits::Differentiation, DifferentiationMode::Manual>::value) {
    for (size_t k = 0; k < E * d; ++k) J[k] = T(0); // ops/linearize.hpp:127
    call_jacobian_t<F, I, T>(v, fv.obs[f], fv.data[f], J, seq);
  } else {
    using Dl = Dual<T, T>;
    for (size_t col = 0; col < d; ++col) { // ops/linearize.hpp:43-79: one seeded column per evaluation
      std::tuple<Dl[slot_dim<F, Is>()]...> p;
      ((slot_traits<F, Is>::parameters(*std::get<Is>(v), (Dl *)std::get<Is>(p))), ...);
      std::get<I>(p)[col].dual = T(1);
      Dl err[E];
      call_error<F, Dl>(v, p, fv.obs[f], fv.data[f], err, seq);
      for (size_t i = 0; i < E; ++i) J[col * E + i] = err[i].dual;
    }
  }
}
// ---- the factor kernel: error, chi2, rho' AND the Jacobian record at the trial point; LM decision by the last workgroup ------------
// (mode 0: the starting point, no decision).  The record of the trial point goes to the OTHER of two record sets: an accepted step
// flips ctl->cur and the next solve assembles from it, a rejected one leaves the accepted point's set untouched — the Jacobians of a
// rejected trial are wasted work (rare), a launch per LM iteration is saved.

template <typename F, size_t... Is>
__global__ void __launch_bounds__(FTPB) k_pe_factor(FactorView<F> fv, const FactorArgs<typename F::Scalar> A, slot_vertex<F, 0> *mirror, int mode, std::index_sequence<Is...> seq) {
  using T = typename F::Scalar;
  constexpr size_t E = F::E;
  constexpr int D = (int)slot_dim<F, 0>(), Ei = (int)F::E, RS = 2 * D * D + D, RF = 2 * RS;
  constexpr bool STAGE = (size_t)FTPB * RF * sizeof(T) <= 96 * 1024; // (larger records are written by their own thread, strided)
  __shared__ double red[FTPB];
  __shared__ int s_last;
  extern __shared__ __align__(16) unsigned char pe_stage_raw[];
  T *const stage = reinterpret_cast<T *>(pe_stage_raw); // [FTPB][RF]: the records of this workgroup's factors
  Ctl *const ctl = A.ctl;
  if (ctl->stop) return;
  const int buf = mode == 0 ? ctl->cur : ctl->cur ^ 1;
  bool have = false;
  const size_t a = blockIdx.x * (size_t)FTPB + threadIdx.x, ga = A.a0 + a; // in this descriptor's active list / among all the graph's active factors
  double c2 = 0;
  if (a < A.na) {
    const size_t f = fv.active_ids[a];
    auto v = std::make_tuple((mirror + A.lij[2 * ga + Is])...);
    std::tuple<T[slot_dim<F, Is>()]...> p;
    ((slot_traits<F, Is>::parameters(*std::get<Is>(v), (T *)std::get<Is>(p))), ...);
    T err[E];
    call_error<F, T>(v, p, fv.obs[f], fv.data[f], err, seq);
    T Pm[E * E];
#pragma unroll
    for (size_t i = 0; i < E * E; ++i) Pm[i] = (T)fv.pmat[f * E * E + i];
    T value = 0; // ops/chi2.hpp:10-44, precision read row-major
#pragma unroll
    for (size_t i = 0; i < E; ++i) {
      T r2 = 0;
#pragma unroll
      for (size_t j = 0; j < E; ++j) r2 += Pm[i * E + j] * err[j];
      value += r2 * err[i];
    }
    c2 = (double)fv.loss[f].loss(value);
    const T w = (T)fv.loss[f].loss_derivative(value);
    // the record: for each side v of the factor  J_v^T W J_v (D x D) | J_v^T W r (D) | J_v^T W J_other (D x D)  with W = rho' P
    // (column-major E x D Jacobian blocks, ops/error.hpp:146-149); the block of a vertex without a column is zero (ops/linearize.hpp:24)
    const int p0 = A.pos[2 * ga], p1 = A.pos[2 * ga + 1]; // (a unary factor has no second slot: p1 = -1, its second side stays zero)
    if (p0 >= 0 || p1 >= 0) {
      T J0[Ei * D], J1[Ei * D], A0[Ei * D], A1[Ei * D], Wr[Ei];
      if (p0 >= 0) pe_jacobian<F, 0>(fv, f, v, J0, seq); else for (int k = 0; k < Ei * D; ++k) J0[k] = T(0);
      if constexpr (F::N == 2) { if (p1 >= 0) pe_jacobian<F, 1>(fv, f, v, J1, seq); else for (int k = 0; k < Ei * D; ++k) J1[k] = T(0); }
      else for (int k = 0; k < Ei * D; ++k) J1[k] = T(0);
#pragma unroll
      for (int i = 0; i < Ei; ++i) {
        T sr = T(0);
#pragma unroll
        for (int j = 0; j < Ei; ++j) sr += w * Pm[i * Ei + j] * err[j];
        Wr[i] = sr;
#pragma unroll
        for (int c = 0; c < D; ++c) {
          T s0 = T(0), s1 = T(0);
#pragma unroll
          for (int j = 0; j < Ei; ++j) { s0 += w * Pm[i * Ei + j] * J0[c * Ei + j]; s1 += w * Pm[i * Ei + j] * J1[c * Ei + j]; }
          A0[i * D + c] = s0; A1[i * D + c] = s1;
        }
      }
      auto side = [&](T *out, const T *Jm, const T *Am, const T *Ao) {
#pragma unroll
        for (int r = 0; r < D; ++r) {
          T g = T(0);
#pragma unroll
          for (int i = 0; i < Ei; ++i) g += Jm[r * Ei + i] * Wr[i];
          out[D * D + r] = g;
#pragma unroll
          for (int c = 0; c < D; ++c) {
            T h = T(0), bb = T(0);
#pragma unroll
            for (int i = 0; i < Ei; ++i) { h += Jm[r * Ei + i] * Am[i * D + c]; bb += Jm[r * Ei + i] * Ao[i * D + c]; }
            out[r * D + c] = h; out[D * D + D + r * D + c] = bb;
          }
        }
      };
      T *const out = STAGE ? stage + (size_t)threadIdx.x * RF : A.frec + ((size_t)buf * A.na_all + ga) * RF;
      side(out, J0, A0, A1);
      side(out + RS, J1, A1, A0);
      have = true;
    }
  }
  if (STAGE) {
  if (!have) for (int q = 0; q < RF; ++q) stage[(size_t)threadIdx.x * RF + q] = T(0);
  __syncthreads();
  { // the workgroup's records leave lane-consecutive (each thread's own 336 bytes at a 336-byte stride were 42 partial-line stores per wave instruction)
    const size_t first = blockIdx.x * (size_t)FTPB, count = first < A.na ? (A.na - first < (size_t)FTPB ? A.na - first : (size_t)FTPB) : 0;
    T *dst = A.frec + ((size_t)buf * A.na_all + A.a0 + first) * RF;
    for (size_t i = threadIdx.x; i < count * RF; i += FTPB) dst[i] = stage[i];
  }
  }
  red[threadIdx.x] = c2;
  __syncthreads();
  for (int o = FTPB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    st_x(A.part + A.block0 + blockIdx.x, red[0]);
    __threadfence();
    const unsigned tk = __hip_atomic_fetch_add(&ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = tk == (unsigned)A.total_blocks - 1u; // (the ticket runs over the launches of all the graph's factor descriptors)
  }
  __syncthreads();
  if (!s_last) return;
  double tot = 0, rd = 0;
  for (int g = threadIdx.x; g < A.total_blocks; g += FTPB) tot += ld_x(A.part + g);
  __syncthreads();
  red[threadIdx.x] = tot;
  __syncthreads();
  for (int o = FTPB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  tot = red[0];
  __syncthreads();
  if (mode) {
    for (int g = threadIdx.x; g < A.n_rho; g += FTPB) rd += A.part_rho[g];
    red[threadIdx.x] = rd;
    __syncthreads();
    for (int o = FTPB / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    rd = red[0];
  }
  if (threadIdx.x != 0) return;
  ctl->ticket = 0;
  if (mode == 0) {
    ctl->chi2 = tot;
    A.tr_chi2[0] = tot; A.tr_mu[0] = ctl->mu; A.tr_clock[0] = wall_clock64();
    return;
  }
  // levenberg_marquardt.hpp:20-47, :150-214
  const T chi2 = (T)ctl->chi2, new_chi2 = (T)tot;
  T mu = (T)ctl->mu, nu = (T)ctl->nu;
  const T denom = (T)rd + T(1.0e-3);
  const T rho = (chi2 - new_chi2) / denom;
  T shown = chi2;
  bool accepted = false;
  if (isfinite(new_chi2) && rho > T(0)) {
    accepted = true;
    double al = 1.0 - pow(2.0 * (double)rho - 1.0, 3.0);
    al = al < 2.0 / 3.0 ? al : 2.0 / 3.0;
    al = al > 1.0 / 3.0 ? al : 1.0 / 3.0;
    mu *= (T)al;
    nu = T(2);
    ctl->chi2 = (double)new_chi2;
    ctl->cur ^= 1;
    ctl->fresh = 1;
    ctl->accepted += 1;
    shown = new_chi2;
  } else {
    mu *= nu;
    nu *= T(2);
    ctl->fresh = 0;
  }
  ctl->mu = (double)mu; ctl->nu = (double)nu; ctl->rho = (double)rho;
  const int it = ctl->it;
  A.tr_chi2[it + 1] = (double)shown; A.tr_mu[it + 1] = (double)mu; A.tr_clock[it + 1] = wall_clock64();
  ctl->it = it + 1;
  int stop = 0;
  if (!isfinite(mu)) stop |= 1;
  if (rho == T(0)) stop |= 2;
  if (A.early && accepted) { // :404-414
    if (((chi2 - new_chi2) * T(1.0e3)) < chi2) ctl->num_bad += 1; else ctl->num_bad = 0;
    if (ctl->num_bad >= 3) stop |= 4;
  }
  ctl->stop = stop;
}

template <typename F, size_t... Is> inline const void *factor_kernel_ptr(std::index_sequence<Is...>) { return (const void *)&k_pe_factor<F, Is...>; }

// buffers of one vertex descriptor's engine, kept between optimiser calls (capacity is reused)
template <typename T> struct Buffers {
  hbm_vector<int> sbase, enbr, efac, k2l, pos, lij;
  hbm_vector<T> frec, B, vert, vec, ex, dx;
  hbm_vector<double> sums, part_rho, part_chi2, tr;
  hbm_vector<long long> clock, dbg;
  hbm_vector<Ctl> ctl;
  hbm_vector<int> fail;
  hbm_vector<unsigned char> start; // the vertices as the call found them (restored when a launch fails)
  // structure cache: the lists below depend on the descriptors' structure (epochs: add / remove / set_active / set_fixed ...), on the
  // optimisation level (through the active list) and on the vertex states and Hessian columns of this initialisation; a call that
  // finds all of them unchanged (the SLAM loop that optimises the same graph again) re-uses lists, uploads and the symmetry check
  bool have_structure = false;
  std::vector<size_t> key_f;  // per factor descriptor: address, structure epoch, active count, digest of the active list
  size_t key_epoch_v = 0, key_nvl = 0, key_dim = 0;
  uint64_t key_digest = 0;
  int c_NV = 0, c_NVp = 0, c_lg = 0, c_nslices = 0;
  size_t c_ngroups = 0;
  const void *B_cleared = nullptr;   // the entry-block buffer whose padding slots hold zeros (pointer and size at the time)
  size_t B_cleared_size = 0;
  bool prefer_cooperative = false;  // a plain launch timed out at a rendezvous once: later calls ask the runtime for co-residency
};
} // namespace pe
} // namespace detail

namespace detail {
namespace pe {
// what the driver asks of a factor descriptor (typed code behind BaseFactorDescriptor's virtuals)
struct FactorInfo {
  int slots = 0, error_dim = 0;
  size_t active = 0, epoch = 0;
  uint64_t active_digest = 0;
  const void *vertex_descriptor = nullptr; // the descriptor of EVERY slot, nullptr when they differ
  bool storage_is_scalar = false, symmetric = true;
};
struct VertexInfo {
  int dim = 0;
  size_t count = 0, vertex_bytes = 0, epoch = 0;
  void *mirror = nullptr;
  bool plain = false, mirrored = false;
};
} // namespace pe
} // namespace detail

// ---- typed pieces: the factor kernel on a factor descriptor's traits ---------------------------------------------------------------
template <typename T, typename S, typename FTraits>
bool FactorDescriptor<T, S, FTraits>::pose_engine_factor(detail::pe::FactorInfo &fi, bool check_symmetry) {
  using namespace detail;
  fi.slots = (int)N; fi.error_dim = (int)E; fi.active = active_count(); fi.epoch = this->structure_epoch;
  fi.storage_is_scalar = std::is_same<T, S>::value && std::is_floating_point<T>::value;
  fi.vertex_descriptor = vertex_descriptors[0];
  for (size_t i = 1; i < N; ++i) if (vertex_descriptors[i] != vertex_descriptors[0]) fi.vertex_descriptor = nullptr;
  fi.active_digest = digest(0x706F7365ull, active_indices.raw(), fi.active * sizeof(size_t));
  if (check_symmetry) // symmetric precision matrices: the blocks of the two sides are formed from one W = rho' P
    for (size_t a = 0; a < fi.active && fi.symmetric; ++a) {
      const S *P = precision_matrices.raw() + active_indices[a] * E * E;
      for (size_t i = 0; i < E; ++i)
        for (size_t j = i + 1; j < E; ++j)
          if (P[i * E + j] != P[j * E + i]) fi.symmetric = false;
    }
  if (!tables_mirrored) refresh_table_mirrors(false);
  else if (m_pmat.get(precision_matrices, true) == precision_matrices.raw()) m_pmat.refresh(precision_matrices); // (the light initialisation leaves the precision matrices in pinned host memory)
  return N <= 2 && E <= 7 && std::is_same<typename std::tuple_element<0, VDTuple>::type, typename std::tuple_element<N - 1, VDTuple>::type>::value;
}
template <typename T, typename S, typename FTraits>
void FactorDescriptor<T, S, FTraits>::pose_engine_ids(int *lij) {
  const size_t na = active_count();
  for (size_t a = 0; a < na; ++a) {
    const size_t f = active_indices[a];
    lij[2 * a] = (int)device_ids[f * N];
    lij[2 * a + 1] = N > 1 ? (int)device_ids[f * N + 1] : -1;
  }
}
template <typename T, typename S, typename FTraits>
void FactorDescriptor<T, S, FTraits>::pose_engine_launch(const detail::pe::FactorArgs<T> &fa, void *mirror, int mode) {
  using namespace detail;
  if constexpr (N <= 2 && std::is_same<T, S>::value && std::is_same<typename std::tuple_element<0, VDTuple>::type, typename std::tuple_element<N - 1, VDTuple>::type>::value) {
    using VD0 = typename std::tuple_element<0, VDTuple>::type;
    constexpr int D = (int)VD0::dim, DD = D * D;
    size_t stage_bytes = (size_t)pe::FTPB * (2 * (2 * DD + D)) * sizeof(T);
    if (stage_bytes > 96 * 1024) stage_bytes = 0; // (written directly, see k_pe_factor)
    constexpr auto seq = std::make_index_sequence<N>{};
    if (stage_bytes > 64 * 1024 && !pose_stage_attr_set) {
      GRAPHITE_HIP(hipFuncSetAttribute(pe::factor_kernel_ptr<FactorDescriptor>(seq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes));
      pose_stage_attr_set = true;
    }
    pe::k_pe_factor<FactorDescriptor><<<(unsigned)((fa.na + pe::FTPB - 1) / pe::FTPB), pe::FTPB, stage_bytes>>>(view(), fa, static_cast<typename VD0::VertexType *>(mirror), mode, seq);
  }
}

// ---- typed pieces: the solve on a vertex descriptor's traits ------------------------------------------------------------------------
template <typename T, typename S, typename VTraits>
bool VertexDescriptor<T, S, VTraits>::pose_engine_vertex(detail::pe::VertexInfo &vi) {
  vi.dim = (int)dim; vi.count = count(); vi.vertex_bytes = sizeof(VertexType); vi.epoch = this->structure_epoch;
  vi.plain = can_mirror; vi.mirrored = mirrored;
  if constexpr (can_mirror) vi.mirror = mirror.raw();
  return can_mirror && dim <= 7;
}
template <typename T, typename S, typename VTraits>
int VertexDescriptor<T, S, VTraits>::pose_engine_occupancy(bool one_pass) {
  using namespace detail;
  int per_cu = 0;
  if constexpr (can_mirror && dim <= 7 && std::is_floating_point<T>::value) {
    constexpr int D = (int)dim;
    constexpr bool CAN = D * D <= 16;
    if (one_pass && !CAN) return 0;
    const void *fn = one_pass ? (const void *)&pe::k_pe_solve<T, VTraits, D, CAN> : (const void *)&pe::k_pe_solve<T, VTraits, D, false>;
    GRAPHITE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, pe::W * pe::WPB, 0));
  }
  return per_cu;
}
template <typename T, typename S, typename VTraits>
void VertexDescriptor<T, S, VTraits>::pose_engine_solve(const detail::pe::SolveArgs<T> &sa, int G, bool cooperative, bool one_pass) {
  using namespace detail;
  if constexpr (can_mirror && dim <= 7 && std::is_floating_point<T>::value) {
    constexpr int D = (int)dim;
    constexpr bool CAN = D * D <= 16;
    auto *vm = mirror.raw();
    auto *bk = backup_ptr();
    if (cooperative) {
      void *args[] = {(void *)&sa, (void *)&vm, (void *)&bk};
      const void *fn = one_pass ? (const void *)&pe::k_pe_solve<T, VTraits, D, CAN> : (const void *)&pe::k_pe_solve<T, VTraits, D, false>;
      GRAPHITE_HIP(hipLaunchCooperativeKernel(fn, dim3(G), dim3(pe::W * pe::WPB), args, 0, nullptr));
    } else if (one_pass) pe::k_pe_solve<T, VTraits, D, CAN><<<G, pe::W * pe::WPB>>>(sa, vm, bk);
    else pe::k_pe_solve<T, VTraits, D, false><<<G, pe::W * pe::WPB>>>(sa, vm, bk);
  }
}
template <typename T, typename S, typename VTraits>
void VertexDescriptor<T, S, VTraits>::pose_engine_finish(const void *ctl, const int *k2l, int NV) {
  using namespace detail;
  if constexpr (can_mirror) pe::k_pe_finish<VTraits><<<blocks((size_t)NV), TPB>>>(static_cast<const pe::Ctl *>(ctl), k2l, NV, mirror.raw(), backup_ptr());
}

namespace detail {
// levenberg_marquardt (early_stop: levenberg_marquardt2) of a graph of ONE vertex descriptor whose factor descriptors are unary or binary
// on it, on the pose-graph engine.  The caller has run Graph::initialize_optimization(level, light) and holds the device mirror of the
// vertices.  Returns -1 (with res.declined) when the graph is not an engine configuration, 0 when the loop ran, 1 when a launch failed
// (the vertices are as the call found them).
template <typename T, typename S>
int pose_engine_run(Graph<T, S> *graph, const PoseEngineOptions &o, PoseEngineResult &res) {
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  auto t_last = t_begin;
  auto lap = [&](const char *what) { const auto now = clk::now(); res.detail += std::string(what) + " " + std::to_string(1e3 * std::chrono::duration<double>(now - t_last).count()) + " ms, "; t_last = now; };
  auto *vd = graph->get_vertex_descriptors()[0];
  auto &fds = graph->get_factor_descriptors();
  const size_t hessian_dim = graph->get_hessian_dimension();
  pe::VertexInfo vi;
  if (!vd->pose_engine_vertex(vi)) { res.declined = "vertex type is not plain data, or a tangent dimension above 7"; return -1; }
  if (!vi.mirrored) { res.declined = "no device mirror of the vertices"; return -1; }
  const int D = vi.dim, DD = D * D;
  const size_t nvl = vi.count;
  if (!vd->pose_engine_state) vd->pose_engine_state = std::make_shared<pe::Buffers<T>>();
  auto &bf = *std::static_pointer_cast<pe::Buffers<T>>(vd->pose_engine_state);
  const bool use_cache = !(getenv("GRAPHITE_POSE_CACHE") && atoi(getenv("GRAPHITE_POSE_CACHE")) == 0);
  // ---- the factor descriptors: all unary / binary on this vertex descriptor; their active factors numbered one after the other ----
  std::vector<pe::FactorInfo> fis(fds.size());
  std::vector<size_t> key_f, a0(fds.size() + 1, 0);
  for (size_t d = 0; d < fds.size(); ++d) {
    // (the symmetry check reads every precision matrix: skipped while the descriptor's epoch is the cached one)
    const bool known = use_cache && bf.have_structure && bf.key_f.size() == 4 * fds.size() && bf.key_f[4 * d] == (size_t)(uintptr_t)fds[d] && bf.key_f[4 * d + 1] == fds[d]->structure_epoch;
    if (!fds[d]->pose_engine_factor(fis[d], !known)) { res.declined = "a factor descriptor with more than two slots or an error dimension above 7"; return -1; }
    if (fis[d].vertex_descriptor != (const void *)vd) { res.declined = "a factor slot on another vertex descriptor"; return -1; }
    if (!fis[d].storage_is_scalar) { res.declined = "Jacobian storage type differs from the graph's scalar"; return -1; }
    if (!fis[d].symmetric) { res.declined = "a precision matrix is not symmetric"; return -1; }
    a0[d + 1] = a0[d] + fis[d].active;
    key_f.push_back((size_t)(uintptr_t)fds[d]); key_f.push_back(fis[d].epoch); key_f.push_back(fis[d].active); key_f.push_back((size_t)fis[d].active_digest);
  }
  const size_t na = a0[fds.size()];
  if (!na || !hessian_dim || hessian_dim % D) { res.declined = "nothing to optimise"; return -1; }
  if (na > 0x3fffffffu || nvl > 0x3fffffffu) { res.declined = "more than 2^30 factors or vertices"; return -1; }
  const uint8_t *state = vd->get_active_state();
  const size_t *hid = vd->get_hessian_ids();
  uint64_t dg = digest(0x706F7365ull, state, nvl);
  dg = digest(dg, hid, nvl * sizeof(size_t));
  const bool cached = use_cache && bf.have_structure && bf.key_f == key_f && bf.key_epoch_v == vi.epoch && bf.key_nvl == nvl && bf.key_dim == hessian_dim && bf.key_digest == dg;
  lap("checks");
  int NV = bf.c_NV, NVp = bf.c_NVp, lg = bf.c_lg, nslices = bf.c_nslices;
  size_t ngroups = bf.c_ngroups;
  if (!cached) {
    bf.have_structure = false;
    // ---- structure: engine vertices = the descriptor's vertices that have a column, in column order; wave-sliced entry lists ----
    NV = (int)(hessian_dim / D);
    std::vector<int> l2k(nvl, -1), k2l(NV, -1);
    for (size_t l = 0; l < nvl; ++l)
      if (is_vertex_active(state, l)) {
        const size_t k = hid[l] / D;
        if (hid[l] % D || k >= (size_t)NV || k2l[k] >= 0) { res.declined = "Hessian columns are not one block per vertex"; return -1; }
        l2k[l] = (int)k; k2l[k] = (int)l;
      }
    for (int k = 0; k < NV; ++k) if (k2l[k] < 0) { res.declined = "Hessian columns are not one block per vertex"; return -1; }
    std::vector<int> lij(2 * na);
    for (size_t d = 0; d < fds.size(); ++d) fds[d]->pose_engine_ids(lij.data() + 2 * a0[d]);
    std::vector<int> deg(NV, 0);
    size_t nentries = 0;
    for (size_t a = 0; a < na; ++a)
      for (int sd = 0; sd < 2; ++sd) {
        const int k = lij[2 * a + sd] >= 0 ? l2k[lij[2 * a + sd]] : -1;
        if (k >= 0) { ++deg[k]; ++nentries; }
      }
    // lanes per vertex: as many as keep the grid at about one 4-wave workgroup per CU, at most a quarter of the mean degree's worth
    // of idle lanes (1 << lg <= mean degree), at most 8
    lg = 0;
    {
      const double mean = (double)nentries / NV;
      while (lg < 3 && (double)(2 << lg) <= mean && (size_t)NV * (size_t)(2 << lg) <= (size_t)pe::W * pe::WPB * 256) ++lg;
      if (getenv("GRAPHITE_POSE_LPV")) { const int want = atoi(getenv("GRAPHITE_POSE_LPV")); lg = want >= 8 ? 3 : want >= 4 ? 2 : want >= 2 ? 1 : 0; }
    }
    const int LPV = 1 << lg, VPW = pe::W >> lg;
    nslices = (NV + VPW - 1) / VPW; NVp = (nslices * VPW + pe::W - 1) / pe::W * pe::W;
    std::vector<int> sbase(nslices + 1, 0);
    for (int w = 0; w < nslices; ++w) {
      int m = 0;
      for (int k = w * VPW; k < std::min(NV, (w + 1) * VPW); ++k) m = std::max(m, (deg[k] + LPV - 1) / LPV);
      sbase[w + 1] = sbase[w] + m;
    }
    ngroups = (size_t)sbase[nslices];
    if (ngroups * pe::W > 0x3fffffffu) { res.declined = "entry lists above 2^30 slots (a vertex of very high degree)"; return -1; }
    std::vector<int> enbr(ngroups * pe::W, -2), efac(ngroups * pe::W, -1), pos(2 * na, -1), fill(NV, 0);
    auto slot_of = [&](int k) { // entry e of vertex k: sub-lane e % LPV, group e / LPV of its slice
      const int e = fill[k]++, w = k / VPW;
      return (sbase[w] + e / LPV) * pe::W + (k % VPW) * LPV + e % LPV;
    };
    for (size_t a = 0; a < na; ++a) { // ascending factor order per (vertex, sub-lane): the order of the sums
      const int ki = lij[2 * a] >= 0 ? l2k[lij[2 * a]] : -1, kj = lij[2 * a + 1] >= 0 ? l2k[lij[2 * a + 1]] : -1;
      if (ki >= 0) { const int slot = slot_of(ki); pos[2 * a] = slot; enbr[slot] = kj; efac[slot] = (int)(2 * a); }
      if (kj >= 0) { const int slot = slot_of(kj); pos[2 * a + 1] = slot; enbr[slot] = ki; efac[slot] = (int)(2 * a + 1); }
    }
    lap("lists");
    bf.sbase.assign(sbase.data(), sbase.size()); bf.enbr.assign(enbr.data(), enbr.size()); bf.efac.assign(efac.data(), efac.size()); bf.k2l.assign(k2l.data(), k2l.size()); bf.pos.assign(pos.data(), pos.size()); bf.lij.assign(lij.data(), lij.size());
    bf.key_f = key_f; bf.key_epoch_v = vi.epoch; bf.key_nvl = nvl; bf.key_dim = hessian_dim; bf.key_digest = dg;
    bf.c_NV = NV; bf.c_NVp = NVp; bf.c_lg = lg; bf.c_nslices = nslices; bf.c_ngroups = ngroups;
    bf.have_structure = true;
  } else lap("structure cache hit (epochs + digests)");
  const int LPV = 1 << lg;
  bf.frec.resize_uninit(2 * na * (size_t)(2 * (2 * DD + D))); bf.B.resize_uninit((ngroups + 1) * DD * pe::W);
  const size_t per_vertex = (size_t)(2 * DD + 3 * D), per_vec = (size_t)6 * D;
  bf.vert.resize_uninit(per_vertex * NVp); bf.vec.resize_uninit(per_vec * NVp); bf.ex.resize_uninit((size_t)NVp * 2 * D);
  bf.dx.resize_uninit(hessian_dim);
  const size_t ntr = o.iterations + 1;
  bf.sums.resize_uninit((size_t)2 * pe::MAX_GRID * 2 * 2); bf.part_rho.resize_uninit((size_t)pe::MAX_GRID * pe::WPB); bf.tr.resize_uninit(2 * ntr); bf.clock.resize_uninit(ntr);
  std::vector<int> block0(fds.size() + 1, 0);
  for (size_t d = 0; d < fds.size(); ++d) block0[d + 1] = block0[d] + (int)((fis[d].active + pe::FTPB - 1) / pe::FTPB);
  const int total_blocks = block0[fds.size()];
  bf.part_chi2.resize_uninit((size_t)std::max(1, total_blocks));
  bf.ctl.resize_uninit(1); bf.fail.resize_uninit(1);
  lap("buffers + uploads");

  // ---- launch shapes ----
  int dev = 0, cus = 0;
  GRAPHITE_HIP(hipGetDevice(&dev));
  GRAPHITE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // (the one-pass form keeps a vertex's state in registers: only when the grid has a wave for every slice, and blocks of at most 4 x 4)
  const int per_cu_multi = vd->pose_engine_occupancy(false), per_cu_one = vd->pose_engine_occupancy(true);
  if (per_cu_multi < 1) { res.declined = "the solve kernel does not fit a compute unit"; return -1; }
  const int G_all = (nslices + pe::WPB - 1) / pe::WPB;
  const int max_grid = getenv("GRAPHITE_POSE_MAX_GRID") ? std::max(1, std::min(pe::MAX_GRID, atoi(getenv("GRAPHITE_POSE_MAX_GRID")))) : pe::MAX_GRID; // (tests: a small grid walks several slices per wave)
  const bool one_pass = per_cu_one >= 1 && G_all <= std::min(max_grid, per_cu_one * cus) && !(getenv("GRAPHITE_POSE_ONE_PASS") && atoi(getenv("GRAPHITE_POSE_ONE_PASS")) == 0);
  const int G = std::max(1, std::min(std::min(G_all, max_grid), (one_pass ? per_cu_one : per_cu_multi) * cus));
  lap("occupancy query");
  res.detail += std::to_string(NV) + " vertices x " + std::to_string(LPV) + " lanes, " + std::to_string(ngroups) + " entry groups, " + std::to_string(fds.size()) + " factor descriptor(s), grid " +
                std::to_string(G) + " x " + std::to_string(pe::W * pe::WPB) + (one_pass ? ", state in registers" : ", state in memory");
  pe::Ctl h{};
  h.mu = o.initial_damping; h.nu = 2; h.fresh = 1;
  GRAPHITE_HIP(hipMemcpy(bf.ctl.raw(), &h, sizeof(h), hipMemcpyHostToDevice));
  GRAPHITE_HIP(hipMemset(bf.fail.raw(), 0, sizeof(int)));
  GRAPHITE_HIP(hipMemsetAsync(bf.sums.raw(), 0, bf.sums.size() * sizeof(double), nullptr)); // no record carries a tag of this call
  // padding slots of the entry blocks (and the zero block behind them) are never written: the operator multiplies their zeros.  They stay
  // zero from call to call while the lists — and the buffer — are the ones that were cleared
  if (!cached || bf.B_cleared != (const void *)bf.B.raw() || bf.B_cleared_size != bf.B.size()) {
    GRAPHITE_HIP(hipMemsetAsync(bf.B.raw(), 0, bf.B.size() * sizeof(T), nullptr));
    bf.B_cleared = bf.B.raw(); bf.B_cleared_size = bf.B.size();
  }

  pe::SolveArgs<T> sa{};
  sa.NV = NV; sa.NVp = NVp; sa.lg = lg; sa.nslices = nslices; sa.sbase = bf.sbase.raw(); sa.enbr = bf.enbr.raw(); sa.k2l = bf.k2l.raw();
  sa.efac = bf.efac.raw(); sa.frec = bf.frec.raw(); sa.na = na; sa.B = bf.B.raw(); sa.zero_block = ngroups * (size_t)DD * pe::W;
  T *vp = bf.vert.raw();
  sa.Hs = vp; sa.Minv = vp + (size_t)DD * NVp; sa.s = vp + (size_t)2 * DD * NVp; sa.b = sa.s + (size_t)D * NVp; sa.dg = sa.b + (size_t)D * NVp;
  T *vv = bf.vec.raw();
  sa.x = vv; sa.xb = vv + (size_t)D * NVp; sa.r = vv + (size_t)2 * D * NVp; sa.t = vv + (size_t)3 * D * NVp; sa.p = vv + (size_t)4 * D * NVp; sa.y = vv + (size_t)5 * D * NVp;
  sa.ex = bf.ex.raw(); sa.sums = bf.sums.raw(); sa.part_rho = bf.part_rho.raw(); sa.fail = bf.fail.raw(); sa.ctl = bf.ctl.raw();
  sa.max_iter = o.pcg_max_iter; sa.identity_precond = o.identity_precond; sa.use_identity = o.use_identity; sa.scale_system = o.scale_system;
  sa.tol = o.pcg_tol; sa.rej = o.pcg_rej;
  const bool debug = getenv("GRAPHITE_POSE_DEBUG") && atoi(getenv("GRAPHITE_POSE_DEBUG")) != 0;
  if (debug) { bf.dbg.resize(64 + pe::MAX_GRID); sa.dbg = bf.dbg.raw(); }
  sa.var = getenv("GRAPHITE_POSE_VAR") ? atoi(getenv("GRAPHITE_POSE_VAR")) : 0;
  sa.timeout = (sa.var & 256) ? 2000000ll : 200000000ll;
  if ((sa.var & 256) && bf.prefer_cooperative) sa.var &= ~256; // (the test's failure happens once)
  sa.graph_b = graph->get_b().raw(); sa.graph_scales = graph->get_jacobian_scales().raw(); sa.dx = bf.dx.raw();
  pe::FactorArgs<T> fa{};
  fa.ctl = bf.ctl.raw(); fa.part = bf.part_chi2.raw(); fa.part_rho = bf.part_rho.raw(); fa.n_rho = G * pe::WPB;
  fa.na_all = na; fa.total_blocks = total_blocks; fa.tr_chi2 = bf.tr.raw(); fa.tr_mu = bf.tr.raw() + ntr; fa.tr_clock = bf.clock.raw(); fa.pos = bf.pos.raw(); fa.lij = bf.lij.raw();
  fa.frec = bf.frec.raw(); fa.early = o.early_stop ? 1 : 0;
  auto factor_pass = [&](int mode) {
    for (size_t d = 0; d < fds.size(); ++d) {
      if (!fis[d].active) continue;
      pe::FactorArgs<T> f = fa;
      f.na = fis[d].active; f.a0 = a0[d]; f.block0 = block0[d];
      fds[d]->pose_engine_launch(f, vi.mirror, mode);
    }
  };

  // The solve needs its whole grid resident (rendezvous inside the launch).  The grid is sized to fit (occupancy query above), so a
  // PLAIN launch into an idle device is resident; hipLaunchCooperativeKernel guarantees it but costs 22 us more per launch and 11 ms
  // at its first use in a process (measured, 10 k poses).  Plain by default; a rendezvous that times out (another process or stream
  // holds CUs) fails the call over to the generic kernels from the untouched start and makes later calls cooperative.
  // GRAPHITE_POSE_COOP=1 / 0 forces either.
  bool coop = bf.prefer_cooperative;
  if (getenv("GRAPHITE_POSE_COOP")) coop = atoi(getenv("GRAPHITE_POSE_COOP")) != 0;
  res.detail += coop ? ", cooperative launch" : ", plain launch";
  const size_t vbytes = vi.count * vi.vertex_bytes;
  bf.start.resize_uninit(vbytes);
  GRAPHITE_HIP(hipMemcpyAsync(bf.start.raw(), vi.mirror, vbytes, hipMemcpyDeviceToDevice, nullptr));
  res.setup_seconds = std::chrono::duration<double>(clk::now() - t_begin).count();
  const auto t_loop = clk::now();
  factor_pass(0);
  for (size_t i = 0; i < o.iterations; ++i) {
    vd->pose_engine_solve(sa, G, coop, one_pass);
    factor_pass(1);
    if (o.stop_flag) { // the caller may ask between iterations (levenberg_marquardt.hpp:232): keep the loop in step with the host
      sync();
      if (*o.stop_flag) { res.stop_bits |= 8; break; }
      GRAPHITE_HIP(hipMemcpy(&h, bf.ctl.raw(), sizeof(h), hipMemcpyDeviceToHost));
      if (h.stop) break;
    }
  }
  vd->pose_engine_finish(bf.ctl.raw(), bf.k2l.raw(), NV);
  sync();
  res.loop_seconds = std::chrono::duration<double>(clk::now() - t_loop).count();
  GRAPHITE_HIP(hipMemcpy(&h, bf.ctl.raw(), sizeof(h), hipMemcpyDeviceToHost));
  int failed = 0;
  GRAPHITE_HIP(hipMemcpy(&failed, bf.fail.raw(), sizeof(int), hipMemcpyDeviceToHost));
  if (failed || (h.stop & 16)) {
    GRAPHITE_HIP(hipMemcpy(vi.mirror, bf.start.raw(), vbytes, hipMemcpyDeviceToDevice)); // the vertices as the call found them
    bf.prefer_cooperative = true;
    res.stop_bits |= 16; res.ok = false;
    res.declined = "a rendezvous inside the solve timed out (its grid was not fully resident: the device is shared); later calls use a cooperative launch";
    return 1;
  }
  if (debug) { // the LAST solve's stamps
    const std::vector<long long> d = bf.dbg.to_host();
    std::cerr << "[graphite] pose-graph engine, last solve, workgroup 0 (us since its start; start | assemble+invert | sums | then per PCG iteration: operator | sums | update | sums ... | step):";
    for (int i = 1; i < (int)d[63] && i < 63; ++i) std::cerr << " " << (double)(d[i] - d[i - 1]) * 0.01;
    std::cerr << std::endl;
    std::vector<double> op(d.begin() + 64, d.begin() + 64 + G);
    std::cerr << "[graphite] pose-graph engine, operator phase of the last PCG iteration by workgroup (us): first " << op.front() * 0.01 << ", last " << op.back() * 0.01;
    { std::vector<double> q = op; std::sort(q.begin(), q.end()); std::cerr << ", min " << q.front() * 0.01 << ", median " << q[q.size() / 2] * 0.01 << ", p90 " << q[q.size() * 9 / 10] * 0.01 << ", max " << q.back() * 0.01; }
    { double a = 0, b = 0; int na_ = 0, nb_ = 0; for (int i = 0; i < G; ++i) { if (i + cus < G || i >= cus) { b += op[i]; ++nb_; } else { a += op[i]; ++na_; } }
      std::cerr << "; mean of workgroups alone on their CU " << (na_ ? a / na_ * 0.01 : 0.0) << " (" << na_ << "), sharing it " << (nb_ ? b / nb_ * 0.01 : 0.0) << " (" << nb_ << ")" << std::endl; }
  }
  const std::vector<double> tr = bf.tr.to_host();
  const std::vector<long long> ck = bf.clock.to_host();
  res.iterations_run = h.it; res.accepted = h.accepted; res.pcg_iterations = h.pcg_its; res.stop_bits |= h.stop;
  res.chi2.assign(tr.begin(), tr.begin() + h.it + 1);
  res.lambda.assign(tr.begin() + ntr, tr.begin() + ntr + h.it + 1);
  res.seconds.resize(h.it);
  for (int i = 0; i < h.it; ++i) res.seconds[i] = (double)(ck[i + 1] - ck[i]) * 1.0e-8; // the 100 MHz wall clock
  res.ok = !(h.stop & 1);
  return 0;
}
} // namespace detail

} // namespace graphite
