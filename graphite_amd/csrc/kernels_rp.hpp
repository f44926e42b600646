// Resident PCG — PCGSolver::solve (solver/pcg.hpp:61-232) + BlockJacobiPreconditioner::apply (preconditioner/block_jacobi.hpp:174-186)
// + Graph::backup_parameters / apply_update (graph.hpp:292-309) as ONE launch — gfx950.
//
// The multi-launch form (kernels_mf.hpp) pays, per inner iteration, three launches that each re-read what does not change
// while the loop runs: the four index / observation streams, the gathered point, the vectors x, r, p, z', the scales, the
// block-Jacobi inverses.  Here one workgroup of 512 threads per CU is resident for the whole solve and
//   * every lane OWNS up to RP_RO observations (camera-major order, the XCD-aware ranges of the other kernels): camera id,
//     point id, point-major position, the point X Y Z and the loss weight rho'(chi2) sit in LDS (one private column per thread:
//     conflict-free); the Jacobian is recomputed from the camera pack (scalar loads) as in k_pcg_operator;
//   * every thread OWNS up to RP_RV scalars of the system (tiles of 56 cameras / 170 points, dealt contiguously): x, r, p, z',
//     the column scale and the clamped diagonal stay in VGPRs (x trails by one update, so a rejected iteration needs no backup);
//   * what crosses workgroups per inner iteration is only: the direction halves [s.z' | s.p] per point (48-byte records,
//     gathered by the observations) and per camera, the per-observation point rows g4 (scattered to point-major slots, summed
//     in fixed order by the point's owner), the (wave, camera) segment sums, and the dot products (64 slots per scalar).
//     All of it is stored write-through and read with sc1 loads (cdna_hip_programming.md Guideline 16, form R1): no fence.
//   * two grid barriers per inner iteration (XCD-hierarchical arrival, per-group generation word), in the LAZY form of the
//     recurrence: the direction p_k = sigma_k z'_k + beta_k p_{k-1} is formed where it is used — by the observations as they
//     gather the two halves, by the owner in registers — so beta needs no pass of its own:
//         operator (needs sigma_k, beta_k)  --barrier A: den-->  update (alpha_k; r, z', the four dots)  --barrier B-->  ...
//   * the launch that finds the loop over applies the trial step from its registers (backup, x (+) dx.s, rho-denominator
//     partials, camera packs): apply_on_exit's work without a launch or a read of dx.
// Bounded spins: a grid that is not fully resident sets the pinned `fail` word and every workgroup leaves.
#pragma once
#include "kernels_mf.hpp"

namespace gr {

constexpr int RTPB = 512;             // threads per workgroup; ONE workgroup per CU
constexpr int RP_RO = 6;              // observation slots per lane
constexpr int RP_RV = 4;              // owned tiles per workgroup (one scalar per thread and tile)
constexpr int RP_PT = 168;            // points per tile: 21 per wave (63 lanes, a point's three scalars never straddle a wave)
constexpr int RP_CT = 56;             // cameras per tile: 7 per wave (63 lanes)
constexpr int RP_BAR_WORDS = 8 * 16 + 16 + 8 * 16; // group counters (a 64-byte line each) | top counter | generation words

typedef unsigned rp_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned rp_u32x2 __attribute__((ext_vector_type(2)));
constexpr int RP_SC1 = 16;            // aux bit of the buffer builtins: sc1 (write-through store / L1-bypassing load)

template <typename T> struct RpParams {
  // (the read-only arrays — structure in plain camera-major order, linearisation point, what k_finalize_bj left — are kernel parameters)
  int No, Nc, Np;
  int loss_kind;
  T loss_delta;
  const unsigned char *cam_fixed, *pt_fixed;
  // exchange buffers
  T *rec6;       // [Np][6]  s.z' | s.p of a point
  T *crec;       // [Nc][18] s.z' | s.p of a camera
  T *g4;         // [No][4]  point rows of one observation at its point-major slot (padded to 4)
  T *op_partial; // [nseg][9]
  // control
  int max_iter, use_identity, identity_precond;
  double tol, rej, mu;
  PcgState st;
  unsigned *bar;        // RP_BAR_WORDS, zero at launch
  volatile int *fail;   // pinned
  T *dx;                // [n] out: the step in scaled coordinates
  ApplyOnExit<T> ap;    // ap.cams == nullptr: no trial step
  const LmDev *lm;
  int var;              // diagnostic (timing only, GR_RP_VAR): 1 no g4 stores, 2 no segment stores, 4 no point gathers, 8 no camera-half loads, 16 no Jacobian, 32 no loads in the update, 64 no stores in the update, 128 loop control ignored
  long long *dbg;       // diagnostic (tools/rp_phases.py): wall-clock stamps of every workgroup at its phase boundaries, [G][64]; nullptr off
};

// ---- exchange accessors: everything another workgroup wrote in this launch is read sc1, everything it will read is stored sc1 ----
template <typename T> __device__ __forceinline__ T rp_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void rp_st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rp_rsrc(const void *p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// six scalars of record `idx`
__device__ __forceinline__ void rp_ld6(__amdgpu_buffer_rsrc_t r, int idx, double (&v)[6]) {
  const int off = idx * 48;
  const rp_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, RP_SC1), b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, RP_SC1),
                 c = __builtin_amdgcn_raw_buffer_load_b128(r, off + 32, 0, RP_SC1);
  const double2 da = __builtin_bit_cast(double2, a), db = __builtin_bit_cast(double2, b), dc = __builtin_bit_cast(double2, c);
  v[0] = da.x; v[1] = da.y; v[2] = db.x; v[3] = db.y; v[4] = dc.x; v[5] = dc.y;
}
__device__ __forceinline__ void rp_ld6(__amdgpu_buffer_rsrc_t r, int idx, float (&v)[6]) {
  const int off = idx * 24;
  const rp_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, RP_SC1);
  const rp_u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(r, off + 16, 0, RP_SC1);
  const float4 fa = __builtin_bit_cast(float4, a);
  const float2 fb = __builtin_bit_cast(float2, b);
  v[0] = fa.x; v[1] = fa.y; v[2] = fa.z; v[3] = fa.w; v[4] = fb.x; v[5] = fb.y;
}
// two consecutive scalars at element offset `e`
__device__ __forceinline__ void rp_st2(__amdgpu_buffer_rsrc_t r, int e, double v0, double v1) {
  double2 d; d.x = v0; d.y = v1;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rp_u32x4, d), r, e * 8, 0, RP_SC1);
}
__device__ __forceinline__ void rp_st2(__amdgpu_buffer_rsrc_t r, int e, float v0, float v1) {
  float2 d; d.x = v0; d.y = v1;
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(rp_u32x2, d), r, e * 4, 0, RP_SC1);
}
// [v0 v1 v2 0] at slot a of g4
__device__ __forceinline__ void rp_st_g4(__amdgpu_buffer_rsrc_t r, int a, double v0, double v1, double v2) {
  rp_st2(r, 4 * a, v0, v1);
  rp_st2(r, 4 * a + 2, v2, 0.0);
}
__device__ __forceinline__ void rp_st_g4(__amdgpu_buffer_rsrc_t r, int a, float v0, float v1, float v2) {
  float4 d; d.x = v0; d.y = v1; d.z = v2; d.w = 0.f;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rp_u32x4, d), r, a * 16, 0, RP_SC1);
}
__device__ __forceinline__ float rp_readlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ double rp_readlane(double v, int l) {
  int2 u = __builtin_bit_cast(int2, v);
  u.x = __builtin_amdgcn_readlane(u.x, l); u.y = __builtin_amdgcn_readlane(u.y, l);
  return __builtin_bit_cast(double, u);
}

// LDS image of one workgroup (dynamic): byte offsets, T-dependent
template <typename T> struct RpLds {
  static constexpr size_t oc = 0;                                      // int [RO][512] camera id (-1: no observation)
  static constexpr size_t ol = oc + sizeof(int) * RP_RO * RTPB;        // int [RO][512] point id
  static constexpr size_t oa = ol + sizeof(int) * RP_RO * RTPB;        // int [RO][512] point-major slot
  static constexpr size_t xyz = oa + sizeof(int) * RP_RO * RTPB;       // T   [RO][3][512] the observed point
  static constexpr size_t ow = xyz + sizeof(T) * RP_RO * 3 * RTPB;     // T   [RO][512] loss weight rho'(e^2) at the linearisation point
  static constexpr size_t sg = ow + sizeof(T) * RP_RO * RTPB;          // int [RO][512] segment slot of the block's first camera
  static constexpr size_t red = sg + sizeof(int) * RP_RO * RTPB;       // double [4][8]
  static constexpr size_t flag = red + sizeof(double) * 32;            // int
  static constexpr size_t bytes = flag + 16;
};

// four sums over the 512 threads at once, the same values (same order) in every thread; red: 32 doubles of LDS
__device__ __forceinline__ void rp_block_allsum4(double (&v)[4], double *red) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = wave_allsum(v[i]);
  __syncthreads(); // (the previous reduction's readers are done)
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red[8 * i + (threadIdx.x >> 6)] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) { const double *r = red + 8 * i; v[i] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7])); }
}
// 64 partial sums of a dot product, written by other workgroups in this launch (whole wave must call)
__device__ __forceinline__ double rp_slot_sum(const double *base) {
  return wave_allsum(__hip_atomic_load(&base[(size_t)(threadIdx.x & 63) * SS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// NV dot products after a barrier: ONE wave of the workgroup reads the 64 slots of each (all 2 048 waves of the grid reading the same
// 512-byte records at the same instant took 3.5 us: one memory channel), the totals reach the other waves through LDS
template <int NV> __device__ __forceinline__ void rp_slot_sums_block(const double *const (&base)[NV], double (&out)[NV], double *red) {
  if (threadIdx.x < 64) {
    double v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __hip_atomic_load(&base[i][(size_t)threadIdx.x * SS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_allsum(v[i]);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) red[i] = v[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) out[i] = red[i];
  __syncthreads(); // (red is reused by the next reduction)
}
__device__ __forceinline__ void rp_slot_add(double *base, double v) {
  (void)__hip_atomic_fetch_add(&base[(size_t)(blockIdx.x & (NS - 1)) * SS], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Grid barrier.  Arrival: every storing wave drains its write-through stores, the workgroup meets, ONE lane adds to its group's
// counter (group = blockIdx % 8: the workgroups that share an XCD; a label, correctness does not depend on it); the group's last
// arriver adds to the top counter, the last of those publishes the epoch to the eight generation words; a workgroup polls its own
// group's word.  Counters are monotonic within a launch (epoch = number of barriers so far) and zero at launch.
__device__ __forceinline__ bool rp_barrier(unsigned *bar, unsigned epoch, volatile int *fail, int *s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7u, gsz = (gridDim.x >> 3) + ((gridDim.x & 7u) > x ? 1u : 0u);
    unsigned *grp = bar + 16 * x, *top = bar + 128, *gen = bar + 144;
    const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == epoch * gsz) {
      const unsigned ngrp = gridDim.x < 8u ? gridDim.x : 8u;
      const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1u == epoch * ngrp) {
#pragma unroll
        for (int q = 0; q < 8; ++q) __hip_atomic_store(gen + 16 * q, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    int ok = 1;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(gen + 16 * x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 200000000ll) { // 2 s of the 100 MHz clock: not every workgroup of the launch is resident
        *fail = 1;
        __threadfence_system();
        ok = 0;
        break;
      }
    }
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

// The arrays this launch only READS come as __restrict__ kernel parameters of their own (not through the parameter record): only then
// does the compiler fetch wave-uniform ones with scalar loads into SGPRs; through the record they were vector loads (48 VGPRs for a
// pack) whose waits also drained the gathers in flight.
//
// Latency discipline (measured: with EVERY global access removed the first form of this kernel still took 11 us per phase):
//  * an observation block's dependent chain holds no memory round trip: everything it needs — the point's two direction halves, the
//    camera's two halves, the camera PACK (one vector load of 24 lanes, moved to SGPRs by v_readlane when used) — is requested while
//    the previous block computes, and nothing in between waits on a vector-memory counter;
//  * the owners' block-Jacobi products and record pieces go through lane shuffles (a vertex's scalars sit in ONE wave: 63 lanes = 21
//    points = 7 cameras), not through LDS and a workgroup barrier per unit; the dots of a phase share one block reduction.
template <typename T, typename JT>
__global__ void __launch_bounds__(RTPB, 2)
k_pcg_resident(const RpParams<T> P, const int *__restrict__ cam_cm, const int *__restrict__ pt_cm, const int *__restrict__ pos_cm,
               const T *__restrict__ obs_cm, const int *__restrict__ blk_seg, const int *__restrict__ seg_slot, const int *__restrict__ pt_ptr,
               const int *__restrict__ cam_seg_ptr, const T *__restrict__ pts, const T *__restrict__ pack, const T *__restrict__ scales,
               const T *__restrict__ diag, const T *__restrict__ MinvC, const T *__restrict__ MinvP, const T *__restrict__ r0,
               const T *__restrict__ z0, const T *__restrict__ zs0) {
  double mu = P.mu;
  if (P.lm) { if (P.lm->stop) return; mu = P.lm->mu; }
  extern __shared__ __align__(16) unsigned char rp_smem[];
  using L = RpLds<T>;
  int *const l_c = reinterpret_cast<int *>(rp_smem + L::oc), *const l_l = reinterpret_cast<int *>(rp_smem + L::ol), *const l_a = reinterpret_cast<int *>(rp_smem + L::oa);
  T *const l_xyz = reinterpret_cast<T *>(rp_smem + L::xyz), *const l_w = reinterpret_cast<T *>(rp_smem + L::ow);
  int *const l_sg = reinterpret_cast<int *>(rp_smem + L::sg);
  double *const l_red = reinterpret_cast<double *>(rp_smem + L::red);
  int *const l_flag = reinterpret_cast<int *>(rp_smem + L::flag);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x, b = blockIdx.x;
  const int No = P.No, Nc = P.Nc, Np = P.Np;
  const unsigned pose_dim = 9u * (unsigned)Nc;
  const PcgState &st = P.st;
  if (b == 0 && tid == 0 && st.ts) st.ts[0] = wall_clock64();
  int n_stamp = 0;
  auto stamp = [&]() __attribute__((always_inline)) { if (P.dbg && tid == 0 && n_stamp < 63) P.dbg[(size_t)b * 64 + n_stamp++] = wall_clock64(); };
  stamp();

  // ---- ownership ----------------------------------------------------------------------------------------------------------
  // observations: the XCD-aware wave-block ranges of xcd_obs_range, 512 per step
  int j0, jlim;
  {
    const int nb = G >> 3, x = b & 7, bi = b >> 3;
    const int nblk = (No + 63) >> 6;
    const int x0 = (int)((long long)x * nblk / 8), x1 = (int)((long long)(x + 1) * nblk / 8);
    const int b0 = x0 + (int)((long long)bi * (x1 - x0) / nb), b1 = x0 + (int)((long long)(bi + 1) * (x1 - x0) / nb);
    j0 = b0 << 6;
    const long long lim = (long long)b1 << 6;
    jlim = lim < (long long)No ? (int)lim : No;
  }
  const int niter = jlim > j0 ? (jlim - j0 + RTPB - 1) / RTPB : 0; // <= RP_RO (host)
  // system scalars: tiles of 56 cameras / 168 points as UNITS of one list, the camera tiles spread evenly through it (every
  // `cstride`-th unit: a camera tile costs more than a point tile, no workgroup should hold two), dealt contiguously.
  // Inside a tile: wave w, lane < 63 -> camera 7 w + lane / 9, entry lane % 9  |  point 21 w + lane / 3, entry lane % 3.
  const int cam_tiles = (Nc + RP_CT - 1) / RP_CT, pt_tiles = (Np + RP_PT - 1) / RP_PT, ntile = cam_tiles + pt_tiles;
  const int cstride = ntile / cam_tiles;
  const int ut0 = (int)((long long)b * ntile / G), ut1 = (int)((long long)(b + 1) * ntile / G); // ut1 - ut0 <= RP_RV (host)
  const unsigned li = (unsigned)lane % 3u, lbase3 = (unsigned)lane - li;      // point entry, first lane of the point
  const unsigned ci = (unsigned)lane % 9u, lbase9 = (unsigned)lane - ci;      // camera entry, first lane of the camera

  const __amdgpu_buffer_rsrc_t r_rec = rp_rsrc(P.rec6, sizeof(T) * 6 * (size_t)Np), r_g4 = rp_rsrc(P.g4, sizeof(T) * 4 * (size_t)No);

  // per owned unit: scalar index t into the system (valid when `on`), the run [sa0, sa1) of its segment sums / observation slots
  T sx[RP_RV], sr[RP_RV], sp[RP_RV], sz[RP_RV]; // (the scale and the clamped diagonal are re-read per iteration: plain loads, in flight with the sums)
  int sa0[RP_RV], sa1[RP_RV];
  unsigned st_[RP_RV];
  unsigned onmask = 0, fixmask = 0, cammask = 0, usedmask = 0;
#pragma unroll
  for (int q = 0; q < RP_RV; ++q) {
    sx[q] = sr[q] = sp[q] = sz[q] = T(0); sa0[q] = sa1[q] = 0; st_[q] = 0;
    const int u = ut0 + q;
    if (u >= ut1) continue;
    usedmask |= 1u << q;
    const bool is_cam = (u % cstride == 0) && (u / cstride < cam_tiles);
    const int ncam_before = min(cam_tiles, (u + cstride - 1) / cstride);
    if (is_cam) {
      cammask |= 1u << q;
      const unsigned c = (unsigned)(u / cstride) * RP_CT + 7u * (unsigned)wave + (unsigned)lane / 9u;
      const unsigned t = 9u * c + ci;
      st_[q] = t;
      if (lane < 63 && c < (unsigned)Nc) {
        onmask |= 1u << q;
        sr[q] = r0[t]; sz[q] = z0[t];
        sa0[q] = cam_seg_ptr[c]; sa1[q] = cam_seg_ptr[c + 1];
        if (P.cam_fixed && P.cam_fixed[c]) fixmask |= 1u << q;
      }
    } else {
      const unsigned l = (unsigned)(u - ncam_before) * RP_PT + 21u * (unsigned)wave + (unsigned)lane / 3u;
      const unsigned t = pose_dim + 3u * l + li;
      st_[q] = t;
      if (lane < 63 && l < (unsigned)Np) {
        onmask |= 1u << q;
        sr[q] = r0[t]; sz[q] = z0[t];
        sa0[q] = pt_ptr[l]; sa1[q] = pt_ptr[l + 1];
        if (P.pt_fixed && P.pt_fixed[l]) fixmask |= 1u << q;
      }
    }
  }

  // ---- the PCG start's dots (pcg.hpp:108-127): k_finalize_bj's per-workgroup partials, summed in the same order by every workgroup ----
  double rr, rzp, zdz;
  {
    double s0 = 0, s1 = 0, s2 = 0;
    for (int i = tid; i < st.n_part0; i += RTPB) { s0 += st.part0[i]; s1 += st.part0[(size_t)st.n_part0 + i]; s2 += st.part0[2 * (size_t)st.n_part0 + i]; }
    double v4[4] = {s0, s1, s2, 0.0};
    rp_block_allsum4(v4, l_red);
    rr = v4[0]; rzp = v4[1]; zdz = v4[2];
  }
  stamp(); // prologue done
  T sigma = (T)(1.0 / (double)(T)sqrt((double)(T)rr)), beta = T(0);
  T rz = (T)rzp * sigma; // the reference's r.z with z = Minv (r / |r|)
  double pdp = (double)sigma * (double)sigma * zdz, rz0 = __builtin_inf();
  int iters = 0;
  unsigned epoch = 0;
  T alpha_pend = T(0); // x trails by one update: x += alpha_pend p is applied when the NEXT iteration has not rejected it

  // ---- one 64-observation block of a wave: u = J p, the camera rows per (wave, camera) segment, the point rows to the point's slot ----
  // What the block's FIRST camera needs was requested a block ahead: pkraw (lanes 0..23: its pack), seg0 (its segment slot) and,
  // except in iteration 0, craw (lanes 0..8 / 9..17: s.z'_c / s.p_c).  Further cameras of the block (one block in six has a second)
  // fetch theirs here.  first: direction sigma s.z'_0 from k_finalize_bj's zs; the loss weight is taken from the residual and returned.
  // The outputs of a block are STORED ONE BLOCK LATER (Pend; flush_pend): a wave's `s_waitcnt vmcnt` for the next block's gathers also
  // waits for every store issued before it (loads and stores return out of order with respect to each other, so the compiler waits
  // for all), and a write-through store is acknowledged late; issued right after the next fetch, the stores have a whole block's
  // arithmetic to complete before anything waits again (ablation: the g4 stores cost 4.5 of the phase's 22 us where they stood).
  struct Pend { T g0, g1, g2, tot; int a, seg; bool hasg, hasp; };
  auto flush_pend = [&](Pend &pd) __attribute__((always_inline)) {
    if (pd.hasg && !(P.var & 1)) rp_st_g4(r_g4, pd.a, pd.g0, pd.g1, pd.g2);
    if (pd.hasp && (lane & 3) == 0 && (lane >> 2) < 9 && !(P.var & 2)) rp_st(&P.op_partial[9 * (size_t)pd.seg + (lane >> 2)], pd.tot);
    pd.hasg = false; pd.hasp = false;
  };
  auto obs_math = [&](const int j, const bool valid, const int c, const int a, const T ox, const T oy, T wl, const T X, const T Y, const T Z,
                      const T pl0, const T pl1, const T pl2, const bool first, T craw, const T pkraw, const int seg0, double &den, Pend &pend) __attribute__((always_inline)) -> T {
    unsigned long long remaining = __ballot(valid);
    if (!remaining) return wl;
    bool head = true;
    int segf = 0;
    while (remaining) {
      const int leader = __builtin_ctzll(remaining);
      const int cl = __builtin_amdgcn_readlane(c, leader);
      const bool mine = valid && c == cl;
      int segl;
      T pk[PACK], pc[9];
      if (head) {
        segl = __builtin_amdgcn_readfirstlane(seg0);
#pragma unroll
        for (int i = 0; i < PACK; ++i) pk[i] = rp_readlane(pkraw, i);
      } else {
        if (segf == 0) segf = blk_seg[__builtin_amdgcn_readfirstlane(j >> 6)] + 1;
        segl = seg_slot[segf++];
        load_pack(pack, cl, pk);
      }
      if (first) {
#pragma unroll
        for (int i = 0; i < 9; ++i) pc[i] = sigma * zs0[9 * (size_t)cl + i];
      } else {
        if (!head) craw = lane < 18 ? rp_ld(P.crec + 18 * (size_t)cl + lane) : T(0);
        const T cdir = lane < 9 ? sigma * craw : beta * craw;
#pragma unroll
        for (int i = 0; i < 9; ++i) pc[i] = rp_readlane(cdir, i) + rp_readlane(cdir, 9 + i);
      }
      head = false;
      T e0, e1, Jc[18], Jp[6];
      if (P.var & 16) {
        e0 = ox; e1 = oy;
#pragma unroll
        for (int i = 0; i < 18; ++i) Jc[i] = X + pk[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) Jp[i] = Y - pk[i];
      } else
      bal_linearize_j<T, JT>(pk, X, Y, Z, ox, oy, e0, e1, Jc, Jp);
      if (first && mine) wl = loss_drho(P.loss_kind, P.loss_delta, e0 * e0 + e1 * e1);
      const T w = mine ? wl : T(0);
      T u0 = Jp[0] * pl0 + Jp[2] * pl1 + Jp[4] * pl2;
      T u1 = Jp[1] * pl0 + Jp[3] * pl1 + Jp[5] * pl2;
#pragma unroll
      for (int i = 0; i < 9; ++i) { u0 += Jc[2 * i] * pc[i]; u1 += Jc[2 * i + 1] * pc[i]; }
      if (mine) den += (double)(w * (u0 * u0 + u1 * u1));
      u0 *= w; u1 *= w;
      T m[16];
#pragma unroll
      for (int i = 0; i < 9; ++i) m[i] = mine ? Jc[2 * i] * u0 + Jc[2 * i + 1] * u1 : T(0);
#pragma unroll
      for (int i = 9; i < 16; ++i) m[i] = T(0);
      if (mine) { pend.g0 = Jp[0] * u0 + Jp[1] * u1; pend.g1 = Jp[2] * u0 + Jp[3] * u1; pend.g2 = Jp[4] * u0 + Jp[5] * u1; pend.a = a; pend.hasg = true; }
      const T tot = wave_transpose_sum<T, 16>(m, lane);
      if (pend.hasp) { // an earlier camera of this block is still held: out it goes (one block in six has a second camera)
        if ((lane & 3) == 0 && (lane >> 2) < 9 && !(P.var & 2)) rp_st(&P.op_partial[9 * (size_t)pend.seg + (lane >> 2)], pend.tot);
      }
      pend.tot = tot; pend.seg = segl; pend.hasp = true;
      remaining &= ~__ballot(mine);
    }
    return wl;
  };
  // the pack of the block's first camera: one vector load, lanes 0..23
  auto pack_lane = [&](const int c, const bool valid) __attribute__((always_inline)) -> T {
    const unsigned long long bal = __ballot(valid);
    T v = T(0);
    if (bal) {
      const int cl = __builtin_amdgcn_readlane(c, __builtin_ctzll(bal));
      if (lane < PACK) v = pack[PACK * (size_t)cl + lane];
    }
    return v;
  };

  // ---- operator of iteration 0: loads what the lane owns (indices, point, loss weight, segment slot -> LDS), direction sigma z'_0 ----
  // two stages ahead: the index streams of block s + 2 and the point / direction / pack fetches of block s + 1 fly under the arithmetic of block s
  double den = 0;
  if (rzp != 0.0 && P.max_iter > 0) {
    struct Idx { int c, l, a, sf; T ox, oy; };
    struct Gat { T X, Y, Z, z0, z1, z2, pk; int sg; };
    auto load_idx = [&](const int s) __attribute__((always_inline)) -> Idx {
      Idx r{-1, 0, 0, 0, T(0), T(0)};
      const int j = j0 + s * RTPB + tid;
      if (s < niter && j < jlim) { r.c = cam_cm[j]; r.l = pt_cm[j]; r.a = pos_cm ? pos_cm[j] : j; r.ox = obs_cm[2 * (size_t)j]; r.oy = obs_cm[2 * (size_t)j + 1]; r.sf = blk_seg[j >> 6]; }
      return r;
    };
    auto gather = [&](const Idx &ix) __attribute__((always_inline)) -> Gat {
      Gat g{T(0), T(0), T(0), T(0), T(0), T(0), T(0), 0};
      if (ix.c >= 0) {
        const T *pp = pts + 3 * (size_t)ix.l, *zl = zs0 + pose_dim + 3 * (size_t)ix.l;
        g.X = pp[0]; g.Y = pp[1]; g.Z = pp[2]; g.z0 = zl[0]; g.z1 = zl[1]; g.z2 = zl[2];
        g.sg = seg_slot[ix.sf];
      }
      g.pk = pack_lane(ix.c, ix.c >= 0);
      return g;
    };
    Idx i_cur = load_idx(0), i_nxt = load_idx(1);
    Gat g_cur = gather(i_cur);
    Pend pend{T(0), T(0), T(0), T(0), 0, 0, false, false};
#pragma unroll 1
    for (int s = 0; s < niter; ++s) {
      const Idx ix = i_cur;
      const Gat g = g_cur;
      i_cur = i_nxt;
      g_cur = gather(i_cur);   // block s + 1: its indices arrived during block s - 1
      i_nxt = load_idx(s + 2);
      flush_pend(pend);        // block s - 1's outputs
      const int j = j0 + s * RTPB + tid;
      const T wl = obs_math(j, ix.c >= 0, ix.c, ix.a, ix.ox, ix.oy, T(0), g.X, g.Y, g.Z, sigma * g.z0, sigma * g.z1, sigma * g.z2, true, T(0), g.pk, g.sg, den, pend);
      l_c[s * RTPB + tid] = ix.c; l_l[s * RTPB + tid] = ix.l; l_a[s * RTPB + tid] = ix.a; l_sg[s * RTPB + tid] = g.sg;
      l_xyz[(3 * s) * RTPB + tid] = g.X; l_xyz[(3 * s + 1) * RTPB + tid] = g.Y; l_xyz[(3 * s + 2) * RTPB + tid] = g.Z;
      l_w[s * RTPB + tid] = wl;
    }
    flush_pend(pend);
  }

  for (int k = 0; k < P.max_iter; ++k) {
    if (rzp == 0.0) break; // pcg.hpp:133
    if (k > 0) {
      den = 0;
      // the fetches of block s + 1 are in flight under the arithmetic of block s
      T gv[6], craw = T(0), pkraw = T(0);
      int cc = -1;
      auto fetch = [&](const int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) gv[i] = T(0);
        craw = T(0);
        cc = l_c[s * RTPB + tid];
        const bool valid = cc >= 0;
        if (valid && !(P.var & 4)) rp_ld6(r_rec, l_l[s * RTPB + tid], gv);
        const unsigned long long bal = __ballot(valid);
        pkraw = T(0);
        if (bal) {
          const int cl = __builtin_amdgcn_readlane(cc, __builtin_ctzll(bal));
          if (lane < 18 && !(P.var & 8)) craw = rp_ld(P.crec + 18 * (size_t)cl + lane);
          if (lane < PACK) pkraw = pack[PACK * (size_t)cl + lane];
        }
      };
      if (niter > 0) fetch(0);
      Pend pend{T(0), T(0), T(0), T(0), 0, 0, false, false};
#pragma unroll 1
      for (int s = 0; s < niter; ++s) {
        const int j = j0 + s * RTPB + tid;
        const int c = cc;
        const T cr = craw, pr = pkraw;
        const T pl0 = sigma * gv[0] + beta * gv[3], pl1 = sigma * gv[1] + beta * gv[4], pl2 = sigma * gv[2] + beta * gv[5];
        if (s + 1 < niter) fetch(s + 1);
        flush_pend(pend); // block s - 1's outputs: behind the next fetch, a block's arithmetic ahead of the next wait
        (void)obs_math(j, c >= 0, c, l_a[s * RTPB + tid], T(0), T(0), l_w[s * RTPB + tid], l_xyz[(3 * s) * RTPB + tid], l_xyz[(3 * s + 1) * RTPB + tid],
                       l_xyz[(3 * s + 2) * RTPB + tid], pl0, pl1, pl2, false, cr, pr, l_sg[s * RTPB + tid], den, pend);
      }
      flush_pend(pend);
    }
    {
      double v4[4] = {den, 0.0, 0.0, 0.0};
      rp_block_allsum4(v4, l_red);
      den = v4[0];
    }
    stamp(); // operator done
    if (tid == 0) rp_slot_add(st.slots(k, DEN), den);
    // what the update needs that no workgroup writes in this launch (column scale, clamped diagonal, the point's row of its inverse) is
    // requested BEFORE the barrier: its latency passes under the wait (3 us stood between "barrier passed" and "den known" for them)
    T sc[RP_RV], dg[RP_RV], mrow[RP_RV][3];
#pragma unroll
    for (int q = 0; q < RP_RV; ++q) {
      sc[q] = dg[q] = T(1); mrow[q][0] = mrow[q][1] = mrow[q][2] = T(0);
      if (!((onmask >> q) & 1u)) continue;
      const unsigned t = st_[q];
      sc[q] = scales[t];
      if (!P.use_identity) dg[q] = diag[t];
      if (!((cammask >> q) & 1u) && !P.identity_precond) { // row li of the point's inverse
        const T *M = MinvP + 9 * (size_t)((t - pose_dim) / 3u);
        mrow[q][0] = M[li]; mrow[q][1] = M[li + 3]; mrow[q][2] = M[li + 6];
      }
    }
    if (!rp_barrier(P.bar, ++epoch, P.fail, l_flag)) return;
    stamp(); // barrier A passed

    // ---- update (pcg.hpp:166-195): alpha, x, r, z' = Minv r, the four dots; the owner forms p_k in registers ---------------
    // every unit's loads are issued before the first is used: ONE exposed round trip per phase, not one per unit
    constexpr int RG = 4; // run entries fetched up front (a point has 4.3 observations on average); longer runs: a loop (eight: 165 spilled VGPRs)
    T gq[RP_RV][RG];
#pragma unroll
    for (int q = 0; q < RP_RV; ++q) {
#pragma unroll
      for (int u = 0; u < RG; ++u) gq[q][u] = T(0);
      if (!((onmask >> q) & 1u)) continue;
      const bool is_cam = (cammask >> q) & 1u;
      const T *src = is_cam ? P.op_partial + ci : P.g4 + li;
      const int stride = is_cam ? 9 : 4;
#pragma unroll
      for (int u = 0; u < RG; ++u) if (sa0[q] + u < sa1[q] && !(P.var & 32)) gq[q][u] = rp_ld(src + (size_t)stride * (size_t)(sa0[q] + u));
    }
    double den_tot;
    { const double *const sl[1] = {st.slots(k, DEN)}; double o[1]; rp_slot_sums_block<1>(sl, o, l_red); den_tot = o[0]; }
    const T alpha = rz / (T)(den_tot + mu * pdp);
    if (P.var & 256) stamp(); // (fine stamps) den known
    double dots[4] = {0.0, 0.0, 0.0, 0.0}; // r.r, r.z', p.D.z', z'.D.z'
#pragma unroll
    for (int q = 0; q < RP_RV; ++q) {
      if (!((usedmask >> q) & 1u)) break;
      const bool on = (onmask >> q) & 1u, is_cam = (cammask >> q) & 1u;
      const unsigned t = st_[q];
      T rn = T(0), pv = T(0), mc[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) mc[i] = T(0);
      if (on) {
        if (is_cam && !P.identity_precond) { // row of the camera's inverse: in flight while the sums finish
          const T *M = MinvC + 81 * (size_t)(t / 9u) + ci;
#pragma unroll
          for (int i = 0; i < 9; ++i) mc[i] = M[9 * i];
        }
        // rows of J^T rho' J p: the camera's segment sums / the point's observations, in fixed order
        double raw = 0;
#pragma unroll
        for (int u = 0; u < RG; ++u) if (sa0[q] + u < sa1[q]) raw += (double)gq[q][u];
        if (sa0[q] + RG < sa1[q] && !(P.var & 32)) { // long runs: the rest, eight loads in flight
          const T *src = is_cam ? P.op_partial + ci : P.g4 + li;
          const int stride = is_cam ? 9 : 4;
          const int a_end = sa1[q];
          for (int a = sa0[q] + RG; a < a_end; a += 8) {
            T gr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) gr[u] = a + u < a_end ? rp_ld(src + (size_t)stride * (size_t)(a + u)) : T(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) if (a + u < a_end) raw += (double)gr[u];
          }
        }
        if ((fixmask >> q) & 1u) raw = 0.0;
        sx[q] = alpha_pend * sp[q] + sx[q]; // the previous iteration's update of x: it was not rejected
        pv = sigma * sz[q] + beta * sp[q];
        sp[q] = pv;
        const T v2 = sc[q] * (T)raw + (T)mu * dg[q] * pv;
        rn = -alpha * v2 + sr[q];
        sr[q] = rn;
      }
      // z' = Minv r: the vertex's entries of r from its lanes (whole wave shuffles)
      T zn = T(0);
      if (is_cam) {
        T acc = T(0);
#pragma unroll
        for (int i = 0; i < 9; ++i) acc += mc[i] * __shfl(rn, (int)lbase9 + i, 64);
        zn = acc;
      } else {
        const T ra = __shfl(rn, (int)lbase3, 64), rb = __shfl(rn, (int)lbase3 + 1, 64), rc = __shfl(rn, (int)lbase3 + 2, 64);
        zn = (T)((double)mrow[q][0] * (double)ra + (double)mrow[q][1] * (double)rb + (double)mrow[q][2] * (double)rc);
      }
      if (P.identity_precond) zn = rn;
      if (!on) zn = T(0);
      if (on) {
        sz[q] = zn;
        const T d = dg[q];
        dots[0] += (double)(rn * rn); dots[1] += (double)(rn * zn); dots[2] += (double)(d * pv * zn); dots[3] += (double)(d * zn * zn);
      }
      // the two halves of the next direction, for the observations
      const T zs = sc[q] * zn, ps = sc[q] * pv;
      if (is_cam) {
        if (on && !(P.var & 64)) {
          const unsigned c = t / 9u;
          rp_st(P.crec + 18 * (size_t)c + ci, zs);
          rp_st(P.crec + 18 * (size_t)c + 9 + ci, ps);
        }
      } else { // piece li of the point's record [zs0 zs1 | zs2 ps0 | ps1 ps2]: one store per lane, the tile's records are contiguous
        const int s0 = (int)lbase3 + (li == 0 ? 0 : li == 1 ? 2 : 1), s1 = (int)lbase3 + (li == 0 ? 1 : li == 1 ? 0 : 2);
        const T za = __shfl(zs, s0, 64), pa = __shfl(ps, s0, 64), zb = __shfl(zs, s1, 64), pb = __shfl(ps, s1, 64);
        const T v0 = li == 2 ? pa : za, v1 = li == 0 ? zb : pb;
        if (on && !(P.var & 64)) rp_st2(r_rec, (int)(2u * (t - pose_dim)), v0, v1); // element 6 l + 2 li
      }
    }
    alpha_pend = alpha;
    stamp(); // update done
    rp_block_allsum4(dots, l_red);
    if (P.var & 256) stamp(); // (fine stamps) dots reduced in the workgroup
    if (tid < 4) rp_slot_add(st.slots(k + 1, tid == 0 ? RR : tid == 1 ? RZP : tid == 2 ? PDZ : ZDZ), tid == 0 ? dots[0] : tid == 1 ? dots[1] : tid == 2 ? dots[2] : dots[3]);
    if (!rp_barrier(P.bar, ++epoch, P.fail, l_flag)) return;
    stamp(); // barrier B passed

    // ---- loop control (pcg.hpp:197-229), the same in every thread -----------------------------------------------------------
    double pdz;
    {
      const double *const sl[4] = {st.slots(k + 1, RR), st.slots(k + 1, RZP), st.slots(k + 1, PDZ), st.slots(k + 1, ZDZ)};
      double o[4];
      rp_slot_sums_block<4>(sl, o, l_red);
      rr = o[0]; rzp = o[1]; pdz = o[2]; zdz = o[3];
    }
    iters = k + 1;
    const T sigma_n = (T)(1.0 / (double)(T)sqrt((double)(T)rr));
    const T rz_new = (T)rzp * sigma_n;
    if (P.var & 128) { beta = T(0.5); continue; }
    if (fabs((double)rz_new) > P.rej * rz0 || rz_new != rz_new) { alpha_pend = T(0); break; } // rejected: x stays where it was, the loop ends
    rz0 = fmin(rz0, fabs((double)rz_new));
    beta = rz_new / rz;
    sigma = sigma_n;
    pdp = (double)beta * (double)beta * pdp + 2.0 * (double)beta * (double)sigma * pdz + (double)sigma * (double)sigma * zdz;
    rz = rz_new;
    if (fabs((double)rz_new) < P.tol) break;
  }

  // ---- the loop has ended: publish, then the trial step from the registers (apply_on_exit's work) ---------------------------
  if (b == 0 && tid == 0) {
    st.iters[0] = iters;
    *st.hiters = iters;
    if (st.left) *st.left = 1;
    if (st.ts) st.ts[1] = wall_clock64();
    __threadfence_system();
  }
  const ApplyOnExit<T> &ap = P.ap;
  double rho = 0;
  // (loads of every unit first, as in the update)
  T xo[RP_RV], sq[RP_RV], bq[RP_RV];
#pragma unroll
  for (int q = 0; q < RP_RV; ++q) {
    xo[q] = bq[q] = T(0); sq[q] = T(1);
    if (!((onmask >> q) & 1u)) continue;
    const unsigned t = st_[q];
    sq[q] = scales[t];
    if (ap.cams) { xo[q] = ((cammask >> q) & 1u) ? ap.cams[t] : ap.pts[t - pose_dim]; bq[q] = ap.bu[t]; }
  }
#pragma unroll
  for (int q = 0; q < RP_RV; ++q) {
    if (!((usedmask >> q) & 1u)) break;
    const bool on = (onmask >> q) & 1u, is_cam = (cammask >> q) & 1u;
    const unsigned t = st_[q];
    const T d = alpha_pend * sp[q] + sx[q], s = sq[q];
    T xn = T(0);
    if (on) {
      P.dx[t] = d;
      if (ap.cams) {
        xn = xo[q] + d * s;
        if (is_cam) {
          ap.cams_bak[t] = xo[q]; ap.cams[t] = xn;
          if (ap.cam_weight) rho += (double)(d * ((T)mu * d + s * bq[q]));
        } else {
          const unsigned qd = t - pose_dim;
          ap.pts_bak[qd] = xo[q]; ap.pts[qd] = xn;
          if (ap.xp) ap.xp[8 * (size_t)(qd / 3u) + li] = xn;
          rho += (double)(d * ((T)mu * d + s * bq[q]));
        }
      }
    }
    if (is_cam && ap.cams) { // the moved cameras' packs: entry 0's lane gathers the camera's nine values
      T cam[9], pk[PACK];
#pragma unroll
      for (int i = 0; i < 9; ++i) cam[i] = __shfl(xn, (int)lbase9 + i, 64);
      if (on && ci == 0) {
        make_campack(cam, pk);
        const unsigned c = t / 9u;
#pragma unroll
        for (int i = 0; i < PACK; ++i) ap.pack[PACK * (size_t)c + i] = pk[i];
      }
    }
  }
  if (ap.cams) {
    double v4[4] = {rho, 0.0, 0.0, 0.0};
    rp_block_allsum4(v4, l_red);
    if (tid == 0) ap.rho_partial[b] = v4[0];
  }
  stamp(); // step applied
  if (P.dbg && tid == 0) P.dbg[(size_t)b * 64 + 63] = n_stamp;
}

} // namespace gr
