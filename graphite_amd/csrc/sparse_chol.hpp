// Sparse direct solve of the reduced camera system S x = b_S: nested-dissection ordering + supernodal (tile) Cholesky
// on MFMA, scheduled by elimination-tree LEVEL.
//
// Role in the reference: cudssSchurSolver / EigenSchurLDLTSolver when S is sparse (solver/cudss_schur.hpp:146-234,
// solver/eigen_schur.hpp:71-108 with Eigen's AMD ordering, src/eigen_solver.cpp:10-29) — SURVEY §8(f)-4.
// Why: with the natural camera order a banded S (Ladybug-1723: 122 tile columns, band of ~6 tiles) is ONE dependency
// chain of 122 x (factorise -> solve -> update) launches, 18.6 ms per LM iteration on a chip that is idle but for a
// dozen workgroups.  Here
//   1. the camera co-observation graph is ordered by nested dissection (recursive bisection through the middle
//      level of a breadth-first level structure, leaves of <= LEAF cameras); every tree node is a supernode whose
//      columns are padded to whole 128-column tiles, so independent subtrees never share a tile;
//   2. a tile-level symbolic factorisation gives the fill and the elimination tree of the tile columns; columns of
//      equal tree LEVEL are independent;
//   3. the numeric phase runs level by level, three batched launches per level: every diagonal tile of the level
//      (k_sp_potrf), every sub-diagonal tile (k_sp_gemm<0>), every target tile that a panel of the level updates,
//      ONE workgroup per target walking its list of source panels (k_sp_gemm<1>: no two workgroups write one tile,
//      no atomics); forward / backward substitution are one launch per level each.
// The chain length drops from the number of tile columns to the height of the tile elimination tree (Ladybug-1723:
// 122 -> ~35), and a level keeps tens to hundreds of workgroups busy instead of a dozen.  A graph that does not
// dissect (Venice-like: every camera pair co-observes) comes out as ONE supernode = the dense factorisation; the
// engine then keeps using DenseChol (chol.hpp), whose paired super-panels are tuned for that case.
// Storage is TILE-SPARSE (round 4): only the structurally non-zero lower tiles of the factor (fill included) exist, each a
// contiguous row-major 128 x 128 image (ld = 128) at A + slot * 128 * 128; bytes() = nz_tiles * 128^2 * sizeof(T), i.e.
// memory follows nnz(L) as cuDSS / SimplicialLDLT do (solver/cudss_schur.hpp:146-219, solver/eigen_schur.hpp:71-108) — a
// 13 000-camera banded S is ~1 GB instead of a 121 GB dense array.  Every launch list carries tile SLOTS, not (row, column)
// coordinates; the scatter of S finds its tile through a dense nt x nt slot map (4 bytes per tile position).
// The kernels reuse chol.hpp's diagonal-block factorisation and its MFMA tile loop.
#pragma once
#include "chol.hpp"
#include <map>
#include <numeric>

namespace gr {

constexpr size_t SP_TT = (size_t)CH_NB * CH_NB; // scalars per tile
constexpr int SP_PT = 512; // threads of the stand-alone diagonal-tile factorisation: eight waves share the trailing updates / inverse rows
                           // beside wave 0's chain (tools/potrf_bench.hip: others' share 28 -> 21 us; the chain itself is 26 us either way)
template <typename T>
__global__ __launch_bounds__(SP_PT) void k_sp_potrf(T *__restrict__ A, const int *__restrict__ panels, const int *__restrict__ dslot, T *__restrict__ Linv, int *__restrict__ fail) {
  extern __shared__ __align__(16) unsigned char ch_smem[];
  const int k = panels[blockIdx.x];
  chol_potrf_block<T, SP_PT>(reinterpret_cast<T *>(ch_smem), A + (size_t)dslot[k] * SP_TT, CH_NB, Linv + (size_t)k * CH_NB * CH_NB, fail, true, 0);
}

// MODE 0: L_ik = A_ik Linv_k^T for tiles[2b] = slot(i, k), tiles[2b+1] = k                       (panel solve, in place)
// MODE 1: A_ij -= sum_{k in list(b)} L_ik L_jk^T, tiles[2b] = slot(i, j) [| FUSE bit], tiles[2b+1] = i,
//         list(b) = klist[2 q], klist[2 q + 1] = slot(i, k), slot(j, k) for q in kptr[b] .. kptr[b+1])
// Same 128x128x16 MFMA pipeline as k_chol_gemm; the K loop of MODE 1 runs over the concatenated source panels.
// MODE 1 targets flagged (i | FUSE_BIT, i) are diagonal tiles whose LAST update this is (their column sits on the
// next tree level): the workgroup factorises the tile on the spot from its accumulators (as k_chol_gemm<FUSE> does),
// Linv_out = the per-panel inverse array, so the next level needs no factorisation launch of its own and the
// 75 us diagonal-block factorisation overlaps with the other updates of this launch.
constexpr int SP_FUSE_BIT = 1 << 30;
// One tile product on the 128x128x16 MFMA pipeline of k_chol_gemm.
//   MODE 0: C = C Linv^T in place (Q0 = Linv_k)                                        [kept for A/B: the level kernels use k_sp_trsm_rows]
//   MODE 1: C -= sum over the nk (slotP, slotQ) pairs of kl of P Q^T; `fuse`: C is a diagonal tile whose LAST update this is —
//           its lower triangle goes from the accumulators straight into the LDS image and is factorised on the spot
//           (chol_potrf_block), Linv_out = where its inverse goes; nothing is stored to C then.
template <typename T, int MODE>
__device__ __forceinline__ void sp_tile_product(T *__restrict__ sm, T *__restrict__ A, T *__restrict__ Cg, const int *__restrict__ kl, int nk,
                                                const T *__restrict__ Q0, bool fuse, T *__restrict__ Linv_out, int *__restrict__ fail) {
  using M = MfmaTile<T>;
  typedef typename M::acc_t acc_t;
  constexpr int KC = CH_KC, CPP = CH_NB / KC; // chunks per panel
  constexpr int ld = CH_NB;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
  const int nch = MODE == 1 ? CPP * nk : CPP;
  // chunk c of the K loop: rows of P / Q start at ptile(c) / qtile(c), its KC columns at kcol(c)
  auto ptile = [&](int c) -> const T * { return MODE == 1 ? A + (size_t)kl[2 * (c / CPP)] * SP_TT : Cg; };
  auto qtile = [&](int c) -> const T * { return MODE == 1 ? A + (size_t)kl[2 * (c / CPP) + 1] * SP_TT : Q0; };
  auto kcol = [&](int c) { return (c % CPP) * KC; };

  acc_t acc[4][4];
  const int ccol = lane & 15;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (MODE == 1) acc[mi][ni][r] = Cg[(size_t)(wr * 64 + mi * 16 + M::row(lane, r)) * ld + wc * 64 + ni * 16 + ccol];
        else acc[mi][ni][r] = T(0);
      }
  constexpr int TPR = KC / 8, RPP = 256 / TPR, NPASS = CH_NB / RPP;
  const int lr = t / TPR, lk = (t % TPR) * 8;
  T pp[NPASS][8], pq[NPASS][8];
  {
    const T *P = ptile(0) + lk, *Q = qtile(0) + lk;
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
      load8<T>(P + (size_t)(lr + u * RPP) * ld, pp[u]);
      load8<T>(Q + (size_t)(lr + u * RPP) * ld, pq[u]);
    }
  }
  constexpr int BUF = 2 * KC * CH_LDP;
  {
    T *Ps = sm, *Qs = sm + KC * CH_LDP;
#pragma unroll
    for (int u = 0; u < NPASS; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) { Ps[(lk + e) * CH_LDP + lr + u * RPP] = MODE == 1 ? -pp[u][e] : pp[u][e]; Qs[(lk + e) * CH_LDP + lr + u * RPP] = pq[u][e]; }
  }
  __syncthreads();
  if (MODE == 1) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) asm volatile("" : "+v"(acc[mi][ni]));
  }
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) {
      const T *P = ptile(c + 1) + kcol(c + 1) + lk, *Q = qtile(c + 1) + kcol(c + 1) + lk;
#pragma unroll
      for (int u = 0; u < NPASS; ++u) {
        load8<T>(P + (size_t)(lr + u * RPP) * ld, pp[u]);
        load8<T>(Q + (size_t)(lr + u * RPP) * ld, pq[u]);
      }
    }
    const T *Ps = sm + (c & 1) * BUF, *Qs = Ps + KC * CH_LDP;
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const int krow = (kk * 4 + (lane >> 4)) * CH_LDP + ccol;
      T a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = Ps[krow + wr * 64 + i * 16]; b[i] = Qs[krow + wc * 64 + i * 16]; }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = M::mma(a[mi], b[ni], acc[mi][ni]);
    }
    if (c + 1 < nch) {
      T *Pn = sm + ((c + 1) & 1) * BUF, *Qn = Pn + KC * CH_LDP;
#pragma unroll
      for (int u = 0; u < NPASS; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) { Pn[(lk + e) * CH_LDP + lr + u * RPP] = MODE == 1 ? -pp[u][e] : pp[u][e]; Qn[(lk + e) * CH_LDP + lr + u * RPP] = pq[u][e]; }
    }
    __syncthreads();
  }
  if (MODE == 1 && fuse) { // block-uniform
    T *L = sm; // the operand buffers are free: every wave passed the barrier that ends the K loop
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wr * 64 + mi * 16 + M::row(lane, r), col = wc * 64 + ni * 16 + ccol;
          L[row * CH_LP + col] = col <= row ? acc[mi][ni][r] : T(0);
        }
    chol_potrf_block<T>(L, Cg, ld, Linv_out, fail, false, 0);
    return;
  }
  int ld2 = ld;
  asm volatile("" : "+v"(ld2));
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cg[(size_t)(wr * 64 + mi * 16 + M::row(lane, r)) * ld2 + wc * 64 + ni * 16 + ccol] = acc[mi][ni][r];
}
// QUADRANT form of the update of a diagonal tile whose LAST update this is (round 5).  The upper levels of the tree
// are one tile column each: their update launch is as long as the ONE workgroup that accumulates the next diagonal tile (K = 128:
// 14 us of MFMA on one CU at the peak, 24 us measured with its loads) and then factorises it (34 us).  Here that tile's lower three
// 64 x 64 quadrants go to three workgroups — two helpers that update quadrants (1, 0) and (1, 1) in place (written through) and arrive
// at the tile's counter, and the factorising workgroup, which takes quadrant (0, 0) straight into its LDS image, waits for the counter,
// fetches the other two quadrants and factorises.  Each workgroup: 2 x 2 waves of 32 x 32 outputs, K in chunks of 16 through LDS.
// The helpers precede their factorising workgroup in the launch (lower block index: dispatched no later), and wait for nothing.
constexpr int SP_QUAD_BIT = 1 << 29;   // tiles[2 b] flag: a quadrant entry; with SP_FUSE_BIT: quadrant (0, 0) + the factorisation
constexpr int SP_QUAD_SHIFT = 27;      // bits 27-28: the quadrant, 2 * (row half) + (column half)
constexpr int SP_QHELP_BIT = 1 << 26;  // the quadrant is written through and its workgroup arrives at the tile's counter (helper of a factorising workgroup)
constexpr int SP_SLOT_MASK = SP_QHELP_BIT - 1;
constexpr int SP_QP = 64 + 4;          // LDS pitch of the [k][row] quadrant operand images
constexpr size_t sp_quad_lds(size_t w) { return (size_t)2 * 2 * CH_KC * SP_QP * w; }
// acc (out): C[qi * 64 + wr * 32 + mi * 16 + row(lane, r)][qj * 64 + wc * 32 + ni * 16 + (lane & 15)] after C -= sum_k P_k Q_k^T
template <typename T>
__device__ __forceinline__ void sp_quad_product(T *__restrict__ sm, const T *__restrict__ A, const T *__restrict__ Cg, const int *__restrict__ kl, int nk, int qi, int qj,
                                                typename MfmaTile<T>::acc_t (&acc)[2][2]) {
  using M = MfmaTile<T>;
  constexpr int KC = CH_KC, CPP = CH_NB / KC, ld = CH_NB;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1, ccol = lane & 15;
  const int nch = CPP * nk;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = Cg[(size_t)(qi * 64 + wr * 32 + mi * 16 + M::row(lane, r)) * ld + qj * 64 + wc * 32 + ni * 16 + ccol];
  // loaders: threads 0..127 fetch P (64 rows x 16 k: row t / 2, eight k's), threads 128..255 fetch Q
  const bool isq = t >= 128;
  const int lr = (t & 127) >> 1, lk = (t & 1) * 8;
  auto src = [&](int c) -> const T * {
    const int slot = kl[2 * (c / CPP) + (isq ? 1 : 0)];
    return A + (size_t)slot * SP_TT + (size_t)((isq ? qj : qi) * 64 + lr) * ld + (c % CPP) * KC + lk;
  };
  T v[8];
  auto stage = [&](int buf) {
    T *dst = sm + (size_t)buf * 2 * KC * SP_QP + (isq ? KC * SP_QP : 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[(lk + e) * SP_QP + lr] = isq ? v[e] : -v[e];
  };
  load8<T>(src(0), v);
  stage(0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) load8<T>(src(c + 1), v);
    const T *Ps = sm + (size_t)(c & 1) * 2 * KC * SP_QP, *Qs = Ps + KC * SP_QP;
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const int krow = (kk * 4 + (lane >> 4)) * SP_QP + ccol;
      T a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { a[i] = Ps[krow + wr * 32 + i * 16]; b[i] = Qs[krow + wc * 32 + i * 16]; }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = M::mma(a[mi], b[ni], acc[mi][ni]);
    }
    if (c + 1 < nch) stage((c + 1) & 1);
    __syncthreads();
  }
}
// one quadrant entry of an update launch (see above).  qcnt: one arrival counter per tile column, zero between uses.
template <typename T>
__device__ __forceinline__ void sp_quad_entry(T *__restrict__ sm, T *__restrict__ A, int cs_raw, int tj, const int *__restrict__ kl, int nk,
                                              T *__restrict__ Linv, unsigned *__restrict__ qcnt, int *__restrict__ fail) {
  using M = MfmaTile<T>;
  typename M::acc_t acc[2][2];
  T *Cg = A + (size_t)(cs_raw & SP_SLOT_MASK) * SP_TT;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1, ccol = lane & 15;
  const bool fuse = (cs_raw & SP_FUSE_BIT) != 0, help = (cs_raw & SP_QHELP_BIT) != 0;
  const int q = (cs_raw >> SP_QUAD_SHIFT) & 3, qi = q >> 1, qj = q & 1;
  sp_quad_product<T>(sm, A, Cg, kl, nk, qi, qj, acc);
  if (!fuse && !help) { // a quadrant of an ordinary target: plain stores (the launch boundary publishes them)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cg[(size_t)(qi * 64 + wr * 32 + mi * 16 + M::row(lane, r)) * CH_NB + qj * 64 + wc * 32 + ni * 16 + ccol] = acc[mi][ni][r];
    return;
  }
  if (!fuse) { // helper: the quadrant in place, written through; then the arrival
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __hip_atomic_store(&Cg[(size_t)(qi * 64 + wr * 32 + mi * 16 + M::row(lane, r)) * CH_NB + qj * 64 + wc * 32 + ni * 16 + ccol], acc[mi][ni][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_add(&qcnt[tj], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // factorising workgroup: quadrant (0, 0) from the accumulators, (0, 1) zero, (1, 0) and (1, 1) from the helpers
  T *L = sm; // (every wave is past the barrier that ends the K loop: the operand images are free)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * 32 + mi * 16 + M::row(lane, r), col = wc * 32 + ni * 16 + ccol;
        L[row * CH_LP + col] = col <= row ? acc[mi][ni][r] : T(0);
      }
  for (int e = t; e < 64 * 64; e += 256) L[(e >> 6) * CH_LP + 64 + (e & 63)] = T(0);
  if (t == 0) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(&qcnt[tj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 200000000ll) { *fail = 1; break; } // 2 s: the helpers never ran (reported as a failed factorisation)
    }
    __hip_atomic_store(&qcnt[tj], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
#pragma unroll 1
  for (int base = 0; base < 64 * 128; base += 256 * 8) { // eight requests of a thread in flight (one at a time: 32 round trips to L2)
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = base + u * 256 + t, row = 64 + (e >> 7), col = e & 127;
      v[u] = col <= row ? __hip_atomic_load(&Cg[(size_t)row * CH_NB + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T(0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = base + u * 256 + t, row = 64 + (e >> 7), col = e & 127;
      L[row * CH_LP + col] = v[u];
    }
  }
  chol_potrf_block<T>(L, Cg, CH_NB, Linv + (size_t)tj * SP_TT, fail, false, 0);
}
// the level-scheduled launches:
// MODE 0: L_ik = A_ik Linv_k^T for tiles[2b] = slot(i, k), tiles[2b+1] = k                       (panel solve, in place)
// MODE 1: A_ij -= sum_{k in list(b)} L_ik L_jk^T, tiles[2b] = slot(i, j) [| FUSE bit], tiles[2b+1] = i,
//         list(b) = klist[2 q], klist[2 q + 1] = slot(i, k), slot(j, k) for q in kptr[b] .. kptr[b+1])
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_sp_gemm(T *__restrict__ A, const int *__restrict__ tiles, const int *__restrict__ kptr, const int *__restrict__ klist,
                                                 const T *__restrict__ Linv, T *__restrict__ Linv_out = nullptr, int *__restrict__ fail = nullptr, unsigned *__restrict__ qcnt = nullptr) {
  extern __shared__ __align__(16) unsigned char ch_smem[];
  const int cs_raw = tiles[2 * blockIdx.x], tj = tiles[2 * blockIdx.x + 1]; // MODE 0: tj = panel k; MODE 1: tj = tile row / column of a diagonal target
  if (MODE == 1 && (cs_raw & SP_QUAD_BIT)) {
    sp_quad_entry<T>(reinterpret_cast<T *>(ch_smem), A, cs_raw, tj, klist + 2 * (size_t)kptr[blockIdx.x], kptr[blockIdx.x + 1] - kptr[blockIdx.x], Linv_out, qcnt, fail);
    return;
  }
  const bool fuse = MODE == 1 && (cs_raw & SP_FUSE_BIT) != 0;
  sp_tile_product<T, MODE>(reinterpret_cast<T *>(ch_smem), A, A + (size_t)(cs_raw & ~SP_FUSE_BIT) * SP_TT,
                           MODE == 1 ? klist + 2 * (size_t)kptr[blockIdx.x] : nullptr, MODE == 1 ? kptr[blockIdx.x + 1] - kptr[blockIdx.x] : 0,
                           MODE == 0 ? Linv + (size_t)tj * SP_TT : nullptr, fuse, fuse ? Linv_out + (size_t)tj * SP_TT : nullptr, fail);
}

// (Round 4, measured and removed: the update launch with 512-thread workgroups — eight waves of 32 x 64 outputs, the sub-panel /
// trailing / inverse phases of the fused factorisation on twice the waves.  In fp64 the kernel is then limited to 256 registers per
// lane (two waves per SIMD) and spills 291 VGPRs: 94 -> 248 us per launch.  The 256-thread form keeps its 256 VGPRs + 240 AGPRs.)

// Panel solve, ROW-SPLIT: L_ik = A_ik Linv_k^T with one workgroup per SP_SLAB-row slab of the tile (tiles[2 t] = slot(i, k),
// tiles[2 t + 1] = k for tile t = b / slabs-per-tile, slab b % slabs-per-tile), in place: a slab only reads its own rows of A_ik.  The upper levels of the tree hold a
// handful of tiles, and one workgroup per 128 x 128 x 128 product is 26 us of a chain that runs 21 times per factorisation
// (14 us of MFMA on ONE CU + launch + cold loads); four slabs on four CUs divide the MFMA work and read Linv_k (128 KB) once
// each from L2.  Wave w computes columns [32 w, 32 w + 32) of the slab: 2 x 2 MFMA tiles, K in chunks of 16 through LDS.
constexpr int SP_SLAB = 32; // rows per slab (panel-solve launch on Ladybug-1723: 25.9 us unsplit, 15.4 us with 32-row slabs, 15.6 us with 16-row slabs;
                            // K chunks of 32 with the chunks above a wave's columns skipped — X is lower triangular —: 15.0 -> 17.0 us; all operands requested up
                            // front, the chunk loop fed from registers: 15.4 us — the launch is not waiting for its loads)
constexpr size_t sp_trsm_lds(size_t w) { return (size_t)2 * CH_KC * (SP_SLAB + 4 + CH_LDP) * w; }
// Cg: the slab's first row inside its tile; Qg: Linv_k; sm: sp_trsm_lds bytes of LDS
template <typename T>
__device__ __forceinline__ void sp_trsm_slab(T *__restrict__ sm, T *__restrict__ Cg, const T *__restrict__ Qg) {
  using M = MfmaTile<T>;
  typedef typename M::acc_t acc_t;
  constexpr int KC = CH_KC, NCH = CH_NB / KC, PP = SP_SLAB + 4, MI = SP_SLAB / 16; // LDS pitch of the [k][row] slab image; MFMA row tiles
  constexpr int PE = SP_SLAB * KC / 256;                                            // slab scalars per thread and chunk (2 or 1)
  static_assert(SP_SLAB == 16 || SP_SLAB == 32, "slab of 16 or 32 rows");
  T (*Ps)[KC * PP] = reinterpret_cast<T (*)[KC * PP]>(sm);
  T (*Qs)[KC * CH_LDP] = reinterpret_cast<T (*)[KC * CH_LDP]>(sm + 2 * KC * PP);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, ccol = lane & 15;
  // loaders: Q chunk = 128 rows x 16 k: thread -> (row t / 2, 8 k's); P chunk = SLAB rows x 16 k: thread -> (row, PE consecutive k's)
  const int qr = t >> 1, qk = (t & 1) * 8, pr = t / (KC / PE), pk = (t % (KC / PE)) * PE;
  T q8[8], p2[PE];
  auto fetch = [&](int c) {
    load8<T>(Qg + (size_t)qr * CH_NB + c * KC + qk, q8);
#pragma unroll
    for (int e = 0; e < PE; ++e) p2[e] = Cg[(size_t)pr * CH_NB + c * KC + pk + e];
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 8; ++e) Qs[buf][(qk + e) * CH_LDP + qr] = q8[e];
#pragma unroll
    for (int e = 0; e < PE; ++e) Ps[buf][(pk + e) * PP + pr] = p2[e];
  };
  acc_t acc[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = acc_t{T(0), T(0), T(0), T(0)};
  fetch(0);
  stage(0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) fetch(c + 1);
    const T *P = Ps[c & 1], *Q = Qs[c & 1];
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const int k = kk * 4 + (lane >> 4);
      T a[MI], b[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = P[k * PP + i * 16 + ccol];
#pragma unroll
      for (int i = 0; i < 2; ++i) b[i] = Q[k * CH_LDP + wave * 32 + i * 16 + ccol];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = M::mma(a[mi], b[ni], acc[mi][ni]);
    }
    if (c + 1 < NCH) stage((c + 1) & 1);
    __syncthreads();
  }
  // every read of the slab's rows happened before the last barrier: the in-place store is safe
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cg[(size_t)(mi * 16 + M::row(lane, r)) * CH_NB + wave * 32 + ni * 16 + ccol] = acc[mi][ni][r];
}
template <typename T>
__global__ __launch_bounds__(256) void k_sp_trsm_rows(T *__restrict__ A, const int *__restrict__ tiles, const T *__restrict__ Linv) {
  extern __shared__ __align__(16) unsigned char ch_smem[];
  constexpr int SPT = CH_NB / SP_SLAB; // slabs per tile
  const int tile = blockIdx.x / SPT, slab = blockIdx.x % SPT;
  sp_trsm_slab<T>(reinterpret_cast<T *>(ch_smem), A + (size_t)tiles[2 * tile] * SP_TT + (size_t)slab * SP_SLAB * CH_NB, Linv + (size_t)tiles[2 * tile + 1] * SP_TT);
}

// (Round 5, measured and removed: the level's panel-solve slabs as the FIRST workgroups of its update launch — written through, a counter
// per tile, update workgroups waiting for their source tiles' four slabs: upper-level launch 61.3 us against 8.8 + 50.4 us and one launch
// boundary; 371 vs 375 LM it/s.  The slab's write-through stores, the poll and the cold first read of the sources cost what the launch did.)
// (Round 4, measured and removed: the whole factorisation as ONE dependency-driven launch — every panel-solve slab, target update
// and diagonal factorisation an item of a level-ordered queue, persistent workgroups waiting on per-tile counters with agent-scope
// release / acquire fences.  2.51 ms per factorisation on Ladybug-1723 against 2.45 ms for the level launches: the launch floors
// and event bubbles it removes (~0.45 ms) come back as fence + poll latency on the same chain, whose length is set by the compute
// of its links — panel solve 10 us, diagonal update + 128 x 128 factorisation 67 us, 21 times.)

// permuted scatter of the upper bs x bs blocks of S (column-major blocks, block (i <= j); bs = 9 for the reduced camera system, any
// block size for gr_spchol) into the lower triangle; camcol[c] = first (padded, permuted) column of node c
template <typename T>
__global__ __launch_bounds__(256) void k_sp_scatter(int64_t nnzb, const int *__restrict__ rowi, const int *__restrict__ coli, const int *__restrict__ camcol,
                                                    const T *__restrict__ S, T *__restrict__ A, const int *__restrict__ tmap, int nt, int bs) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int bb = bs * bs;
  if (e >= (int64_t)bb * nnzb) return;
  const int64_t q = e / bb;
  const int w = (int)(e - (int64_t)bb * q), c = w / bs, r = w - bs * c; // S_q(r, c) = S(bs i + r, bs j + c), i <= j
  int R = camcol[rowi[q]] + r, C = camcol[coli[q]] + c;
  if (rowi[q] == coli[q]) { if (R < C) return; } // diagonal block: its lower half
  else if (R < C) { const int x = R; R = C; C = x; }
  A[(size_t)tmap[(size_t)(R >> 7) * nt + (C >> 7)] * SP_TT + (size_t)(R & 127) * CH_NB + (C & 127)] = S[e];
}
// zero the structurally non-zero lower tiles; padding columns get a unit diagonal
template <typename T>
__global__ __launch_bounds__(256) void k_sp_clear(T *__restrict__ A, const unsigned char *__restrict__ is_pad, const int *__restrict__ tiles) {
  const int ti = tiles[2 * blockIdx.x], tj = tiles[2 * blockIdx.x + 1]; // slot = blockIdx.x: the tiles are stored in this list's order
  T *Cg = A + (size_t)blockIdx.x * SP_TT;
  for (int e = threadIdx.x; e < CH_NB * CH_NB; e += 256) {
    const int r = e >> 7, c = e & 127;
    Cg[e] = (ti == tj && r == c && is_pad[ti * CH_NB + r]) ? T(1) : T(0);
  }
}
template <typename T> __global__ void k_sp_rhs(int npad, const int *__restrict__ src, const T *__restrict__ b, T *__restrict__ vb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) vb[i] = src[i] >= 0 ? b[src[i]] : T(0);
}
template <typename T> __global__ void k_sp_unpermute(int npad, const int *__restrict__ src, const T *__restrict__ vx, T *__restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad && src[i] >= 0) x[src[i]] = vx[i];
}
// Substitution kernels.  A level's panels are cut into work ITEMS (panel k, a slice of <= slice tiles of its row
// list / column list), one workgroup per item: the long rows of the separator columns near the root (a hundred tiles)
// are spread over many workgroups instead of one.  An item leaves its 128 partial sums in `partial`; the item that
// arrives LAST at the panel's ticket adds them in item order (fixed order: reproducible) and finishes the panel.
// Visibility across XCDs: write-through (agent-scope relaxed atomic) stores, drained, then a relaxed ticket; the last
// arriver reads with agent-scope loads (cdna_hip_programming.md G16, form R1) — the same hand-off as grid_sum2.
// tiles per substitution item: SparseChol::slice (gr_bal_tuning.spchol_slice; 6 -> 2: 249 -> 260 LM it/s on Ladybug-1723; 2 -> 1: 319.5 -> 322.5)
struct SpItems { const int *panel, *beg, *end, *first, *count; int base; }; // per item: panel, slice, first item / item count of its panel (absolute item ids); base = id of this launch's item 0
template <typename T> __device__ __forceinline__ bool sp_last_arriver(T val, bool writer, T *__restrict__ partial, int item, int idx, unsigned *__restrict__ ticket, int panel, int count) {
  __shared__ bool s_last;
  if (writer) __hip_atomic_store(&partial[(size_t)item * CH_NB + idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned tk = __hip_atomic_fetch_add(&ticket[panel], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = tk == (unsigned)count - 1;
    if (s_last) __hip_atomic_store(&ticket[panel], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return s_last;
}
// forward: y_k = Linv_k (b_k - sum_{j in row(k)} L_kj y_j)
template <typename T>
__device__ __forceinline__ void sp_fwd_item(T *__restrict__ bk /* LDS, CH_NB scalars */, const int item, const T *A, const T *Linv, const SpItems &it, const int *__restrict__ rcols,
                                            const int *__restrict__ rslot, const T *__restrict__ b, T *y, T *__restrict__ partial, unsigned *__restrict__ ticket) {
  const int k = it.panel[item], item_id = it.base + item;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  T tot[2] = {T(0), T(0)};
  for (int e = it.beg[item]; e < it.end[item]; ++e) {
    const int j = rcols[e];
    const T y0 = y[j * CH_NB + lane], y1 = y[j * CH_NB + 64 + lane];
#pragma unroll
    for (int h = 0; h < 2; ++h) tot[h] += rows16_dot<T>(A + (size_t)rslot[e] * SP_TT + (size_t)(wave * 32 + h * 16) * CH_NB, (size_t)CH_NB, y0, y1, lane);
  }
  // lane L of (wave, h) holds the partial of row wave*32 + h*16 + (L >> 2)
  bool last = true;
  if (it.count[item] > 1) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if ((lane & 3) == 0) __hip_atomic_store(&partial[(size_t)item_id * CH_NB + wave * 32 + h * 16 + (lane >> 2)], tot[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = sp_last_arriver<T>(T(0), false, partial, item_id, 0, ticket, k, it.count[item]);
    if (!last) return;
    if (t < CH_NB) {
      T s = T(0);
      for (int q = it.first[item]; q < it.first[item] + it.count[item]; ++q) s += __hip_atomic_load(&partial[(size_t)q * CH_NB + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bk[t] = b[k * CH_NB + t] - s;
    }
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if ((lane & 3) == 0) bk[wave * 32 + h * 16 + (lane >> 2)] = b[k * CH_NB + wave * 32 + h * 16 + (lane >> 2)] - tot[h];
  }
  __syncthreads();
  const T b0 = bk[lane], b1 = bk[64 + lane];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r0 = wave * 32 + h * 16;
    const T s = rows16_dot<T>(Linv + (size_t)k * CH_NB * CH_NB + r0 * CH_NB, CH_NB, b0, b1, lane);
    if ((lane & 3) == 0) y[k * CH_NB + r0 + (lane >> 2)] = s;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_sp_fwd(const T *__restrict__ A, const T *__restrict__ Linv, SpItems it, const int *__restrict__ rcols, const int *__restrict__ rslot,
                                                const T *__restrict__ b, T *__restrict__ y, T *__restrict__ partial, unsigned *__restrict__ ticket) {
  __shared__ T bk[CH_NB];
  sp_fwd_item<T>(bk, (int)blockIdx.x, A, Linv, it, rcols, rslot, b, y, partial, ticket);
}
// Level l's update launch WITH level l's forward-substitution items as its last workgroups (both need the panels of level l and
// nothing of each other: the update writes tiles of later columns, the substitution reads tiles of earlier ones).  The substitution
// used to ride on a second stream behind an event per level: each event record was a bubble on the factorisation's stream — in the
// kernel trace 10 us between panel solve and update, 16 us between update and the next panel solve, on each of 26 levels.
template <typename T>
__global__ __launch_bounds__(256) void k_sp_update_fwd(T *A, const int *__restrict__ tiles, const int *__restrict__ kptr, const int *__restrict__ klist,
                                                       T *Linv, int *__restrict__ fail, int nup, SpItems it, const int *__restrict__ rcols,
                                                       const int *__restrict__ rslot, const T *__restrict__ b, T *y, T *__restrict__ partial, unsigned *__restrict__ ticket,
                                                       unsigned *__restrict__ qcnt) {
  extern __shared__ __align__(16) unsigned char ch_smem[];
  if ((int)blockIdx.x >= nup) {
    sp_fwd_item<T>(reinterpret_cast<T *>(ch_smem), (int)blockIdx.x - nup, A, Linv, it, rcols, rslot, b, y, partial, ticket);
    return;
  }
  const int cs_raw = tiles[2 * blockIdx.x], tj = tiles[2 * blockIdx.x + 1];
  if (cs_raw & SP_QUAD_BIT) {
    sp_quad_entry<T>(reinterpret_cast<T *>(ch_smem), A, cs_raw, tj, klist + 2 * (size_t)kptr[blockIdx.x], kptr[blockIdx.x + 1] - kptr[blockIdx.x], Linv, qcnt, fail);
    return;
  }
  const bool fuse = (cs_raw & SP_FUSE_BIT) != 0;
  sp_tile_product<T, 1>(reinterpret_cast<T *>(ch_smem), A, A + (size_t)(cs_raw & ~SP_FUSE_BIT) * SP_TT, klist + 2 * (size_t)kptr[blockIdx.x],
                        kptr[blockIdx.x + 1] - kptr[blockIdx.x], nullptr, fuse, fuse ? Linv + (size_t)tj * SP_TT : nullptr, fail);
}
// backward: x_k = Linv_k^T (y_k - sum_{i in col(k)} L_ik^T x_i)
// (Measured and removed: ONE launch for all levels with an in-kernel grid barrier between them, x exchanged through
// agent-scope atomics — 246-250 LM it/s either way on Ladybug-1723: a level's 20 us are the cold reads of its L tiles by a
// few workgroups, not the launch boundary.  What did help is more, smaller items per panel: slice 6 -> 2, 249 -> 260.)
template <typename T>
__global__ __launch_bounds__(256) void k_sp_bwd(const T *__restrict__ A, const T *__restrict__ Linv, SpItems it, const int *__restrict__ crows, const int *__restrict__ cslot,
                                                const T *__restrict__ y, T *__restrict__ x, T *__restrict__ partial, unsigned *__restrict__ ticket) {
  __shared__ T v[CH_NB];
  __shared__ T half[CH_NB];
  const int item = blockIdx.x, k = it.panel[item], item_id = it.base + item;
  const int t = threadIdx.x, c = t & 127, h = t >> 7;
  T s = T(0);
  for (int e = it.beg[item]; e < it.end[item]; ++e) {
    const int i = crows[e];
    const T *Lg = A + (size_t)cslot[e] * SP_TT + (size_t)(h * 64) * CH_NB + c;
    const T *xg = x + i * CH_NB + h * 64;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) s += Lg[(size_t)r * CH_NB] * xg[r];
  }
  if (h == 1) half[c] = s;
  __syncthreads();
  s += half[c]; // valid for h == 0
  if (it.count[item] > 1) {
    if (!sp_last_arriver<T>(s, h == 0, partial, item_id, c, ticket, k, it.count[item])) return;
    if (h == 0) {
      s = T(0);
      for (int q = it.first[item]; q < it.first[item] + it.count[item]; ++q) s += __hip_atomic_load(&partial[(size_t)q * CH_NB + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (h == 0) v[c] = y[k * CH_NB + c] - s;
  __syncthreads();
  const T *Lk = Linv + (size_t)k * CH_NB * CH_NB;
  T s2 = T(0);
#pragma unroll 8
  for (int r = h * 64; r < h * 64 + 64; ++r) s2 += Lk[r * CH_NB + c] * v[r];
  __syncthreads();
  if (h == 1) half[c] = s2;
  __syncthreads();
  if (h == 0) x[k * CH_NB + c] = s2 + half[c];
}

// Backward substitution as ONE dependency-driven launch (round 5).  The level-by-level form is 21 launches of 10-12 us on Ladybug-1723 —
// each a chain launch floor -> cold read of the level's L tiles -> partial sums -> ticket -> read of Linv_k -> x_k — for a handful of
// workgroups per level: 240 us of a 2.7 ms LM iteration.  Here every work item is a workgroup of one launch, in the level order the
// launches had (top of the tree first: a workgroup only ever waits for workgroups with a LOWER block index, which are resident or done):
//   item (panel k, tile (i, k)) : fetches its 128 x 128 tile into REGISTERS at once (64 scalars per thread), only then waits for x_i
//                                 (ready[i] == seq), forms L_ik^T x_i, leaves its 128 partial sums (written through) and arrives at cnt[k];
//   finisher (panel k)          : fetches Linv_k into registers, waits for the panel's items, adds their partial sums in item order,
//                                 x_k = Linv_k^T (y_k - sum), written through, then ready[k] = seq.
// What crosses workgroups is written through and drained before the arrival / the ready word and read back with agent-scope loads
// (cdna_hip_programming.md G16, form R1).  The tiles of the factor are final before the launch: plain loads.  Same products, same order of
// additions as k_sp_bwd: same bits.
struct SpChain { const int *kind_panel, *row, *slot, *idx; const int *pfirst, *pcount; }; // per entry: (finisher ? ~k : k), tile row i, tile slot, item index in its panel; per panel: first partial row, items
template <typename T>
__global__ __launch_bounds__(256) void k_sp_bwd_chain(const T *__restrict__ A, const T *__restrict__ Linv, SpChain ch, const T *__restrict__ y, T *x, T *partial,
                                                      unsigned *cnt, unsigned *ready, unsigned seq, int *__restrict__ fail) {
  __shared__ T xs[CH_NB];
  __shared__ T half[CH_NB];
  const int e = blockIdx.x, kp = ch.kind_panel[e];
  const bool finisher = kp < 0;
  const int k = finisher ? ~kp : kp;
  const int t = threadIdx.x, c = t & 127, h = t >> 7;
  auto wait_for = [&](unsigned *word, unsigned want, bool at_least) {
    if (t == 0) {
      const long long t0 = wall_clock64();
      for (;;) {
        const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (at_least ? v >= want : v == want) break;
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 200000000ll) { *fail = 1; break; } // 2 s: an earlier workgroup never ran (reported as a failed factorisation)
      }
    }
    __syncthreads();
  };
  T Lr[64];
  const T *Lg = (finisher ? Linv + (size_t)k * SP_TT : A + (size_t)ch.slot[e] * SP_TT) + (size_t)(h * 64) * CH_NB + c;
#pragma unroll
  for (int r = 0; r < 64; ++r) Lr[r] = Lg[(size_t)r * CH_NB];
  if (!finisher) {
    const int i = ch.row[e];
    wait_for(&ready[i], seq, false);
    if (t < CH_NB) xs[t] = __hip_atomic_load(&x[i * CH_NB + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    T s = T(0);
#pragma unroll
    for (int r = 0; r < 64; ++r) s += Lr[r] * xs[h * 64 + r];
    if (h == 1) half[c] = s;
    __syncthreads();
    if (h == 0) __hip_atomic_store(&partial[(size_t)(ch.pfirst[k] + ch.idx[e]) * CH_NB + c], s + half[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_add(&cnt[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int n = ch.pcount[k];
  if (n > 0) {
    wait_for(&cnt[k], (unsigned)n, true);
    if (t == 0) __hip_atomic_store(&cnt[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (t < CH_NB) {
    T s = T(0);
    for (int q = 0; q < n; ++q) s += __hip_atomic_load(&partial[(size_t)(ch.pfirst[k] + q) * CH_NB + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    xs[t] = y[k * CH_NB + t] - s;
  }
  __syncthreads();
  T s2 = T(0);
#pragma unroll
  for (int r = 0; r < 64; ++r) s2 += Lr[r] * xs[h * 64 + r];
  if (h == 1) half[c] = s2;
  __syncthreads();
  if (h == 0) __hip_atomic_store(&x[k * CH_NB + c], s2 + half[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) __hip_atomic_store(&ready[k], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T> struct SparseChol {
  hipStream_t stream = nullptr;
  int n = 0, npad = 0, nt = 0, nlevels = 0, nsuper = 0;
  int64_t factor_tiles = 0;       // structurally non-zero lower tiles after fill
  double flops = 0;
  DevBuf<T> A, Linv, vb, vy, vx;
  DevBuf<int> d_camcol, d_src, d_nz, d_fail, d_panels, d_trsm, d_upd, d_kptr, d_klist, d_rptr, d_rcols, d_cptr, d_crows;
  DevBuf<int> d_tmap, d_dslot, d_rslot, d_cslot; // tile-sparse storage: (i, j) -> slot map, diagonal slots, slots beside d_rcols / d_crows
  DevBuf<unsigned char> d_pad;
  DevBuf<int> d_itf[5], d_itb[5];
  DevBuf<T> partial;
  DevBuf<unsigned> ticket;
  std::vector<int> lvl_fitem_off, lvl_bitem_off;
  SpItems items(DevBuf<int> (&d)[5], int off) const { return SpItems{d[0].p + off, d[1].p + off, d[2].p + off, d[3].p + off, d[4].p + off, off}; }
  std::vector<int> lvl_panel_off, lvl_trsm_off, lvl_upd_off; // per level offsets into d_panels / d_trsm (pairs) / d_upd (pairs)
  std::vector<int> h_kptr;
  int nz_tiles = 0;
  int *h_fail = nullptr;
  CholProfSink *sink = nullptr;
  bool attrs_set = false;
  static constexpr int LEAF = 56; // cameras per leaf supernode: 504 columns = 4 tiles (8 padding columns); other block sizes: 504 / bs nodes
  int bs = 9;                     // scalar columns per node (9: cameras of the reduced system)
  bool fuse_potrf = true; // gr_bal_tuning.spchol_fuse: the next level's diagonal tiles factorised inside this level's update launch
  bool fuse_quads = true; // ... and that tile's update spread over three workgroups by quadrant (spchol_fuse = 2; 1: one workgroup)
  int quad_max_targets = 1 << 30; // levels with more update targets would keep the one-workgroup form (measured: 64 / 128 / 256 / all -> 364 / 366 / 368 / 367 LM it/s: all)
  // backward substitution as one dependency-driven launch (k_sp_bwd_chain; gr_bal_tuning.spchol_overlap = 3 keeps the level launches)
  DevBuf<int> d_ch[6];
  DevBuf<unsigned> d_bcnt, d_ready;
  int chain_entries = 0;
  unsigned chain_seq = 0;
  bool bwd_chain = true;
  DevBuf<unsigned> d_qcnt; // [nt] arrivals of the quadrant helpers, zero between uses
  std::vector<double> lvl_upd_tiles; // tile products per level's update launch (a quadrant entry counts a quarter per source)
  int slice = 1;          // gr_bal_tuning.spchol_slice: tiles per substitution work item
  bool row_split_trsm = true; // panel solves by 32-row slabs (k_sp_trsm_rows)

  SparseChol() = default;
  SparseChol(const SparseChol &) = delete;
  ~SparseChol() {
    if (h_fail) (void)hipHostFree(h_fail);
    for (hipEvent_t e : lvl_done) (void)hipEventDestroy(e);
    if (aux_done) (void)hipEventDestroy(aux_done);
    if (rhs_ready) (void)hipEventDestroy(rhs_ready);
    if (aux) (void)hipStreamDestroy(aux);
  }

  // nested dissection of the camera graph: returns the supernodes (camera lists) in elimination order
  static std::vector<std::vector<int>> nested_dissection(int Nc, const std::vector<std::vector<int>> &adj, int leaf = LEAF) {
    std::vector<std::vector<int>> out;
    std::vector<int> mark(Nc, -1), dist(Nc, 0);
    int stamp = 0;
    // bfs inside `nodes` (mark == stamp) from `root`; returns the visit order, dist[] = level
    auto bfs = [&](int root, std::vector<int> &order) {
      order.clear();
      order.push_back(root); mark[root] = -2 - stamp; dist[root] = 0; // visited within this stamp
      for (size_t h = 0; h < order.size(); ++h) {
        const int v = order[h];
        for (int u : adj[v]) if (mark[u] == stamp) { mark[u] = -2 - stamp; dist[u] = dist[v] + 1; order.push_back(u); }
      }
    };
    struct Job { std::vector<int> nodes; bool emit_only; };
    // explicit stack: (nodes) -> split into A, B, separator; order = nd(A), nd(B), separator
    std::vector<Job> stack;
    {
      std::vector<int> all(Nc);
      std::iota(all.begin(), all.end(), 0);
      stack.push_back(Job{std::move(all), false});
    }
    while (!stack.empty()) {
      Job job = std::move(stack.back());
      stack.pop_back();
      auto &nodes = job.nodes;
      if (nodes.empty()) continue;
      if (job.emit_only || (int)nodes.size() <= leaf) { out.push_back(std::move(nodes)); continue; }
      ++stamp;
      for (int v : nodes) mark[v] = stamp;
      std::vector<int> order, order2;
      bfs(nodes[0], order);
      if (order.size() < nodes.size()) { // disconnected inside `nodes`: the reached component and the rest are independent
        std::vector<int> rest;
        for (int v : nodes) if (mark[v] == stamp) rest.push_back(v);
        stack.push_back(Job{std::move(rest), false});
        stack.push_back(Job{std::move(order), false});
        continue;
      }
      // pseudo-peripheral start: restart from the farthest node once
      const int far = order.back();
      for (int v : nodes) mark[v] = stamp;
      bfs(far, order2);
      const int depth = dist[order2.back()];
      if (depth < 2) { out.push_back(std::move(nodes)); continue; } // does not dissect: one dense supernode
      // separator = the level at which half of the nodes have been passed (kept off the two end levels)
      std::vector<int> count(depth + 1, 0);
      for (int v : order2) count[dist[v]]++;
      int s = 1, acc = count[0];
      while (s < depth - 1 && acc + count[s] < (int)nodes.size() / 2) acc += count[s++];
      std::vector<int> Apart, Bpart, Sep;
      for (int v : order2) (dist[v] < s ? Apart : dist[v] == s ? Sep : Bpart).push_back(v);
      // LIFO: pushed last = processed first; wanted output order: nd(A), nd(B), Sep
      stack.push_back(Job{std::move(Sep), true});
      stack.push_back(Job{std::move(Bpart), false});
      stack.push_back(Job{std::move(Apart), false});
    }
    // the explicit stack emits A-subtree, B-subtree, separator in that order only if the separator is emitted after
    // BOTH subtrees are fully done, which the LIFO order above guarantees (Sep sits below A and B on the stack)
    return out;
  }

  // Nc cameras, upper block list (rowi <= coli) of S.  Returns false when the graph does not dissect (one supernode).
  bool set_structure(int Nc, const std::vector<int> &rowi, const std::vector<int> &coli, hipStream_t s, int block = 9) {
    stream = s;
    bs = block;
    n = bs * Nc;
    std::vector<std::vector<int>> adj(Nc);
    for (size_t q = 0; q < rowi.size(); ++q) if (rowi[q] != coli[q]) { adj[rowi[q]].push_back(coli[q]); adj[coli[q]].push_back(rowi[q]); }
    const auto supers = nested_dissection(Nc, adj, bs == 9 ? LEAF : std::max(8, 504 / bs));
    nsuper = (int)supers.size();
    if (nsuper <= 1) return false;
    // padded permuted columns
    std::vector<int> camcol(Nc, 0), src;
    std::vector<unsigned char> pad;
    for (const auto &sn : supers) {
      for (int c : sn) { camcol[c] = (int)src.size(); for (int k = 0; k < bs; ++k) { src.push_back(bs * c + k); pad.push_back(0); } }
      while (src.size() % CH_NB) { src.push_back(-1); pad.push_back(1); }
    }
    npad = (int)src.size(); nt = npad / CH_NB;
    // tile structure + fill + elimination tree
    std::vector<char> tz((size_t)nt * nt, 0);
    auto nz = [&](int i, int j) -> char & { return tz[(size_t)i * nt + j]; };
    for (int i = 0; i < nt; ++i) nz(i, i) = 1;
    for (size_t q = 0; q < rowi.size(); ++q) {
      const int a0 = camcol[rowi[q]] / CH_NB, a1 = (camcol[rowi[q]] + bs - 1) / CH_NB, b0 = camcol[coli[q]] / CH_NB, b1 = (camcol[coli[q]] + bs - 1) / CH_NB;
      for (int a = a0; a <= a1; ++a)
        for (int b = b0; b <= b1; ++b) nz(std::max(a, b), std::min(a, b)) = 1;
    }
    std::vector<std::vector<int>> U(nt);
    std::vector<int> level(nt, 0);
    factor_tiles = 0; flops = 0;
    for (int j = 0; j < nt; ++j) {
      for (int i = j + 1; i < nt; ++i) if (nz(i, j)) U[j].push_back(i);
      for (size_t x = 0; x < U[j].size(); ++x)
        for (size_t y = 0; y <= x; ++y) nz(U[j][x], U[j][y]) = 1;
      if (!U[j].empty()) level[U[j][0]] = std::max(level[U[j][0]], level[j] + 1); // parent = first sub-diagonal tile
      factor_tiles += 1 + (int64_t)U[j].size();
      flops += (2.0 / 3.0 + 2.0 * U[j].size() + 1.0 * U[j].size() * (U[j].size() + 1)) * CH_NB * CH_NB * CH_NB; // potrf + inverse, trsm, updates
    }
    // a column must also wait for every column that UPDATES it, not only its tree children: level = 1 + max over the
    // columns k with nz(j, k) (all of them are its descendants, so this is the same number; computed explicitly)
    for (int j = 0; j < nt; ++j)
      for (int i : U[j]) level[i] = std::max(level[i], level[j] + 1);
    nlevels = 1 + *std::max_element(level.begin(), level.end());
    std::vector<std::vector<int>> by_level(nlevels);
    for (int j = 0; j < nt; ++j) by_level[level[j]].push_back(j);
    // tile slots: the structurally non-zero lower tiles in row-major order of (i, j); everything below addresses tiles by slot
    std::vector<int> h_nz, tmap((size_t)nt * nt, -1), dslot(nt, 0);
    for (int i = 0; i < nt; ++i)
      for (int j = 0; j <= i; ++j) if (nz(i, j)) { tmap[(size_t)i * nt + j] = (int)(h_nz.size() / 2); h_nz.push_back(i); h_nz.push_back(j); }
    nz_tiles = (int)(h_nz.size() / 2);
    auto slot = [&](int i, int j) { return tmap[(size_t)i * nt + j]; };
    for (int k = 0; k < nt; ++k) dslot[k] = slot(k, k);
    std::vector<int> h_panels, h_trsm, h_upd, h_klist;
    h_kptr.assign(1, 0);
    lvl_panel_off.assign(1, 0); lvl_trsm_off.assign(1, 0); lvl_upd_off.assign(1, 0); lvl_upd_tiles.clear();
    for (int l = 0; l < nlevels; ++l) {
      std::map<std::pair<int, int>, std::vector<int>> targets;
      double tiles_l = 0;
      for (int k : by_level[l]) {
        h_panels.push_back(k);
        for (int i : U[k]) { h_trsm.push_back(slot(i, k)); h_trsm.push_back(k); }
        for (size_t x = 0; x < U[k].size(); ++x)
          for (size_t y = 0; y <= x; ++y) targets[{U[k][x], U[k][y]}].push_back(k);
      }
      // the diagonal tiles of the NEXT level lead the list (their workgroups also factorise: longest, scheduled first)
      for (int pass = 0; pass < 2; ++pass)
        for (auto &tg : targets) {
          const int ti = tg.first.first, tj = tg.first.second;
          const bool diag_next = fuse_potrf && ti == tj && level[ti] == l + 1;
          if (diag_next != (pass == 0)) continue;
          auto entry = [&](int head) {
            h_upd.push_back(head); h_upd.push_back(ti);
            for (int k : tg.second) { h_klist.push_back(slot(ti, k)); h_klist.push_back(slot(tj, k)); }
            h_kptr.push_back((int)(h_klist.size() / 2));
          };
          if (diag_next && fuse_quads && (int)targets.size() <= quad_max_targets) { // helpers first: a lower block index is dispatched no later than its factorising workgroup
            entry(slot(ti, tj) | SP_QUAD_BIT | SP_QHELP_BIT | (2 << SP_QUAD_SHIFT));
            entry(slot(ti, tj) | SP_QUAD_BIT | SP_QHELP_BIT | (3 << SP_QUAD_SHIFT));
            entry(slot(ti, tj) | SP_QUAD_BIT | SP_FUSE_BIT);
            tiles_l += 0.75 * tg.second.size();
          } else {
            // (measured, not kept: EVERY target by quadrants — four small workgroups instead of one 128 x 128 one: 352 vs 362 LM it/s.  The
            // launch's LDS size is the factorising workgroups' 132 KB for every workgroup, so the small ones do not share a CU either.)
            entry(slot(ti, tj) | (diag_next ? SP_FUSE_BIT : 0));
            tiles_l += 1.0 * tg.second.size();
          }
        }
      lvl_panel_off.push_back((int)h_panels.size());
      lvl_trsm_off.push_back((int)(h_trsm.size() / 2));
      lvl_upd_off.push_back((int)(h_upd.size() / 2));
      lvl_upd_tiles.push_back(tiles_l);
    }
    // row structure (forward substitution) and column structure (backward), cut into items of <= slice tiles
    std::vector<int> h_rptr(nt + 1, 0), h_rcols, h_rslot, h_cptr(nt + 1, 0), h_crows, h_cslot;
    for (int k = 0; k < nt; ++k) {
      for (int j = 0; j < k; ++j) if (nz(k, j)) { h_rcols.push_back(j); h_rslot.push_back(slot(k, j)); }
      h_rptr[k + 1] = (int)h_rcols.size();
      for (int i : U[k]) { h_crows.push_back(i); h_cslot.push_back(slot(i, k)); }
      h_cptr[k + 1] = (int)h_crows.size();
    }
    std::vector<int> it_f[5], it_b[5]; // panel, beg, end, first, count
    lvl_fitem_off.assign(1, 0); lvl_bitem_off.assign(1, 0);
    auto cut = [&](std::vector<int> (&it)[5], int k, int beg, int end) {
      const int cnt = std::max(1, (end - beg + slice - 1) / slice), first = (int)it[0].size();
      for (int q = 0; q < cnt; ++q) {
        it[0].push_back(k); it[1].push_back(std::min(end, beg + q * slice)); it[2].push_back(std::min(end, beg + (q + 1) * slice));
        it[3].push_back(first); it[4].push_back(cnt);
      }
    };
    for (int l = 0; l < nlevels; ++l) {
      for (int k : by_level[l]) { cut(it_f, k, h_rptr[k], h_rptr[k + 1]); cut(it_b, k, h_cptr[k], h_cptr[k + 1]); }
      lvl_fitem_off.push_back((int)it_f[0].size()); lvl_bitem_off.push_back((int)it_b[0].size());
    }
    for (int q = 0; q < 5; ++q) { d_itf[q].upload(it_f[q], stream); d_itb[q].upload(it_b[q], stream); }
    { // entries of k_sp_bwd_chain: top level first; a panel's items (one tile each), then its finisher
      std::vector<int> ch[6]; // kind_panel, row, slot, idx | pfirst, pcount (per panel)
      ch[4].assign(nt, 0); ch[5].assign(nt, 0);
      int prow = 0;
      for (int l = nlevels - 1; l >= 0; --l)
        for (int k : by_level[l]) {
          const int nk_ = h_cptr[k + 1] - h_cptr[k];
          ch[4][k] = prow; ch[5][k] = nk_;
          for (int q = 0; q < nk_; ++q) { ch[0].push_back(k); ch[1].push_back(h_crows[h_cptr[k] + q]); ch[2].push_back(h_cslot[h_cptr[k] + q]); ch[3].push_back(q); }
          ch[0].push_back(~k); ch[1].push_back(0); ch[2].push_back(0); ch[3].push_back(0);
          prow += nk_;
        }
      chain_entries = (int)ch[0].size();
      for (int q = 0; q < 6; ++q) d_ch[q].upload(ch[q], stream);
      d_bcnt.alloc(nt); d_bcnt.zero(stream); d_ready.alloc(nt); d_ready.zero(stream);
      chain_seq = 0;
    }
    partial.alloc((size_t)std::max({it_f[0].size(), it_b[0].size(), h_crows.size()}) * CH_NB);
    ticket.alloc(nt); ticket.zero(stream);
    auto up = [&](DevBuf<int> &d, std::vector<int> &h) { if (h.empty()) h.push_back(0); d.upload(h, stream); };
    up(d_panels, h_panels); up(d_trsm, h_trsm); up(d_upd, h_upd); up(d_klist, h_klist); up(d_nz, h_nz);
    d_kptr.upload(h_kptr, stream); d_rptr.upload(h_rptr, stream); up(d_rcols, h_rcols); d_cptr.upload(h_cptr, stream); up(d_crows, h_crows);
    up(d_rslot, h_rslot); up(d_cslot, h_cslot); d_tmap.upload(tmap, stream); d_dslot.upload(dslot, stream);
    d_camcol.upload(camcol, stream); d_src.upload(src, stream); d_pad.upload(pad, stream);
    d_qcnt.alloc(nt); d_qcnt.zero(stream);
    d_fail.alloc(1); // the matrix itself (bytes()) is allocated by allocate(), once the caller has decided to use this solver
    if (!h_fail) GR_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_fail), sizeof(int), hipHostMallocDefault));
    if (!attrs_set) {
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sp_gemm<T, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_gemm_lds(sizeof(T))));
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sp_gemm<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(chol_gemm_lds(sizeof(T)), chol_potrf_lds(sizeof(T)))));
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sp_potrf<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_potrf_lds(sizeof(T))));
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sp_update_fwd<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(chol_gemm_lds(sizeof(T)), chol_potrf_lds(sizeof(T)))));
      attrs_set = true;
    }
    GR_HIP(hipStreamSynchronize(stream));
    return true;
  }
  // the factor's tiles (fill included) and the per-panel inverses of the diagonal tiles: memory follows nnz(L)
  size_t bytes() const { return ((size_t)nz_tiles + (size_t)nt) * SP_TT * sizeof(T); }
  size_t dense_bytes() const { return (size_t)npad * npad * sizeof(T); } // what the padded dense array of rounds 2-3 took
  // the tiles, the tile inverses and the substitution vectors: only after the caller's memory guard / solver choice
  void allocate() {
    A.alloc((size_t)nz_tiles * SP_TT); Linv.alloc((size_t)nt * SP_TT);
    Linv.zero(stream); // chol_potrf_block writes the lower triangle of every inverse only
    vb.alloc(npad); vy.alloc(npad); vx.alloc(npad);
  }

  struct Sc {
    CholProfSink *s;
    Sc(CholProfSink *s_, const char *nm, double by, double fl) : s(s_) { if (s) s->begin(nm, by, fl); }
    ~Sc() { if (s) s->end(); }
  };
  // S (upper 9x9 blocks) -> permuted dense lower triangle
  void load(int64_t nnzb, const int *rowi, const int *coli, const T *S) {
    k_sp_clear<T><<<nz_tiles, 256, 0, stream>>>(A.p, d_pad.p, d_nz.p);
    k_sp_scatter<T><<<(unsigned)(((int64_t)bs * bs * nnzb + 255) / 256), 256, 0, stream>>>(nnzb, rowi, coli, d_camcol.p, S, A.p, d_tmap.p, nt, bs);
  }
  void factor() { factor_levels([](int) {}); }
  template <typename After> void factor_levels(After &&after_level, bool ride_fwd = false) {
    GR_HIP(hipMemsetAsync(d_fail.p, 0, sizeof(int), stream));
    const size_t lds_g = chol_gemm_lds(sizeof(T)), lds_p = chol_potrf_lds(sizeof(T));
    const double tb = (double)CH_NB * CH_NB * sizeof(T), tf = 2.0 * CH_NB * CH_NB * CH_NB;
    for (int l = 0; l < nlevels; ++l) {
      const int np_ = lvl_panel_off[l + 1] - lvl_panel_off[l], ntr = lvl_trsm_off[l + 1] - lvl_trsm_off[l], nup = lvl_upd_off[l + 1] - lvl_upd_off[l];
      if (l == 0 || !fuse_potrf) { // deeper levels: factorised by the previous level's update launch
        Sc sc(sink, "spchol_potrf", 3.0 * np_ * tb, np_ * tf / 3);
        k_sp_potrf<T><<<np_, SP_PT, lds_p, stream>>>(A.p, d_panels.p + lvl_panel_off[l], d_dslot.p, Linv.p, d_fail.p);
      }
      if (ntr) {
        Sc sc(sink, "spchol_trsm", 3.0 * ntr * tb, ntr * tf);
        if (row_split_trsm) k_sp_trsm_rows<T><<<(CH_NB / SP_SLAB) * ntr, 256, sp_trsm_lds(sizeof(T)), stream>>>(A.p, d_trsm.p + 2 * (size_t)lvl_trsm_off[l], Linv.p);
        else k_sp_gemm<T, 0><<<ntr, 256, lds_g, stream>>>(A.p, d_trsm.p + 2 * (size_t)lvl_trsm_off[l], nullptr, nullptr, Linv.p);
      }
      const int nf = ride_fwd ? lvl_fitem_off[l + 1] - lvl_fitem_off[l] : 0;
      if (nup) {
        const double nk = lvl_upd_tiles[l];
        Sc sc(sink, "spchol_update", (2.0 * nup + 2.0 * nk) * tb, nk * tf);
        if (nf) k_sp_update_fwd<T><<<nup + nf, 256, std::max(lds_g, lds_p), stream>>>(A.p, d_upd.p + 2 * (size_t)lvl_upd_off[l], d_kptr.p + lvl_upd_off[l], d_klist.p, Linv.p, d_fail.p, nup,
                                                                                     items(d_itf, lvl_fitem_off[l]), d_rcols.p, d_rslot.p, vb.p, vy.p, partial.p, ticket.p, d_qcnt.p);
        else k_sp_gemm<T, 1><<<nup, 256, std::max(lds_g, lds_p), stream>>>(A.p, d_upd.p + 2 * (size_t)lvl_upd_off[l], d_kptr.p + lvl_upd_off[l], d_klist.p, nullptr, Linv.p, d_fail.p, d_qcnt.p);
      } else if (nf) k_sp_fwd<T><<<nf, 256, 0, stream>>>(A.p, Linv.p, items(d_itf, lvl_fitem_off[l]), d_rcols.p, d_rslot.p, vb.p, vy.p, partial.p, ticket.p);
      // level l's panels (L_kk^-1, trsm'ed sub-diagonal tiles) are final here; what the update launch above still writes are
      // tiles of LATER columns
      after_level(l);
    }
  }
  // factor() with the forward substitution of `b` riding beside it: level l of L y = b only needs the panels of level l, so
  // it is enqueued on a second stream behind level l's factor kernels and runs under level l + 1's (a level's substitution
  // launch is 6-25 us of latency, its factor launches 110 us: hidden except for the last level).  Then back-substitution.
  hipStream_t aux = nullptr;
  std::vector<hipEvent_t> lvl_done;
  hipEvent_t aux_done = nullptr, rhs_ready = nullptr;
  int overlap_form = 1; // 1: level l's forward substitution rides in level l's update launch; 2: on a second stream behind an event per level (round 3)
  void factor_solve(const T *b, T *x) {
    if (overlap_form != 2) {
      k_sp_rhs<T><<<(npad + 255) / 256, 256, 0, stream>>>(npad, d_src.p, b, vb.p);
      factor_levels([](int) {}, true);
      Sc sc(sink, "spchol_solve", 1.0 * (double)factor_tiles * CH_NB * CH_NB * sizeof(T), 2.0 * (double)factor_tiles * CH_NB * CH_NB);
      backward();
      k_sp_unpermute<T><<<(npad + 255) / 256, 256, 0, stream>>>(npad, d_src.p, vx.p, x);
      return;
    }
    if (!aux) {
      GR_HIP(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
      GR_HIP(hipEventCreateWithFlags(&aux_done, hipEventDisableTiming));
      GR_HIP(hipEventCreateWithFlags(&rhs_ready, hipEventDisableTiming));
    }
    while ((int)lvl_done.size() < nlevels) { hipEvent_t e; GR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); lvl_done.push_back(e); }
    k_sp_rhs<T><<<(npad + 255) / 256, 256, 0, stream>>>(npad, d_src.p, b, vb.p);
    GR_HIP(hipEventRecord(rhs_ready, stream));
    GR_HIP(hipStreamWaitEvent(aux, rhs_ready, 0));
    factor_levels([&](int l) {
      GR_HIP(hipEventRecord(lvl_done[l], stream));
      GR_HIP(hipStreamWaitEvent(aux, lvl_done[l], 0));
      k_sp_fwd<T><<<lvl_fitem_off[l + 1] - lvl_fitem_off[l], 256, 0, aux>>>(A.p, Linv.p, items(d_itf, lvl_fitem_off[l]), d_rcols.p, d_rslot.p, vb.p, vy.p, partial.p, ticket.p);
    });
    GR_HIP(hipEventRecord(aux_done, aux));
    GR_HIP(hipStreamWaitEvent(stream, aux_done, 0));
    Sc sc(sink, "spchol_solve", 1.0 * (double)factor_tiles * CH_NB * CH_NB * sizeof(T), 2.0 * (double)factor_tiles * CH_NB * CH_NB);
    backward();
    k_sp_unpermute<T><<<(npad + 255) / 256, 256, 0, stream>>>(npad, d_src.p, vx.p, x);
  }
  void backward() {
    if (bwd_chain) {
      ++chain_seq;
      k_sp_bwd_chain<T><<<chain_entries, 256, 0, stream>>>(A.p, Linv.p, SpChain{d_ch[0].p, d_ch[1].p, d_ch[2].p, d_ch[3].p, d_ch[4].p, d_ch[5].p}, vy.p, vx.p, partial.p, d_bcnt.p, d_ready.p, chain_seq, d_fail.p);
      return;
    }
    for (int l = nlevels - 1; l >= 0; --l)
      k_sp_bwd<T><<<lvl_bitem_off[l + 1] - lvl_bitem_off[l], 256, 0, stream>>>(A.p, Linv.p, items(d_itb, lvl_bitem_off[l]), d_crows.p, d_cslot.p, vy.p, vx.p, partial.p, ticket.p);
  }
  // b, x: device vectors of length n in the CALLER's (camera-major) order (x may alias b)
  void solve(const T *b, T *x) {
    Sc sc(sink, "spchol_solve", 2.0 * (double)factor_tiles * CH_NB * CH_NB * sizeof(T), 4.0 * (double)factor_tiles * CH_NB * CH_NB);
    k_sp_rhs<T><<<(npad + 255) / 256, 256, 0, stream>>>(npad, d_src.p, b, vb.p);
    for (int l = 0; l < nlevels; ++l)
      k_sp_fwd<T><<<lvl_fitem_off[l + 1] - lvl_fitem_off[l], 256, 0, stream>>>(A.p, Linv.p, items(d_itf, lvl_fitem_off[l]), d_rcols.p, d_rslot.p, vb.p, vy.p, partial.p, ticket.p);
    backward();
    k_sp_unpermute<T><<<(npad + 255) / 256, 256, 0, stream>>>(npad, d_src.p, vx.p, x);
  }
  bool ok() {
    GR_HIP(hipMemcpyAsync(h_fail, d_fail.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
    if (*h_fail != 0) {
      // a bounded wait gave up: workgroups that arrived after it have left the per-panel counters, the quadrant counters
      // and the ready words of the dependency-driven launches in an unknown state — clear them (and the chain's sequence number)
      // before anybody factorises again, or a later finisher could be released early and read stale partial sums
      if (d_bcnt.n) d_bcnt.zero(stream);
      if (d_ready.n) d_ready.zero(stream);
      if (d_qcnt.n) d_qcnt.zero(stream);
      chain_seq = 0;
      GR_HIP(hipStreamSynchronize(stream));
      return false;
    }
    return true;
  }
};

} // namespace gr
