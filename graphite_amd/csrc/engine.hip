// libgraphite_mi355x.so — host side of the BAL hot path on MI355X.
//
// Engine<T> owns the device-resident problem and mirrors, method for method,
// the reference objects on this path:
//   Graph<T,S>            linearize / chi2 / apply_update / backup / revert   (graph.hpp)
//   Hessian<T,S>          update_values / apply_damping                        (hessian.hpp)
//   SchurComplement<T,S>  build_structure / update_values / landmark update    (schur.hpp)
//   PCGSchurSolver, PCGSolver + block-Jacobi preconditioners                   (solver/, preconditioner/)
//   optimizer::levenberg_marquardt                                             (optimizer/levenberg_marquardt.hpp)
// The C ABI at the bottom (include/graphite_mi355x.h) is the drop-in boundary.
#include "../../include/graphite_mi355x_test.h"
#include "../../include/graphite_mi355x_model.h"
#include "comm.hpp"
#include <hip/hip_ext.h>
#include <functional>
#include "kernels_is.hpp"
#include "chol.hpp"
#include "sparse_chol.hpp"
#include "kernels_model.hpp"
#include "kernels_sf.hpp"
#include "kernels_rp.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <numeric>

namespace gr {

static thread_local std::string g_last_error;

static inline int cdiv(size_t a, size_t b) { return (int)((a + b - 1) / b); }

static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static void tuning_default(gr_bal_tuning &t) {
  std::memset(&t, 0, sizeof(t));
  t.point_tiles = env_int("GR_PTILES", -1);
  t.g3_gather = env_int("GR_G3_GATHER", -1);
  t.point_records = env_int("GR_POINT_RECORDS", -1);
  t.pcg_lazy = env_int("GR_PCG_LAZY", -1);
  t.pcg_single_reduction = env_int("GR_PCG_CG", -1);
  t.sparse_cholesky = env_int("GR_SPARSE_CHOL", -1);
  t.spchol_overlap = env_int("GR_SPCHOL_OVERLAP", 1);
  t.lm_speculate = env_int("GR_LM_SPECULATE", 1);
  t.lm_ahead = env_int("GR_LM_AHEAD", 1);
  t.lm_fused = env_int("GR_LM_FUSED", 1);
  t.grid_mult = std::max(0, env_int("GR_GRID_MULT", 0));
  t.vec_per_thread = std::max(1, env_int("GR_VEC_PER_THREAD", 2)); // replayed back to back on Ladybug-1723: 5.3 us (1), 4.4 (2), 4.9 (4), 5.8 (8)
  t.schur_item = env_int("GR_SCHUR_ITEM", 56);
  t.verbose = getenv("GR_VERBOSE") ? 1 : 0;
  t.ipc_timeout_ms = env_int("GR_IPC_TIMEOUT_MS", 30000);
  t.shard_fused = env_int("GR_SHARD_FUSED", -1);
  t.chol_fuse = env_int("GR_CHOL_FUSE", 1);
  t.chol_pin = env_int("GR_CHOL_PIN", 1);
  t.spchol_fuse = env_int("GR_SPCHOL_FUSE", 2);
  t.spchol_bwd_chain = env_int("GR_SPCHOL_BWD_CHAIN", 1);
  t.schur_fused = env_int("GR_SCHUR_FUSED", -1);
  t.pcg_resident = env_int("GR_PCG_RESIDENT", -1);
  t.comm_transport = env_int("GR_COMM_TRANSPORT", -1);
  t.spchol_slice = std::max(1, env_int("GR_SPCHOL_SLICE", 1)); // tiles per substitution item (Ladybug-1723 direct Schur: 1 -> 322.5, 2 -> 319.5, 3 -> 314.5, 4 -> 310 LM it/s)
}

// GR_VERBOSE: host laps of the set-up phases (gr_bal_create is what a drop-in user waits for before the first iteration)
struct Laps {
  bool on; const char *tag;
  std::chrono::steady_clock::time_point t;
  Laps(bool on_, const char *tag_) : on(on_), tag(tag_), t(std::chrono::steady_clock::now()) {}
  void operator()(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[graphite-mi355x] %s: %s %.2f ms\n", tag, what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

struct KernelProf {
  std::string name;
  int64_t launches = 0, noop = 0; // noop: look-ahead launches that found the PCG loop already finished
  double total_ms = 0, bytes = 0, flops = 0; // bytes / flops: summed over the scoped launches
  int64_t scoped = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct EngineBase {
  virtual ~EngineBase() = default;
  virtual void set_loss(int kind, double delta) = 0;
  virtual void set_scale_system(bool on) = 0;
  virtual void set_fixed(const unsigned char *cam_fixed, const unsigned char *pt_fixed) = 0;
  virtual void set_jacobian_precision(int dtype) = 0;
  virtual void set_params(const void *c, const void *p) = 0;
  virtual void get_params(void *c, void *p) = 0;
  virtual void linearize() = 0;
  virtual double chi2() = 0;
  virtual void backup() = 0;
  virtual void revert() = 0;
  virtual void apply_update(const void *dx) = 0;
  virtual void solver_update_structure(int solver) = 0;
  virtual void solver_update_values(int solver) = 0;
  virtual void solver_set_damping(int solver, double mu, bool use_identity) = 0;
  virtual bool solver_solve(int solver, int max_iter, double tol, double rej, void *dx, int *iters) = 0;
  virtual void schur_update_values() = 0;
  virtual void schur_matvec(const void *x, void *y) = 0;
  virtual void landmark_update(const void *xp, void *xl) = 0;
  virtual void schur_structure(int64_t *nnzb, int64_t *colptr, int64_t *rowidx) = 0;
  virtual void get(int which, void *out, int64_t *count) = 0;
  virtual void hessian_structure(int64_t *nblocks, int64_t *colptr, int64_t *rowidx, int64_t *offsets) = 0;
  virtual void export_csc(int which, int64_t *nnz, int64_t *indptr, int64_t *indices, void *values) = 0;
  virtual void lm(const gr_lm_options &opt, gr_lm_stats &st, double *chi2_trace, double *lambda_trace) = 0;
  virtual int kernel_stats(gr_kernel_stat *out, int cap) = 0;
  virtual void direct_solver_info(gr_direct_solver_info &o) = 0;
  virtual double diag_time(int which, int variant, int reps) = 0;
  virtual void set_comm(std::unique_ptr<Comm> c) = 0;
  virtual void set_contributors(const unsigned *mask, int64_t count) = 0;
  virtual void allreduce_host(double *v, size_t n) = 0;
  virtual void apply_tuning() = 0; // after `tune` changed
  virtual void orders(int32_t *obs_order, int32_t *landmark_order) = 0;
  virtual void comm_info(gr_comm_info &o) = 0;
  gr_bal_tuning tune;
  std::vector<double> lm_iter_seconds; // last lm(): host time from the start of the loop at which iteration i's decision was observed
  EngineBase() { tuning_default(tune); }
  int device = 0;
  char *ipc_box = nullptr; // mailbox allocated by gr_bal_comm_ipc_mailbox, owned by the IpcComm once it exists
  size_t ipc_slot = 0;
  int ipc_world = 0;
};

template <typename T> struct Engine final : EngineBase {
  hipStream_t stream = nullptr;
  int64_t Nc = 0, Np = 0, No = 0;
  size_t n = 0, pose_dim = 0;
  int loss_kind = 0;
  T loss_delta = 0;
  bool scale_system = true;
  bool jac32 = false; // Jacobian entries evaluated in fp32 and promoted (Graph<double, float>); T = double only
  // USER-TRAITS engine (graphite_mi355x_model.h): the kernels that call the user's error / jacobian / update live in the user's
  // translation unit and are reached through this table; nullptr = the built-in BAL camera model (bal_device.hpp)
  const gr_model_ops *model = nullptr;
  DevBuf<char> jst;          // stored weighted Jacobian, GR_MODEL_JSTREAMS streams of jst_stride scalars of the storage type
  int64_t jst_stride = 0;
  hipEvent_t hook_b = nullptr; // profiling: the stop event of a scope whose body is a user-side launcher
  void hook_begin() { ++launch_count; if (ext_a) { (void)hipEventRecord(ext_a, stream); hook_b = ext_b; ext_a = ext_b = nullptr; } }
  void hook_end(int err, const char *what) {
    if (hook_b) { (void)hipEventRecord(hook_b, stream); hook_b = nullptr; }
    if (err != 0) throw HipError(std::string("user-traits kernel launcher `") + what + "` failed: " + hipGetErrorString((hipError_t)err));
  }
  void no_model(const char *what) const { if (model) throw std::invalid_argument(std::string(what) + ": not available on a user-traits problem (gr_bal_create_model)"); }

  // host structure
  std::vector<int> h_pt_ptr, h_cam_pm, h_pt_pm, h_cam_ptr, h_pt_cm, h_pos_cm, h_pm_of_orig;
  std::vector<int> h_chunk_cam, h_chunk_beg, h_cam_chunk_ptr, h_cam_seg_ptr;
  std::vector<int> h_pt_new2old, h_pt_old2new; // internal point order: sorted by first observing camera
  int nch = 0, nb_pm = 0, nseg = 0;
  int num_cu = 256, grid_obs = 0, grid_vec = 0, grid_chi2 = 0; // persistent grids
  int fbj_per_cu = 4;            // k_finalize_bj: workgroups resident per CU (occupancy query)
  int grid_lin = 0, grid_op = 0; // k_linearize / k_pcg_operator: exactly the workgroups that are resident at once (apply_tuning)
  // Schur structure (lazy)
  bool schur_ready = false;
  int64_t nnzb = 0, nprod = 0;
  std::vector<int> h_S_colptr, h_S_rowi, h_S_coli, h_S_diag;

  // device data
  DevBuf<T> cams, pts, cams_bak, pts_bak, pack, obs_pm, obs_cm;
  DevBuf<int> pt_ptr, cam_pm, pt_pm, cam_ptr, pt_cm, pos_cm, cam_cm, chunk_cam, chunk_beg, cam_chunk_ptr;
  DevBuf<T> cam_partial, op_partial, g9, g3, part9, xl_acc;
  DevBuf<int> boundary_flag;
  DevBuf<unsigned> ticket;
  DevBuf<int> cam_seg_ptr;
  // Observation order of the per-observation kernels.  Plain: the camera-major order above.  TILED (Venice / Final
  // shapes, build_tiled_order): (point tile, camera, point) with point tiles of equal observation count sized for an
  // XCD's L2 and dealt round-robin to the 8 XCDs, so that the point gathers and the g3 / g9 scatters of a whole
  // XCD fall into one L2-resident window instead of costing a 128-byte line each.  Segments = (64-observation block,
  // camera run) pairs: blk_seg[b] = flat id of block b's first segment (the runs that start inside b follow it),
  // seg_slot[flat id] = its slot among its camera's segments, cam_seg_ptr[c] = first slot of camera c.
  bool tiled = false;
  int n_ptiles = 0;
  DevBuf<int> t_cam, t_pt, t_pos, blk_seg, seg_slot;
  DevBuf<T> t_obs;
  const int *o_cam() const { return tiled ? t_cam.p : cam_cm.p; }
  const int *o_pt() const { return tiled ? t_pt.p : pt_cm.p; }
  const int *o_pos() const { return tiled ? t_pos.p : pos_cm.p; }
  const T *o_obs() const { return tiled ? t_obs.p : obs_cm.p; }
  int o_ntiles() const { return tiled ? -nb_pm : nb_pm; }
  // g3 (the operator's per-observation J_p^T w) in observation order, gathered by the update kernel: kernels_mf.hpp G3Gather
  // Lazy PCG direction (kernels_mf.hpp PcgState): default on; GR_PCG_LAZY=0 keeps the direction kernel's vector pass.
  // GR_PCG_LAZY=0/1 forces it.  Measured A/B on one box (LM it/s, bench line): Ladybug-49 fp32 13 230 -> 14 000,
  // Ladybug-1723 fp64 5 060 -> 5 100, Venice-1778 fp32 1 165 -> 1 210, Final-13682 fp64 148.5 -> 139.9: the direction
  // kernel's pass over five n-vectors goes, but the update kernel stores three more vectors (p, s.*p, s.*z'), the operator
  // gathers two per point instead of one and every wave of it re-derives the loop decision from the dot-product slots
  // (Ladybug-1723: operator 24.9 -> 29.7 us, update 21.3 -> 27.3 us for the 12.7 us direction launch saved).  It pays
  // where launches dominate and is a wash or a loss where bytes do.  Default: vectors up to 1 MB (Ladybug-49-sized
  // problems).  Off in diagnostic builds (their kernels are the direction-kernel form).
  bool pcg_lazy() const {
#ifdef GR_DIAG
    return false; // the ablation variants of tools/diag_*.py are instantiated for the direction-kernel form only
#endif
    // round 3: off by default at every size — the fused LM iteration (direction-kernel form, k_finalize_bj) beats the lazy host
    // loop also where launches dominate (Ladybug-49 fp32 PCG: 15 590 lazy host loop -> 17 520 LM it/s fused; fp64 14 400 -> 15 170)
    return lazy_cfg < 0 ? false : lazy_cfg != 0;
  }
  int lazy_cfg = -1; // gr_bal_tuning.pcg_lazy, latched per solver_update_structure
  // Single-reduction PCG (kernels_mf.hpp PcgState, lazy == 2): on landmark shards by default — ONE all-reduce per inner
  // iteration (camera rows + every dot product of the iteration) instead of two; GR_PCG_CG=0/1 forces it (1 also on a
  // single GPU: that is how tests hold it to the oracle's solve_pcg_cg).
  int cg_cfg = -1;
  bool pcg_cg() const {
#ifdef GR_DIAG
    return false;
#endif
    return cg_cfg < 0 ? (comm && comm->size > 1) : cg_cfg != 0;
  }
  int pcg_mode() const { return model ? 0 : (pcg_cg() ? 2 : (pcg_lazy() ? 1 : 0)); }
  DevBuf<T> v_sv;
  DevBuf<T> v_zs;
  bool g3_obs_order = false;
  // GR_G3_GATHER=0/1 forces it.  Default: with the point-tiled order only, i.e. on graphs whose g3 does not stay in the
  // caches between the operator and the update kernel.  There the scattered 12/24-byte stores of the pm layout left L2
  // one by one (Venice-1778 fp32, rocprofv3: 5.46 M store requests in, 4.94 M memory writes out, 242 MB for 60 MB of
  // payload); in observation order the operator fell from 176 to 86 us inside the solve and the update kernel rose from
  // 83 to 105 us (Final-13682 fp64: 1481 -> 786 and 472 -> 712).  Ladybug-1723 (everything cache-resident, plain order):
  // operator 23.1 -> 20.5 us, update 20.3 -> 23.7 us, no gain, so it keeps the pm layout.
  bool want_g3_gather() const { return tune.g3_gather >= 0 ? tune.g3_gather == 1 : (tiled || g3_plain_pays); }
  bool g3_plain_pays = false; // plain order, working set beyond the L2s: decided by timing operator + update both ways (tune_tiling)
  DevBuf<int> g3_gidx, ptile_ptr;
  int g3_ptiles = 0;
  const int *g3_pos() const { return g3_obs_order ? nullptr : o_pos(); }
  G3Gather g3_gather() const {
    G3Gather g;
    if (g3_obs_order) { g.gidx = g3_gidx.p; g.ptile_ptr = ptile_ptr.p; g.n_ptiles = g3_ptiles; }
    g.cam_fixed = cam_fixed_p(); g.pt_fixed = pt_fixed_p();
    return g;
  }
  int is_points_blocks() const { return g3_obs_order ? std::max(8, std::min(cdiv(Np, TPB), num_cu * 8) / 8 * 8) : cdiv(Np, TPB); }
  int update_blocks() const {
    const int b = std::min(cdiv(pose_dim, 252) + cdiv(Np, 85), num_cu * 8);
    return g3_obs_order ? std::max(8, (b + 7) / 8 * 8) : b;
  }
  // multi-GPU: landmark shard of a larger problem (comm != null), camera rows all-reduced
  std::unique_ptr<Comm> comm;
  bool shard = false;
  DevBuf<T> raw_c;
  int cam_weight() const { return (!comm || comm->rank == 0) ? 1 : 0; }
  void set_comm(std::unique_ptr<Comm> c) override { comm = std::move(c); comm->set_timeout_ms(tune.ipc_timeout_ms); raw_c.alloc(pose_dim); }
  // after a stream synchronisation that follows an all-reduce whose result the HOST consumes (block structure, chi2 / rho of
  // the LM decision, stop-flag agreement): a rank whose wait timed out holds rank-local values and must not act on them
  void check_comm(const char *where) {
    if (comm && comm->failed()) throw CommError(std::string(where) + ": an all-reduce timed out waiting for a peer (IPC mailbox transport); the result is not valid");
  }
  void allreduce_host(double *v, size_t n) override {
    if (!comm) throw std::invalid_argument("no communicator");
    DevBuf<double> d;
    std::vector<double> h(v, v + n);
    d.upload(h, stream);
    comm->allreduce(d.p, n, true, stream);
    h = d.download(stream);
    check_comm("allreduce_host");
    std::copy(h.begin(), h.end(), v);
  }
  int64_t coll_count = 0; // collectives issued (a group counts once): gr_lm_stats.collectives
  bool coll_in_group = false;
  void allreduce_T(T *buf, size_t count) { if (!coll_in_group) { ++coll_count; ++comm_launches; } comm->allreduce(buf, count, sizeof(T) == 8, stream); }
  void allreduce_d(double *buf, size_t count) { if (!coll_in_group) { ++coll_count; ++comm_launches; } comm->allreduce(buf, count, true, stream); }
  int64_t comm_launches = 0; // kernels the communicator launched for them (one per message on the mailbox transport)
  void group_start() { ++coll_count; ++comm_launches; coll_in_group = true; comm->group_start(); }
  void group_end() { comm->group_end(); coll_in_group = false; }
  // matrix-free PCG control
  DevBuf<double> ctl; // PCG slot accumulators + loop state
  DevBuf<int> ctl_i;
  DevBuf<double> grid_partial;
  DevBuf<int> pcg_iters;
  int ctl_cap = 0, state_fresh_cap = -1; // state_fresh_cap == ctl_cap: the loop state was reset by the last set_damping
  // LM host loop: lm_x = its step vector; state_clean_cap == ctl_cap: the loop state was cleared by the trial-step kernel and
  // no solve ran since, so set_damping may start the next PCG loop inside k_block_jacobi (update0_done)
  T *lm_x = nullptr;
  int state_clean_cap = -1;
  bool update0_done = false, update0_identity = false;
  // pinned host mirror: h_res[0..3] = trial chi2, rho denominator, new damping, accepted (the last two from k_finalize_bj);
  // h_seq[0] = sequence number of the last published result, h_seq[1 + bank] = iteration count of the PCG loop that uses
  // flag bank `bank`; h_flag = two banks of per-iteration exit flags.  Two banks because, on an accept streak, the first
  // kernels of the NEXT solve are enqueued while look-ahead launches of the current one may still write their flags.
  double *h_res = nullptr;
  volatile int *h_seq = nullptr, *h_flag = nullptr;
  int h_flag_cap = 0, seq_counter = 0, flag_bank = 0;
  volatile int *flags() const { return h_flag + (size_t)flag_bank * h_flag_cap; }
  volatile int *h_iters() const { return h_seq + 1 + flag_bank; }
  DevBuf<T> Hcc, Hll, Hcp, scales, bu; // bu = [bc (9Nc) ; bl (3Np)] unscaled -J^T rho' r
  struct View { T *p = nullptr; } bc, bl;
  DevBuf<double> chi2_partial, dscalars; // dscalars[0]=chi2, [1]=rho denom
  // Schur
  DevBuf<int> prod_a, prod_b, prod_pm, S_rowi, S_coli, S_diag, row_ptr, row_blk, row_col;
  DevBuf<int> item_blk, item_beg, item_end, item_single, multi_blk;
  DevBuf<int> item_multi, multi_first, multi_n; // k_schur_reduce (kernels_sf.hpp)
  DevBuf<unsigned> multi_cnt;
  DevBuf<T> slab;
  int nitems = 0, nmulti = 0;
  DevBuf<T> S, b_schur, Hll_inv, Mp, vl, MinvS;
  // PCG work vectors
  DevBuf<T> v_r, v_p, v_z, v_Ap, v_xb, v_dx, v_ps, v_diag, MinvC, MinvP;
  DevBuf<T> xp;          // point records of the matrix-free operator [X Y Z ps(3) pad(2)]
  bool xp_valid = false; // X part in step with pts
  // the records pay where the point / direction gathers miss L2 (Venice / Final shapes: -27 % operator time)
  // and cost a few % where they hit (banded Ladybug): decided ONCE per problem by timing the operator both
  // ways (same arithmetic either way, so results do not depend on the choice).  GR_POINT_RECORDS=0/1 forces it.
  bool use_records = false, records_tuned = false;
  void tune_point_records() {
    if (records_tuned) return;
    records_tuned = true;
    if (model) { use_records = false; return; } // landmark parameters are the user's vertex objects, not [X Y Z] records
    if (pcg_mode() == 1) { use_records = false; return; } // the lazy form gathers zs AND ps; the 8-scalar record has room for one (single-reduction form: zs only)
    if (tune.point_records >= 0) { use_records = tune.point_records != 0; return; }
    use_records = false;
    const double t_plain = diag_time(0, 0, 5);
    use_records = true;
    const double t_rec = diag_time(0, 0, 5);
    use_records = t_rec < 0.97 * t_plain;
    xp_valid = false;
    if (tune.verbose) std::fprintf(stderr, "[graphite-mi355x] operator %.1f us plain, %.1f us with point records -> %s\n", t_plain, t_rec, use_records ? "records" : "plain");
  }
  void ensure_point_records() {
    if (!use_records) return;
    if (xp.n != 8 * (size_t)Np) { xp.alloc(8 * (size_t)Np); xp.zero(stream); xp_valid = false; }
    if (!xp_valid) { k_points_to_records<T><<<cdiv(3 * (size_t)Np, TPB), TPB, 0, stream>>>((int)Np, pts.p, xp.p); xp_valid = true; }
  }
  DevBuf<double> sc_d;
  DevBuf<int> sc_i;
  int sc_cap = 0;
  // tmp: staging of get() — b (n), Hcp (27 No), Hcc (81 Nc) —, allocated by its first use (146 MB on Ladybug-1723 that an
  // optimisation never touches)
  DevBuf<T> tmp;
  void ensure_tmp() { tmp.alloc(std::max({n, 27 * (size_t)No, 81 * (size_t)Nc})); }

  double damping = 0;
  bool damping_identity = false;
  bool hcp_valid = false;
  int n_point_blocks = 0, n_chi2_blocks = 0;

  // profiling
  bool profiling = false;
  std::map<std::string, KernelProf> prof;

  Engine(int64_t nc, int64_t np, int64_t no, const void *c, const void *p, const void *o,
         const int32_t *ci, const int32_t *pi, int dev, hipStream_t s, bool shard_ = false, const gr_model_ops *model_ = nullptr) {
    device = dev;
    shard = shard_;
    stream = s;
    model = model_;
    Nc = nc; Np = np; No = no;
    pose_dim = 9 * (size_t)Nc;
    n = pose_dim + 3 * (size_t)Np;
    Laps lap(tune.verbose != 0, "gr_bal_create");
    std::vector<T> hc(9 * Nc), hp(3 * Np), ho(model ? 0 : 2 * No); // user-traits problems: vertex values and observations stay on the user side
    std::vector<int32_t> hci(No), hpi(No);
    // inputs may be host or device pointers; plain host memory (the usual case) is copied by the CPU, not through the runtime
    auto copy_in = [](void *dst, const void *src, size_t bytes) {
      hipPointerAttribute_t at{};
      const hipError_t e = hipPointerGetAttributes(&at, src);
      if (e != hipSuccess) (void)hipGetLastError(); // memory the runtime has never seen: pageable host memory
      if (e != hipSuccess || at.type == hipMemoryTypeHost || at.type == hipMemoryTypeUnregistered) std::memcpy(dst, src, bytes);
      else GR_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
    };
    if (!model) {
      copy_in(hc.data(), c, hc.size() * sizeof(T));
      copy_in(hp.data(), p, hp.size() * sizeof(T));
      copy_in(ho.data(), o, ho.size() * sizeof(T));
    }
    copy_in(hci.data(), ci, No * sizeof(int32_t));
    copy_in(hpi.data(), pi, No * sizeof(int32_t));
    // Internal point order = sorted by the first (lowest) camera that observes the point, so that
    // the observations of one camera (cm order, sorted by point) gather from a narrow range of
    // points: the 24-byte point / direction gathers then share cache lines between neighbouring
    // lanes (PMC: 2.6-5x over-fetch with the user's order).  The API keeps the user's order.
    {
      std::vector<int> first(Np, std::numeric_limits<int>::max());
      for (int64_t i = 0; i < No; ++i)
        if (hpi[i] >= 0 && hpi[i] < Np) first[hpi[i]] = std::min(first[hpi[i]], (int)hci[i]);
      // stable by (first camera, old index): a counting sort over the Nc + 1 keys (a point nobody observes sorts last) —
      // the same order std::stable_sort gave, without its 12 ms on Ladybug-1723
      std::vector<int> bucket(Nc + 2, 0);
      auto key = [&](int64_t l) { return first[l] == std::numeric_limits<int>::max() ? (int)Nc : first[l]; };
      for (int64_t l = 0; l < Np; ++l) bucket[key(l) + 1]++;
      for (int64_t c = 0; c <= Nc; ++c) bucket[c + 1] += bucket[c];
      h_pt_new2old.resize(Np);
      for (int64_t l = 0; l < Np; ++l) h_pt_new2old[bucket[key(l)]++] = (int)l;
      h_pt_old2new.resize(Np);
      for (int64_t q = 0; q < Np; ++q) h_pt_old2new[h_pt_new2old[q]] = (int)q;
      for (int64_t i = 0; i < No; ++i)
        if (hpi[i] >= 0 && hpi[i] < Np) hpi[i] = h_pt_old2new[hpi[i]];
      std::vector<T> hp2(hp.size());
      for (int64_t q = 0; q < Np; ++q)
        for (int k = 0; k < 3; ++k) hp2[3 * q + k] = hp[3 * (size_t)h_pt_new2old[q] + k];
      hp.swap(hp2);
    }
    lap("inputs to the host + point order by first camera");
    cams.upload(hc, stream);
    pts.upload(hp, stream);
    lap("cams / pts upload");
    build_orderings(hci, hpi, ho);
    lap("build_orderings (two counting sorts, segments, uploads)");
    cams_bak.alloc(hc.size());
    pts_bak.alloc(hp.size());
    pack.alloc(PACK * (size_t)Nc);
    Hcc.alloc(81 * (size_t)Nc); Hll.alloc(9 * (size_t)Np);
    bu.alloc(n); bc.p = bu.p; bl.p = bu.p + pose_dim;
    scales.alloc(n);
    lap("Hcc, Hll, b, scales, pack, backups");
    nb_pm = cdiv(No, TPB);
    {
      int cus = 0; // (hipGetDeviceProperties fills the whole property struct: 25 ms on this stack; one attribute is what is needed)
      GR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      num_cu = cus > 0 ? cus : 256;
    }
    lap("vertex uploads + first arrays + device properties");
    apply_tuning(); // grid_obs, grid_vec
    lap("apply_tuning (occupancy queries)");
    grid_chi2 = std::max(8, std::min(nb_pm, num_cu * 2) & ~7); // measured: the light chi2 pass prefers 2 long blocks per CU (16 vs 22 us at 8)
    n_chi2_blocks = std::min(cdiv(No, TPB), 1024);
    chi2_partial.alloc(std::max<size_t>(nb_pm, 2 * (size_t)cdiv(std::max<size_t>(No, n), TPB)) + 64);
    dscalars.alloc(4);
    ticket.alloc(1 + TICKET_GROUPS); ticket.zero(stream);
    cam_partial.alloc(54 * (size_t)nseg); op_partial.alloc(9 * (size_t)nseg); part9.alloc(9 * (size_t)nch);
    grid_partial.alloc(2 * (size_t)std::max(cdiv(std::max<size_t>(No, n), TPB), cdiv(n, 252) + cdiv(Np, TPB)) + 64);
    pcg_iters.alloc(1);
    alloc_pinned(64);
    g9.alloc(8 * (size_t)No); g3.alloc(3 * (size_t)No);
    v_dx.alloc(n);
    GR_HIP(hipStreamSynchronize(stream));
    lap("vertex uploads + working arrays");
  }

  // persistent grids and the per-problem choices that depend on the tuning; called by the constructor and by gr_bal_set_tuning
  void apply_tuning() override {
    grid_obs = std::max(8, std::min(nb_pm, num_cu * (tune.grid_mult > 0 ? tune.grid_mult : 4)) & ~7); // multiple of 8: one contiguous tile range per XCD
    // The two heaviest persistent kernels get EXACTLY the workgroups that fit at once (occupancy query), unless grid_mult forces
    // a count: k_linearize is built for 3 workgroups per CU, and 4 per CU (the old common grid) left a quarter of them waiting
    // for a free slot — Ladybug-1723 fp64: 38.4 -> 33.8 us (grid x 1 41.5, x 2 36.8, x 3 33.8, x 4 38.4); operator 21.5 -> 20.9
    auto resident = [&](const void *fn, int dflt, int cap) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, TPB, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = dflt; }
      const int mult = tune.grid_mult > 0 ? tune.grid_mult : std::min(nb, cap);
      // fewer tiles than resident slots: one tile per workgroup, rounded UP to the XCD count (rounded down, 8 workgroups of a
      // 125-tile problem walked two tiles each: a second dependent load chain in a kernel that is nothing but such a chain)
      if (nb_pm <= num_cu * mult) return std::max(8, (nb_pm + 7) & ~7);
      return std::max(8, (num_cu * mult) & ~7);
    };
    grid_lin = resident(reinterpret_cast<const void *>(&k_linearize<T, false>), LIN_WAVES, 8);
    if (model) { // the user-side lineariser's own residency (gr_model_ops.lin_wg_per_cu, from its occupancy query)
      const int mult = tune.grid_mult > 0 ? tune.grid_mult : std::max(1, std::min(model->lin_wg_per_cu > 0 ? model->lin_wg_per_cu : 3, 8));
      grid_lin = std::max(8, std::min(nb_pm, num_cu * mult) & ~7);
    }
    {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_finalize_bj<T>), TPB, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 3; }
      fbj_per_cu = env_int("GR_FBJ_PER_CU", 0) > 0 ? env_int("GR_FBJ_PER_CU", 0) : std::min(nb, 8);
    }
    // the operator prefers 3 (fp64) / 4 (fp32) workgroups per CU even where more would fit (Venice-1778 fp32: 83.6 us at 4, 89.7 at its occupancy)
    grid_op = resident(reinterpret_cast<const void *>(&k_pcg_operator<T, 0, T>), 3, sizeof(T) == 8 ? 3 : 4);
    if (model) {
      int mult = tune.grid_mult;
      if (mult <= 0) {
        if (model->store_jacobians) {
          int nb = 0;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_pcg_operator_stored<T, T, 9>), TPB, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 4; }
          mult = std::min(nb, 8);
        } else mult = std::max(1, std::min(model->op_wg_per_cu > 0 ? model->op_wg_per_cu : 3, 8));
      }
      grid_op = std::max(8, std::min(nb_pm, num_cu * mult) & ~7);
    }
    // light vector kernels: several elements per thread (every wave first re-derives the loop scalars from the
    // dot-product slots, so one element per thread made that prologue most of the kernel)
    {
      // grid-stride kernels: no more workgroups than are resident at once (k_pcg_direction takes 69 VGPRs in fp64 = 7 per CU; the 8
      // assumed before left Final-13682's 2 048-workgroup launches with a second, nearly empty round)
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_pcg_direction<T>), TPB, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 4; }
      grid_vec = std::max(1, std::min(cdiv(n, (size_t)TPB * std::max(1, tune.vec_per_thread)), num_cu * std::min(nb, 8)));
    }
    if (tiling_tuned) { if (tiled) untile(); tiling_tuned = false; } // timed choices are re-made by the next solver_update_structure
    fused_agreed = false; // ... before the ranks agree on the fused message again (agree_on_fusion, collective): the timing runs must not fuse
    records_tuned = false;
    rp_fit = -1;
    if (chol_ready) { chol_ready = false; use_spchol = false; }
    if (schur_ready && nitems) schur_ready = false;
    if (comm) comm->set_timeout_ms(tune.ipc_timeout_ms);
  }
  void alloc_pinned(int flag_cap) {
    if (h_res && flag_cap <= h_flag_cap) return;
    if (h_res) (void)hipHostFree(h_res);
    void *p = nullptr;
    GR_HIP(hipHostMalloc(&p, 64 + sizeof(int) * 2 * (size_t)(flag_cap + 16), hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(p, 0, 64 + sizeof(int) * 2 * (size_t)(flag_cap + 16));
    h_res = static_cast<double *>(p);
    h_seq = reinterpret_cast<volatile int *>(static_cast<char *>(p) + 32);
    h_flag = reinterpret_cast<volatile int *>(static_cast<char *>(p) + 64);
    h_flag_cap = flag_cap + 16;
  }
  ~Engine() override {
    if (h_res) (void)hipHostFree(h_res);
    if (h_ts) (void)hipHostFree(h_ts);
  }
  // spin on a pinned word written by a kernel (system-scope fence on the device side)
  template <typename Pred> void spin_until(Pred pred) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0;; ++it) {
      if (pred()) return;
      // on landmark shards a flag that does not come may be a peer that does not: surface the transport's failure as CommError
      // (and stop enqueueing collectives behind it) as soon as its error word is up, not after the generic 20 s
      if (comm && (it & 0x3FF) == 0x3FF) check_comm("PCG loop");
      if ((it & 0xFFFF) == 0xFFFF && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) {
        GR_HIP(hipStreamSynchronize(stream)); // surfaces a kernel fault instead of hanging
        if (pred()) return;
        throw HipError("timeout waiting for a device flag");
      }
    }
  }

  // ---- symbolic phase -----------------------------------------------------------
  // Replaces the host hash-map walks of factor.hpp:458-461 / 731-758: two counting
  // sorts of the observation list (by point, by camera).
  void build_orderings(const std::vector<int32_t> &ci, const std::vector<int32_t> &pi, const std::vector<T> &ho) {
    Laps lap(tune.verbose != 0, "build_orderings");
    for (int64_t o = 0; o < No; ++o)
      if (ci[o] < 0 || ci[o] >= Nc || pi[o] < 0 || pi[o] >= Np) throw std::invalid_argument("observation index out of range");
    h_pt_ptr.assign(Np + 1, 0);
    for (int64_t o = 0; o < No; ++o) h_pt_ptr[pi[o] + 1]++;
    for (int64_t l = 0; l < Np; ++l) h_pt_ptr[l + 1] += h_pt_ptr[l];
    std::vector<int> pm_obs(No), w(h_pt_ptr.begin(), h_pt_ptr.end() - 1);
    for (int64_t o = 0; o < No; ++o) pm_obs[w[pi[o]]++] = (int)o;
    const int nth = host_threads();
    {
      const auto before = [&](int a, int b) { return ci[a] != ci[b] ? ci[a] < ci[b] : a < b; };
      par_chunks((size_t)Np, nth, [&](size_t l0, size_t l1, int) {
        for (size_t l = l0; l < l1; ++l) {
          int *lo = pm_obs.data() + h_pt_ptr[l], *hi = pm_obs.data() + h_pt_ptr[l + 1];
          if (hi - lo > 16) { std::sort(lo, hi, before); continue; }
          for (int *q = lo + 1; q < hi; ++q) { // a handful of observations per point: insertion sort
            const int v = *q;
            int *r = q;
            for (; r > lo && before(v, r[-1]); --r) *r = r[-1];
            *r = v;
          }
        }
      });
    }
    lap("point-major counting sort + per-point camera order");
    h_cam_pm.resize(No); h_pt_pm.resize(No); h_pm_of_orig.resize(No);
    const bool with_obs = !ho.empty();
    std::vector<T> h_obs_pm(with_obs ? 2 * No : 0);
    // per-thread camera histograms of the pm order: the camera-major order below is a stable counting sort by camera whose
    // chunks scatter side by side, each from its own offsets
    std::vector<std::vector<int>> hist(nth, std::vector<int>(Nc + 1, 0));
    par_chunks((size_t)No, nth, [&](size_t a0, size_t a1, int k) {
      std::vector<int> &hk = hist[k];
      for (size_t a = a0; a < a1; ++a) {
        const int o = pm_obs[a];
        h_cam_pm[a] = ci[o]; h_pt_pm[a] = pi[o]; h_pm_of_orig[o] = (int)a;
        if (with_obs) { h_obs_pm[2 * a] = ho[2 * (size_t)o]; h_obs_pm[2 * a + 1] = ho[2 * (size_t)o + 1]; }
        hk[ci[o]]++;
      }
    });
    par_chunks((size_t)Np, nth, [&](size_t l0, size_t l1, int) {
      for (size_t l = l0; l < l1; ++l)
        for (int a = h_pt_ptr[l] + 1; a < h_pt_ptr[l + 1]; ++a)
          if (h_cam_pm[a] == h_cam_pm[a - 1]) throw std::domain_error("duplicate (camera, point) edge");
    });
    h_cam_ptr.assign(Nc + 1, 0);
    for (int64_t c = 0; c < Nc; ++c) {
      int tot = 0;
      for (int k = 0; k < nth; ++k) tot += hist[k][c];
      h_cam_ptr[c + 1] = tot;
    }
    for (int64_t c = 0; c < Nc && !shard; ++c) if (h_cam_ptr[c + 1] == 0) throw std::invalid_argument("camera without observations (the reference deactivates it, graph.hpp:171; remove it from the problem)");
    for (int64_t l = 0; l < Np; ++l) if (h_pt_ptr[l + 1] == h_pt_ptr[l]) throw std::invalid_argument("point without observations (the reference deactivates it, graph.hpp:171; remove it from the problem)");
    for (int64_t c = 0; c < Nc; ++c) h_cam_ptr[c + 1] += h_cam_ptr[c];
    // hist[k][c] -> first camera-major slot of chunk k's observations of camera c
    for (int64_t c = 0; c < Nc; ++c) {
      int at = h_cam_ptr[c];
      for (int k = 0; k < nth; ++k) { const int cnt = hist[k][c]; hist[k][c] = at; at += cnt; }
    }
    h_pos_cm.resize(No); h_pt_cm.resize(No);
    std::vector<int> h_cam_cm(No);
    std::vector<T> h_obs_cm(with_obs ? 2 * No : 0);
    par_chunks((size_t)No, nth, [&](size_t a0, size_t a1, int k) {
      std::vector<int> &wc = hist[k];
      for (size_t a = a0; a < a1; ++a) {
        const int j = wc[h_cam_pm[a]]++;
        h_pos_cm[j] = (int)a; h_pt_cm[j] = h_pt_pm[a]; h_cam_cm[j] = h_cam_pm[a];
        if (with_obs) { h_obs_cm[2 * (size_t)j] = h_obs_pm[2 * a]; h_obs_cm[2 * (size_t)j + 1] = h_obs_pm[2 * a + 1]; }
      }
    });
    // camera-major chunks: <= CHUNK consecutive observations of one camera per wave
    h_chunk_cam.clear(); h_chunk_beg.clear(); h_cam_chunk_ptr.assign(Nc + 1, 0);
    for (int64_t c = 0; c < Nc; ++c) {
      for (int j = h_cam_ptr[c]; j < h_cam_ptr[c + 1]; j += CHUNK) { h_chunk_cam.push_back((int)c); h_chunk_beg.push_back(j); }
      h_cam_chunk_ptr[c + 1] = (int)h_chunk_cam.size();
    }
    nch = (int)h_chunk_cam.size();
    h_chunk_beg.push_back((int)No);
    lap("pm arrays, duplicate check, camera-major order, chunks");
    build_segments(h_cam_cm, /*ptiles=*/0);
    lap("segments");
    chunk_cam.upload(h_chunk_cam, stream); chunk_beg.upload(h_chunk_beg, stream); cam_chunk_ptr.upload(h_cam_chunk_ptr, stream);
    pt_ptr.upload(h_pt_ptr, stream); cam_pm.upload(h_cam_pm, stream); pt_pm.upload(h_pt_pm, stream);
    cam_ptr.upload(h_cam_ptr, stream); pt_cm.upload(h_pt_cm, stream); pos_cm.upload(h_pos_cm, stream); cam_cm.upload(h_cam_cm, stream);
    obs_pm.upload(h_obs_pm, stream); obs_cm.upload(h_obs_cm, stream);
    GR_HIP(hipStreamSynchronize(stream));
    lap("index / observation uploads");
  }

  // (64-observation block, camera) segments of an observation order given by its camera stream
  void build_segments(const std::vector<int> &cam_of, int ptiles) {
    const int64_t nblk = (No + 63) / 64;
    std::vector<int> h_blk_seg(4 * (size_t)cdiv(No, TPB) + 8, 0), seg_cam; // padded: tail waves of the last tile read their entry too
    for (int64_t b = 0; b < nblk; ++b) {
      h_blk_seg[b] = (int)seg_cam.size();
      const int64_t j0 = 64 * b, j1 = std::min<int64_t>(No, j0 + 64);
      // one segment per DISTINCT camera of the block, in order of first appearance: exactly what the kernels' leader loop
      // does (a block that crosses a point-tile boundary can hold two runs of one camera: they are one reduction)
      const size_t first = seg_cam.size();
      for (int64_t j = j0; j < j1; ++j) {
        bool seen = false;
        for (size_t q = first; q < seg_cam.size() && !seen; ++q) seen = seg_cam[q] == cam_of[j];
        if (!seen) seg_cam.push_back(cam_of[j]);
      }
    }
    nseg = (int)seg_cam.size();
    h_cam_seg_ptr.assign(Nc + 1, 0);
    for (int c : seg_cam) h_cam_seg_ptr[c + 1]++;
    for (int64_t c = 0; c < Nc; ++c) h_cam_seg_ptr[c + 1] += h_cam_seg_ptr[c];
    std::vector<int> wpos(h_cam_seg_ptr.begin(), h_cam_seg_ptr.end() - 1), h_seg_slot(std::max(nseg, 1));
    for (int q = 0; q < nseg; ++q) h_seg_slot[q] = wpos[seg_cam[q]]++; // flat order inside a camera = ascending blocks: a fixed order
    cam_seg_ptr.upload(h_cam_seg_ptr, stream); blk_seg.upload(h_blk_seg, stream); seg_slot.upload(h_seg_slot, stream);
    n_ptiles = ptiles;
  }
  // Point-tiled observation order for graphs whose point gathers miss L2 (see the members above).  K point tiles
  // (a multiple of 8) of equal observation count; order key (tile % 8, tile / 8, camera, point): XCD x's eighth of the
  // list = its own tiles, one after the other.
  void build_tiled_order(int K) {
    std::vector<int> tile_of_pt(Np);
    for (int64_t l = 0; l < Np; ++l) tile_of_pt[l] = (int)std::min<int64_t>(K - 1, (int64_t)h_pt_ptr[l] * K / No);
    auto key_tile = [&](int t) { return (t % 8) * ((K + 7) / 8) + t / 8; };
    const int KT = 8 * ((K + 7) / 8);
    // stable counting sort of the pm order (point, camera) by (tile key, camera): points ascend inside a (tile, camera) run
    std::vector<int64_t> cnt((size_t)KT * Nc + 1, 0);
    for (int64_t a = 0; a < No; ++a) cnt[(size_t)key_tile(tile_of_pt[h_pt_pm[a]]) * Nc + h_cam_pm[a] + 1]++;
    for (size_t q = 0; q + 1 < cnt.size(); ++q) cnt[q + 1] += cnt[q];
    std::vector<int> h_cam(No), h_pt(No), h_pos(No);
    std::vector<T> h_obs(2 * (size_t)No), obs_pm_h = obs_pm.download(stream);
    for (int64_t a = 0; a < No; ++a) {
      const int64_t j = cnt[(size_t)key_tile(tile_of_pt[h_pt_pm[a]]) * Nc + h_cam_pm[a]]++;
      h_cam[j] = h_cam_pm[a]; h_pt[j] = h_pt_pm[a]; h_pos[j] = (int)a;
      h_obs[2 * (size_t)j] = obs_pm_h[2 * (size_t)a]; h_obs[2 * (size_t)j + 1] = obs_pm_h[2 * (size_t)a + 1];
    }
    t_cam.upload(h_cam, stream); t_pt.upload(h_pt, stream); t_pos.upload(h_pos, stream); t_obs.upload(h_obs, stream);
    build_segments(h_cam, K);
    cam_partial.alloc(54 * (size_t)nseg); op_partial.alloc(9 * (size_t)nseg);
    tiled = true;
    g3_obs_order = false;
    GR_HIP(hipStreamSynchronize(stream));
  }
  void untile() {
    if (!tiled) return;
    tiled = false;
    g3_obs_order = false;
    std::vector<int> h_cam_cm = cam_cm.download(stream);
    build_segments(h_cam_cm, 0);
    cam_partial.alloc(54 * (size_t)nseg); op_partial.alloc(9 * (size_t)nseg);
  }
  // g3 in observation order + XCD-matched point sweep of the update kernel (G3Gather), for the CURRENT observation order.
  // Point tiles: the tiled order's own; in the plain order 8 ranges of equal observation count (XCD x walks the x-th eighth
  // of the camera-major observations, and points are numbered by first camera, so its points are mostly the x-th range).
  void build_g3_gather() {
    std::vector<int> pos = (tiled ? t_pos : pos_cm).download(stream), gidx(No);
    for (int64_t j = 0; j < No; ++j) gidx[pos[j]] = (int)j;
    const int K = tiled ? n_ptiles : 8;
    std::vector<int> pp(K + 1, (int)Np);
    pp[0] = 0;
    for (int64_t l = 0, t = 0; l < Np; ++l) {
      const int tl = (int)std::min<int64_t>(K - 1, (int64_t)h_pt_ptr[l] * K / No);
      while (t < tl) pp[++t] = (int)l;
    }
    g3_gidx.upload(gidx, stream); ptile_ptr.upload(pp, stream);
    g3_ptiles = K;
    g3_obs_order = true;
    GR_HIP(hipStreamSynchronize(stream));
  }
  // decided once per problem, by timing: GR_PTILES = number of point tiles (0 = plain order) forces it
  bool tiling_tuned = false;
  bool tiling_for_pcg = false; // the solver being set up is the matrix-free PCG (solver_update_structure)
  void tune_tiling() {
    if (tiling_tuned) return;
    tiling_tuned = true;
    if (model) return; // user-traits problems keep the plain camera-major order (their per-observation streams are laid out in it once)
    tune_tiling_order();
    const size_t per_point = (size_t)(6 * sizeof(T) + 3 * sizeof(T) * (double)No / (double)Np);
    if ((size_t)Np * per_point >= ((size_t)24 << 20)) tune_g3_order();
  }
  void tune_tiling_order() {
    // landmark shards that can fuse the inner iteration's message keep the plain order: the fused form needs it (one rank that tiled
    // would un-fuse every rank), and a shard cut by camera locality (dist.py) has the camera runs of the unsharded problem
    // (only for the solver that can fuse: the matrix-free PCG in its single-reduction form with a message that fits the slot;
    // the implicit-Schur PCG and the two-reduction form keep the timed choice)
    if (comm && ipc_comm() && tune.point_tiles < 0 && tiling_for_pcg && pcg_cg() && shard_fused_local()) return;
    const size_t per_point = (size_t)(6 * sizeof(T) + 3 * sizeof(T) * (double)No / (double)Np); // X + direction + this point's g3 slots
    int K = 0;
    if (tune.point_tiles >= 0) K = tune.point_tiles;
    else {
      if ((size_t)Np * per_point < ((size_t)24 << 20)) return; // the whole working set sits in the L2s as it is
      // tiles small enough for an XCD's L2 (3 MB of points + scatter slots each), but not so many that a camera's run inside
      // a tile drops below ~64 observations (a wave would straddle several cameras; measured on Final-13682: 304 tiles run
      // the linearisation 2x slower than 32)
      const int k_l2 = cdiv((size_t)Np * per_point, (size_t)(3u << 20)), k_run = (int)std::max<int64_t>(8, No / std::max<int64_t>(1, Nc * 64));
      K = std::min(k_l2, k_run);
      // a multiple of 8 (one share per XCD): rounded UP when the L2 asks for the tiles, DOWN when the run length caps them
      // (Final-13682: 33 -> 40 tiles left 53-observation runs, every wave straddling two cameras; 32 keeps 66)
      K = k_run < k_l2 ? std::max(8, K / 8 * 8) : std::max(8, (K + 7) / 8 * 8);
    }
    if (K <= 0) return;
    K = std::max(8, (K + 7) / 8 * 8);
    const bool forced = tune.point_tiles >= 0;
    const double t_plain = forced ? 0.0 : diag_time(0, 0, 5) + diag_time(1, 0, 5) + diag_time(3, 0, 5);
    build_tiled_order(K);
    if (!forced) {
      const double t_tiled = diag_time(0, 0, 5) + diag_time(1, 0, 5) + diag_time(3, 0, 5);
      if (tune.verbose) std::fprintf(stderr, "[graphite-mi355x] operator + linearise + update: %.1f us plain order, %.1f us with %d point tiles -> %s\n", t_plain, t_tiled, K, t_tiled < 0.95 * t_plain ? "tiled" : "plain");
      if (!(t_tiled < 0.95 * t_plain)) untile();
    }
  }
  // Plain camera-major order on a graph whose g3 does not stay in the L2s (a landmark shard of Final-13682: 87 MB): the operator's
  // scattered 24-byte stores against whole lines + a gather in the update launch, timed both ways (one shard of Final-13682, fp64,
  // records: operator 137 -> 92 us, update 45 -> 72 us).  Rank-local: the sums keep their order, results do not depend on it.
  void tune_g3_order() {
    g3_plain_pays = false;
    if (tiled || tune.g3_gather >= 0) return;
    const double t_pm = diag_time(0, 0, 5) + diag_time(3, 0, 5);
    build_g3_gather();
    const double t_obs = diag_time(0, 0, 5) + diag_time(3, 0, 5);
    g3_plain_pays = t_obs < 0.95 * t_pm;
    if (tune.verbose) std::fprintf(stderr, "[graphite-mi355x] operator + update: %.1f us with g3 in point-major order, %.1f us in observation order + gather -> %s\n", t_pm, t_obs, g3_plain_pays ? "observation order" : "point-major");
    if (!g3_plain_pays) g3_obs_order = false;
  }

  // SchurComplement::build_structure (schur.hpp:194-225).  The reference walks
  // Pi tuples with two hash lookups each on the host (:556-580); here the block
  // map is a dense Nc x Nc table (Nc <= 16384) and the product list is counting-
  // sorted by destination block.
  void build_schur_structure() {
    if (schur_ready) return;
    if (Nc > 16384) throw std::invalid_argument("explicit Schur complement supports at most 16384 cameras");
    std::vector<int> map((size_t)Nc * Nc, -1); // map[j*Nc+i], i<=j
    for (int64_t c = 0; c < Nc; ++c) map[(size_t)c * Nc + c] = 0;
    int64_t np = 0;
    for (int64_t l = 0; l < Np; ++l) {
      const int64_t k = h_pt_ptr[l + 1] - h_pt_ptr[l];
      np += k * (k + 1) / 2;
      for (int a = h_pt_ptr[l]; a < h_pt_ptr[l + 1]; ++a)
        for (int b = a; b < h_pt_ptr[l + 1]; ++b) map[(size_t)h_cam_pm[b] * Nc + h_cam_pm[a]] = 0;
    }
    // rank-local condition: with a communicator it is agreed on through the flags all-reduce below, so that
    // either every rank throws or none does (a rank that left early would leave its peers in the collective)
    bool too_many = np >= (int64_t)std::numeric_limits<int>::max();
    if (too_many && !comm) throw std::invalid_argument("too many Schur products for 32-bit indices");
    nprod = np;
    if (comm) {
      // Landmark shards: every rank holds the products of its own points but the block list must be the
      // union over all ranks (S is all-reduced).  Presence flags travel 6 x 8 bits per double: integer
      // sums below 2^53 are exact in the all-reduce, one flag counts at most `size` <= 255.
      if (comm->size > 255) throw std::invalid_argument("explicit Schur complement: at most 255 landmark shards");
      const size_t ntri = (size_t)Nc * (Nc + 1) / 2;
      std::vector<double> packed((ntri + 5) / 6, 0.0);
      for (int64_t j = 0; j < Nc; ++j)
        for (int64_t i = 0; i <= j; ++i)
          if (map[(size_t)j * Nc + i] == 0) {
            const size_t idx = (size_t)j * (j + 1) / 2 + i;
            packed[idx / 6] += (double)(1ull << (8 * (idx % 6)));
          }
      packed.push_back(too_many ? 1.0 : 0.0);
      DevBuf<double> flags;
      flags.upload(packed, stream);
      allreduce_d(flags.p, packed.size());
      packed = flags.download(stream);
      check_comm("build_schur_structure"); // a timed-out rank would build another block list than its peers
      too_many = packed.back() != 0.0;
      packed.pop_back();
      if (too_many) throw std::invalid_argument("too many Schur products for 32-bit indices (on at least one landmark shard)");
      for (int64_t j = 0; j < Nc; ++j)
        for (int64_t i = 0; i <= j; ++i) {
          const size_t idx = (size_t)j * (j + 1) / 2 + i;
          if (((unsigned long long)packed[idx / 6] >> (8 * (idx % 6))) & 255ull) map[(size_t)j * Nc + i] = 0;
        }
    }
    h_S_colptr.assign(Nc + 1, 0); h_S_rowi.clear(); h_S_coli.clear(); h_S_diag.assign(Nc, -1);
    for (int64_t j = 0; j < Nc; ++j) {
      for (int64_t i = 0; i <= j; ++i)
        if (map[(size_t)j * Nc + i] == 0) {
          map[(size_t)j * Nc + i] = (int)h_S_rowi.size();
          if (i == j) h_S_diag[j] = (int)h_S_rowi.size();
          h_S_rowi.push_back((int)i); h_S_coli.push_back((int)j);
        }
      h_S_colptr[j + 1] = (int)h_S_rowi.size();
    }
    nnzb = (int64_t)h_S_rowi.size();
    std::vector<int> h_prod_ptr(nnzb + 1, 0);
    for (int64_t l = 0; l < Np; ++l)
      for (int a = h_pt_ptr[l]; a < h_pt_ptr[l + 1]; ++a)
        for (int b = a; b < h_pt_ptr[l + 1]; ++b) h_prod_ptr[map[(size_t)h_cam_pm[b] * Nc + h_cam_pm[a]] + 1]++;
    for (int64_t q = 0; q < nnzb; ++q) h_prod_ptr[q + 1] += h_prod_ptr[q];
    std::vector<int> h_prod_a(nprod), h_prod_b(nprod), h_prod_pm(nprod), w(h_prod_ptr.begin(), h_prod_ptr.end() - 1);
    for (int64_t l = 0; l < Np; ++l)
      for (int a = h_pt_ptr[l]; a < h_pt_ptr[l + 1]; ++a)
        for (int b = a; b < h_pt_ptr[l + 1]; ++b) {
          const int q = w[map[(size_t)h_cam_pm[b] * Nc + h_cam_pm[a]]]++;
          h_prod_a[q] = a; h_prod_b[q] = b; h_prod_pm[q] = (int)l;
        }
    // work items: <= `isz` products of one block per wave (7 lane groups, so a multiple of 7).  Measured on
    // Ladybug-49 (82 K products over 1 225 blocks): 7 / 14 / 28 / 56 / 84 / 112 / 168 / 336 products per item ->
    // 118 / 60 / 34 / 23.3 / 23.5 / 27 / 35 / 53 us: smaller items turn every block into a multi-item block (81 atomics
    // per item), larger ones lengthen the serial product loop of a wave.  Per-item partials in a scratch array summed by
    // the finishing pass instead of atomics: 19 us at 56, 17 us at 7..28, but the heavier finishing pass gives it back
    // (8 765 vs 8 839 LM it/s): the kernel's floor is its chain of dependent index loads, not the atomics.
    int isz = 56;
    if (tune.schur_item > 0) isz = std::max(7, tune.schur_item / 7 * 7);
    std::vector<int> h_item_blk, h_item_beg, h_item_end, h_item_single, h_multi;
    for (int64_t q = 0; q < nnzb; ++q) {
      const int beg = h_prod_ptr[q], end = h_prod_ptr[q + 1];
      const bool single = (end - beg) <= isz;
      if (!single) h_multi.push_back((int)q);
      int b0 = beg;
      do {
        const int e0 = std::min(b0 + isz, end);
        h_item_blk.push_back((int)q); h_item_beg.push_back(b0); h_item_end.push_back(e0); h_item_single.push_back(single ? 1 : 0);
        b0 = e0;
      } while (b0 < end);
    }
    nitems = (int)h_item_blk.size(); nmulti = (int)h_multi.size();
    { // k_schur_reduce: the items of a multi-item block are consecutive; its last arriver adds their slabs in item order
      std::vector<int> h_item_multi(nitems, -1), h_multi_first(std::max(nmulti, 1), 0), h_multi_n(std::max(nmulti, 1), 0);
      int m = -1;
      for (int it = 0; it < nitems; ++it) {
        if (h_item_single[it]) continue;
        if (m < 0 || h_multi[m] != h_item_blk[it]) { ++m; h_multi_first[m] = it; }
        h_item_multi[it] = m; h_multi_n[m]++;
      }
      item_multi.upload(h_item_multi, stream); multi_first.upload(h_multi_first, stream); multi_n.upload(h_multi_n, stream);
      multi_cnt.alloc(std::max(nmulti, 1)); multi_cnt.zero(stream);
      slab.alloc(81 * (size_t)std::max(nitems, 1));
    }
    // row lists for y = S x: upper blocks of row i, then lower blocks as transposes (~blk)
    std::vector<int> h_row_ptr(Nc + 1, 0);
    for (int64_t q = 0; q < nnzb; ++q) {
      h_row_ptr[h_S_rowi[q] + 1]++;
      if (h_S_rowi[q] != h_S_coli[q]) h_row_ptr[h_S_coli[q] + 1]++;
    }
    for (int64_t c = 0; c < Nc; ++c) h_row_ptr[c + 1] += h_row_ptr[c];
    std::vector<int> h_row_blk(h_row_ptr[Nc]), h_row_col(h_row_ptr[Nc]), wr(h_row_ptr.begin(), h_row_ptr.end() - 1);
    for (int64_t q = 0; q < nnzb; ++q) {
      const int i = h_S_rowi[q], j = h_S_coli[q];
      int e = wr[i]++;
      h_row_blk[e] = (int)q; h_row_col[e] = j;
      if (i != j) { e = wr[j]++; h_row_blk[e] = ~(int)q; h_row_col[e] = i; }
    }
    item_blk.upload(h_item_blk, stream); item_beg.upload(h_item_beg, stream); item_end.upload(h_item_end, stream);
    item_single.upload(h_item_single, stream); multi_blk.upload(h_multi, stream);
    prod_a.upload(h_prod_a, stream); prod_b.upload(h_prod_b, stream); prod_pm.upload(h_prod_pm, stream);
    S_rowi.upload(h_S_rowi, stream); S_coli.upload(h_S_coli, stream); S_diag.upload(h_S_diag, stream);
    row_ptr.upload(h_row_ptr, stream); row_blk.upload(h_row_blk, stream); row_col.upload(h_row_col, stream);
    S.alloc(81 * (size_t)nnzb); b_schur.alloc(pose_dim);
    Hll_inv.alloc(9 * (size_t)Np); Mp.alloc(9 * (size_t)Np); vl.alloc(3 * (size_t)Np);
    MinvS.alloc(81 * (size_t)Nc);
    Hcp.alloc(27 * (size_t)No);
    xl_acc.alloc(3 * (size_t)Np); xl_acc.zero(stream);
    boundary_flag.alloc(Np); boundary_flag.zero(stream);
    v_r.alloc(n); v_p.alloc(n); v_z.alloc(n); v_Ap.alloc(n); v_xb.alloc(n);
    GR_HIP(hipStreamSynchronize(stream));
    schur_ready = true;
  }

  // ---- profiling ------------------------------------------------------------------
  // A scope's two events ride ON the first launch() inside it (hipExtLaunchKernelGGL start / stop events: the
  // dispatch packet's own begin / end stamps, what rocprofv3 reports) — no marker packets between kernels, which
  // cost a ~4-6 us bubble each.  Scopes whose body goes through no launch() fall back to recording around it.
  hipEvent_t ext_a = nullptr, ext_b = nullptr;
  struct Scope {
    Engine *e; KernelProf *kp = nullptr; hipEvent_t a{}, b{}; bool ext;
    Scope(Engine *e_, const char *name, double bytes, double flops, bool ext_ = false) : e(e_), ext(ext_) {
      if (!e->profiling) return;
      kp = &e->prof[name];
      kp->name = name; kp->bytes += bytes; kp->flops += flops; kp->scoped++;
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      if (ext) { e->ext_a = a; e->ext_b = b; }
      else (void)hipEventRecord(a, e->stream);
    }
    ~Scope() {
      if (!kp) return;
      if (ext && e->ext_a == a) { e->ext_a = e->ext_b = nullptr; (void)hipEventRecord(a, e->stream); ext = false; } // nothing launched
      if (!ext) (void)hipEventRecord(b, e->stream);
      kp->pending.emplace_back(a, b);
    }
  };
  int64_t launch_count = 0; // kernels of the per-iteration paths (launch()) + the sharded form's k_cam_rows / mailbox kernels
  template <typename K, typename... A> void launch(K kernel, size_t grid, A... args) {
    ++launch_count;
    if (ext_a) {
      hipEvent_t a = ext_a, b = ext_b;
      ext_a = ext_b = nullptr;
      hipExtLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(TPB), 0, stream, a, b, 0, args...);
    } else hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(TPB), 0, stream, args...);
  }
  void flush_prof() {
    for (auto &it : prof) {
      for (auto &ev : it.second.pending) {
        (void)hipEventSynchronize(ev.second);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ev.first, ev.second);
        it.second.total_ms += ms; it.second.launches++;
        (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second);
      }
      it.second.pending.clear();
    }
  }
  int kernel_stats(gr_kernel_stat *out, int cap) override {
    flush_prof();
    int k = 0;
    for (auto &it : prof) {
      if (k < cap) {
        std::memset(&out[k], 0, sizeof(gr_kernel_stat));
        std::strncpy(out[k].name, it.second.name.c_str(), sizeof(out[k].name) - 1);
        out[k].launches = it.second.launches; out[k].total_ms = it.second.total_ms;
        const double sc = (double)std::max<int64_t>(it.second.scoped, 1);
        out[k].bytes_per_launch = it.second.bytes / sc; out[k].flops_per_launch = it.second.flops / sc;
        out[k].active_launches = it.second.launches - it.second.noop;
      }
      ++k;
    }
    return k;
  }

  // ---- Graph --------------------------------------------------------------------
  void comm_info(gr_comm_info &o) override {
    std::memset(&o, 0, sizeof(o));
    o.device = device;
    if (!comm) return;
    o.rank = comm->rank; o.size = comm->size; o.transport = comm->transport(); o.rccl_ranks = comm->rccl_ranks();
    o.mailboxes_opened = comm->mailboxes_opened(); o.fused_agreed = fused_agreed ? 1 : 0;
    comm->message_counts(o.oneshot_messages, o.fallback_messages);
  }
  // gr_bal_model_orders: input index of the observation at camera-major position j; caller's index of engine landmark q
  void orders(int32_t *obs_order, int32_t *landmark_order) override {
    if (obs_order) {
      std::vector<int> orig_of_pm(No);
      for (int64_t o = 0; o < No; ++o) orig_of_pm[h_pm_of_orig[o]] = (int)o;
      for (int64_t j = 0; j < No; ++j) obs_order[j] = orig_of_pm[h_pos_cm[j]];
    }
    if (landmark_order) for (int64_t q = 0; q < Np; ++q) landmark_order[q] = h_pt_new2old[q];
  }
  void set_loss(int kind, double delta) override { no_model("gr_bal_set_loss"); loss_kind = kind; loss_delta = (T)delta; }
  void set_scale_system(bool on) override { scale_system = on; }
  // VertexDescriptor::set_fixed (vertex.hpp:262-264).  The reference gives a fixed vertex no Hessian column and its kernels
  // skip the vertex's Jacobian block; here the vertex keeps its (empty) column: zero block, zero gradient, scale 1, and its
  // rows of every operator product are dropped, so its step is exactly 0 and every other quantity is what the reduced
  // system gives.  The explicit Schur solvers get there through zero camera-point blocks (k_linearize), the implicit-Schur
  // PCG through a zero point "inverse" (k_point_prepare) and dropped camera rows (k_is_finalize, k_is_apply).
  DevBuf<unsigned char> d_cam_fixed, d_pt_fixed;
  bool has_fixed = false;
  void set_fixed(const unsigned char *cam_fixed, const unsigned char *pt_fixed) override {
    std::vector<unsigned char> hc(Nc, 0), hp(Np, 0);
    bool any = false;
    if (cam_fixed) for (int64_t c = 0; c < Nc; ++c) { hc[c] = cam_fixed[c] ? 1 : 0; any |= hc[c] != 0; }
    if (pt_fixed) for (int64_t l = 0; l < Np; ++l) { const unsigned char f = pt_fixed[l] ? 1 : 0; hp[h_pt_old2new.empty() ? l : h_pt_old2new[l]] = f; any |= f != 0; }
    has_fixed = any;
    if (any) { d_cam_fixed.upload(hc, stream); d_pt_fixed.upload(hp, stream); GR_HIP(hipStreamSynchronize(stream)); }
    hcp_valid = false;
  }
  const unsigned char *cam_fixed_p() const { return has_fixed ? d_cam_fixed.p : nullptr; }
  const unsigned char *pt_fixed_p() const { return has_fixed ? d_pt_fixed.p : nullptr; }
  void set_jacobian_precision(int dtype) override {
    no_model("gr_bal_set_jacobian_precision");
    rp_fit = -1;
    if (dtype == GR_F32 && sizeof(T) == 4) { jac32 = false; return; } // already fp32 throughout
    if (dtype != GR_F32 && dtype != GR_F64) throw std::invalid_argument("jacobian precision: GR_F32 or GR_F64");
    if (dtype == GR_F64 && sizeof(T) == 4) throw std::invalid_argument("an fp32 problem cannot evaluate fp64 Jacobians");
    jac32 = dtype == GR_F32;
    hcp_valid = false;
  }
  // user point order <-> internal point order at the API boundary (width scalars per point)
  void points_in(const void *user_src, T *dev_dst, int width) {
    std::vector<T> a((size_t)Np * width), b((size_t)Np * width);
    GR_HIP(hipMemcpy(a.data(), user_src, a.size() * sizeof(T), hipMemcpyDefault));
    for (int64_t q = 0; q < Np; ++q)
      for (int k = 0; k < width; ++k) b[(size_t)q * width + k] = a[(size_t)h_pt_new2old[q] * width + k];
    UploadRing::get().upload(dev_dst, b.data(), b.size() * sizeof(T), stream);
    GR_HIP(hipStreamSynchronize(stream));
  }
  void points_out(const T *dev_src, void *user_dst, int width) {
    std::vector<T> a((size_t)Np * width), b((size_t)Np * width);
    GR_HIP(hipMemcpyAsync(a.data(), dev_src, a.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
    for (int64_t q = 0; q < Np; ++q)
      for (int k = 0; k < width; ++k) b[(size_t)h_pt_new2old[q] * width + k] = a[(size_t)q * width + k];
    GR_HIP(hipMemcpy(user_dst, b.data(), b.size() * sizeof(T), hipMemcpyDefault));
  }
  void set_params(const void *c, const void *p) override {
    no_model("gr_bal_set_params");
    GR_HIP(hipMemcpyAsync(cams.p, c, cams.n * sizeof(T), hipMemcpyDefault, stream));
    points_in(p, pts.p, 3);
    xp_valid = false;
  }
  void get_params(void *c, void *p) override {
    no_model("gr_bal_get_params");
    GR_HIP(hipMemcpyAsync(c, cams.p, cams.n * sizeof(T), hipMemcpyDefault, stream));
    points_out(pts.p, p, 3);
  }
  double w() const { return (double)sizeof(T); }

  void campack(const T *dx = nullptr, T *bak = nullptr) {
    if (model) return; // no pack: the user-side kernels read the user's vertex objects
    k_campack<T><<<cdiv(Nc, 64), 64, 0, stream>>>((int)Nc, cams.p, pack.p, dx, scales.p, bak);
  }

  // Graph::linearize (graph.hpp:236-290) fused with Hessian::update_values
  // (hessian.hpp:290-307): ONE launch over the observations (pm blocks + cm chunks)
  // followed by the finalize kernel (chunk sums, scales, chi2).
  // spec_seq != 0: speculative trial linearisation of the LM loop; the finalize kernel also folds the
  // rho-denominator partials and publishes (chi2, denominator, spec_seq) to pinned host memory
  DevBuf<double> rho_partial;
  int rho_blocks = 0;
  // part: 1 = the k_linearize launch only, 2 = its finalisation (+ the collectives of landmark shards) only, 3 = both.  The LM
  // loop of landmark shards enqueues part 1 AHEAD of the PCG exit flag (gated on the device, no collective inside: a rank whose
  // gate was still closed repeats it later on its own) and part 2 once the flag has been seen.
  // user-traits problems with a stored Jacobian: the lineariser also writes the weighted blocks (2 x (pose + landmark) scalars per observation)
  double model_jac_bytes() const {
    return (model && model->store_jacobians) ? No * (2.0 * model->pose_dim + 2.0 * model->landmark_dim) * (model->storage_dtype == GR_F64 ? 8.0 : 4.0) : 0.0;
  }
  void linearize_impl(bool write_hcp, bool pack_valid = false, int spec_seq = 0, const int *gate = nullptr, int part = 3) {
    if (!pack_valid) campack();
    if (part & 1) {
      // algorithmic bytes: every array touched once (obs, 3 index streams, points, packs, g9 out, partials, Hcp)
      double bytes = No * (2 * w() + 12.0) + (24.0 * Nc + 3.0 * Np) * w() + 8.0 * No * w() + 54.0 * nseg * w() + (write_hcp ? 27.0 * No * w() : 0.0);
      bytes += model_jac_bytes();
      Scope sc(this, write_hcp ? "linearize_hcp" : "linearize", bytes, No * (250.0 + 48 + 117 + (write_hcp ? 81.0 : 0.0)), true);
      launch_linearize_cam(write_hcp, g9.p, gate);
    }
    if (!(part & 2)) return;
    {
      const int np_fin = (int)Np;
      Scope sc(this, "linearize_finalize", 8.0 * No * w() + 54.0 * nseg * w() + (90.0 * Nc + 15.0 * Np) * w(), 9.0 * No + 54.0 * nseg, true);
      const bool fz_lin = comm && !diag_running && shard_fused_lin();
      IpcFused fz{};
      if (fz_lin) fz = fused_fz();
      launch(k_linearize_finalize<T>, cdiv(90 * (size_t)Nc, TPB) + cdiv(FIN_PL * (size_t)np_fin, TPB), (int)Nc, np_fin, scale_system ? 1 : 0, comm ? 0 : 1, cam_seg_ptr.p, cam_partial.p, pt_ptr.p, g9.p, Hcc.p, bc.p, Hll.p, bl.p, scales.p, grid_lin, chi2_partial.p, dscalars.p,
                                                                                                    spec_seq ? rho_partial.p : nullptr, spec_seq ? rho_blocks : 0, (spec_seq && (!comm || fz_lin)) ? h_res : nullptr, h_seq, spec_seq,
                                                                                                    gate, cam_fixed_p(), pt_fixed_p(), fz, (unsigned long long)shard_lin_scal_off());
      if (fz_lin) { // the camera-space sums over the landmark shards travelled with the finalize launch: no all-reduce kernel, no publish kernel
        ++coll_count; ++fused_messages;
        launch(k_shard_cam_sums<T>, cdiv(90 * (size_t)Nc, TPB), (int)Nc, scale_system ? 1 : 0, fz, Hcc.p, bc.p, scales.p, cam_fixed_p());
        hcp_valid = write_hcp;
        return;
      }
    }
    if (comm && !diag_running) { // camera-space sums over the landmark shards; not inside a rank-local timing run (diag_time)
      group_start();
      allreduce_T(Hcc.p, 81 * (size_t)Nc);
      allreduce_T(bc.p, pose_dim);
      allreduce_d(dscalars.p, spec_seq ? 2 : 1);
      group_end();
      k_camera_scales<T><<<cdiv(pose_dim, TPB), TPB, 0, stream>>>((int)Nc, scale_system ? 1 : 0, Hcc.p, scales.p);
      ++launch_count;
      if (spec_seq) { publish_scalars(2, spec_seq); ++launch_count; }
    }
    hcp_valid = write_hcp;
  }
  bool lin_reset = false; // LM loop, fused form: this k_linearize launch also clears the PCG loop state (its last workgroup)
  void launch_linearize_cam(bool hcp, T *g9p, const int *gate) {
    if (model) {
      if (model->store_jacobians && !jst.p) {
        const size_t sw = model->storage_dtype == GR_F64 ? 8 : 4;
        jst_stride = ((int64_t)No + 63) / 64 * 64;
        jst.alloc((size_t)GR_MODEL_JSTREAMS * (size_t)jst_stride * sw);
        GR_HIP(hipMemsetAsync(jst.p, 0, jst.n, stream)); // the streams of zero-padded rows / columns are never written
      }
      gr_model_lin_args a{};
      a.No = (int)No; a.ntiles = o_ntiles(); a.grid = grid_lin;
      a.cam = o_cam(); a.pt = o_pt(); a.pos = o_pos(); a.blk_seg = blk_seg.p; a.seg_slot = seg_slot.p;
      a.g9 = g9p; a.Hcp = hcp ? Hcp.p : nullptr; a.cam_partial = cam_partial.p; a.chi2_partial = chi2_partial.p;
      a.jst = model->store_jacobians ? jst.p : nullptr; a.jst_stride = jst_stride;
      a.lm = nullptr; a.gate = gate; a.cam_fixed = hcp ? cam_fixed_p() : nullptr; a.pt_fixed = hcp ? pt_fixed_p() : nullptr; a.stream = stream;
      if (hcp && sf_cont) { a.Hcp_alt = Hcp1.p; a.hsel = lmdev.p; }
      hook_begin();
      hook_end(model->linearize(model->ctx, &a), "linearize");
      return;
    }
    const PcgState rst = lin_reset ? pcg_state() : PcgState{};
    const int rst_cap = lin_reset ? ctl_cap : 0;
#ifdef GR_DIAG
    { // diagnostic builds: GR_LIN_VAR=1|2|4|8 runs an ablated lineariser INSIDE the solve (wrong numbers, real cache state)
      static const int var = getenv("GR_LIN_VAR") ? atoi(getenv("GR_LIN_VAR")) : 0;
#define GR_LINV(V) launch(k_linearize<T, false, T, V>, grid_lin, (int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, g9p, nullptr, cam_partial.p, chi2_partial.p, nullptr, gate, nullptr, nullptr, rst, rst_cap, nullptr, nullptr)
      if (!hcp && var == 1) { GR_LINV(1); return; }
      if (!hcp && var == 2) { GR_LINV(2); return; }
      if (!hcp && var == 4) { GR_LINV(4); return; }
      if (!hcp && var == 8) { GR_LINV(8); return; }
#undef GR_LINV
    }
#endif
    if constexpr (sizeof(T) == 8) {
      if (jac32) {
        if (hcp) launch(k_linearize<T, true, float>, grid_lin, (int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, g9p, Hcp.p, cam_partial.p, chi2_partial.p, nullptr, gate, cam_fixed_p(), pt_fixed_p(), rst, rst_cap, sf_cont ? Hcp1.p : nullptr, sf_cont ? lmdev.p : nullptr);
        else launch(k_linearize<T, false, float>, grid_lin, (int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, g9p, nullptr, cam_partial.p, chi2_partial.p, nullptr, gate, nullptr, nullptr, rst, rst_cap, nullptr, nullptr);
        return;
      }
    }
    if (hcp) launch(k_linearize<T, true>, grid_lin, (int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, g9p, Hcp.p, cam_partial.p, chi2_partial.p, nullptr, gate, cam_fixed_p(), pt_fixed_p(), rst, rst_cap, sf_cont ? Hcp1.p : nullptr, sf_cont ? lmdev.p : nullptr);
    else launch(k_linearize<T, false>, grid_lin, (int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, g9p, nullptr, cam_partial.p, chi2_partial.p, nullptr, gate, nullptr, nullptr, rst, rst_cap, nullptr, nullptr);
  }
  bool want_hcp = false;
  void linearize() override { linearize_impl(want_hcp); }

  // Graph::compute_error + Graph::chi2 (graph.hpp:212-225)
  double chi2() override {
    campack();
    const int seq = chi2_async(nullptr, nullptr, 0.0);
    wait_chi2(seq);
    return h_res[0];
  }
  // dscalars[0] = chi2; with dx also dscalars[1] = sum dx (mu dx + b) (compute_rho).  The last
  // block mirrors both into pinned host memory and then publishes `seq`.
  int chi2_async(T *res_out, const T *dx, double mu) {
    const int seq = ++seq_counter;
    if (model) {
      Scope sc(this, "chi2", No * 12.0 + (dx ? 3.0 * n * w() : 0.0), No * 40.0);
      gr_model_chi2_args a{};
      a.No = (int)No; a.grid = grid_chi2; a.cam = o_cam(); a.pt = o_pt(); a.pos = o_pos(); a.chi2_partial = chi2_partial.p; a.res_out = res_out; a.stream = stream;
      hook_begin();
      hook_end(model->chi2(model->ctx, &a), "chi2");
      const int nb = std::max(1, std::min(cdiv(n, (size_t)TPB * 4), 256));
      k_chi2_finish<T><<<nb, TPB, 0, stream>>>((unsigned)n, (unsigned)pose_dim, cam_weight(), chi2_partial.p, grid_chi2, dx, bu.p, scales.p, mu, grid_partial.p, ticket.p, dscalars.p, comm ? nullptr : h_res, h_seq, seq);
      if (comm) { allreduce_d(dscalars.p, 2); publish_scalars(2, seq); }
      return seq;
    }
    Scope sc(this, "chi2", No * (2 * w() + 8) + (24.0 * Nc + 3.0 * Np) * w() + (dx ? 3.0 * n * w() : 0.0), No * 40.0);
    k_chi2<T><<<grid_chi2, TPB, 0, stream>>>((int)No, (unsigned)n, (unsigned)pose_dim, cam_weight(), o_cam(), o_pt(), o_pos(), o_obs(), pts.p, pack.p, loss_kind, loss_delta, dx, bu.p, scales.p, mu, chi2_partial.p, ticket.p, dscalars.p, comm ? nullptr : h_res, h_seq, seq, res_out);
    if (comm) { allreduce_d(dscalars.p, 2); publish_scalars(2, seq); }
    return seq;
  }
  // landmark shards: the sums over ranks live in device memory; one tiny launch mirrors them to the pinned words the host polls
  void publish_scalars(int count, int seq) { k_publish_scalars<<<1, 1, 0, stream>>>(dscalars.p, count, h_res, h_seq, seq); }
  void wait_chi2(int seq) {
    spin_until([&] { return __atomic_load_n(const_cast<const int *>(h_seq), __ATOMIC_ACQUIRE) == seq; });
    if (comm) check_comm("levenberg_marquardt"); // the accept decision of every rank hangs on these sums
  }
  double read_scalar(int idx) {
    double v = 0;
    GR_HIP(hipMemcpyAsync(&v, dscalars.p + idx, sizeof(double), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
    return v;
  }

  // user-traits problems: Graph::backup_parameters + apply_update through the user's Traits::update (gr_model_ops.step)
  // clear_state: the PCG loop state is reset inside the step launch (what k_pcg_state_init does in its gated form: the slots, done, pdp,
  // rz0 = {inf, 0 ...}, the iteration count; the exit word `left` stays — it IS the gate)
  void model_step(const T *dx, bool with_backup, double mu, double *rho_part, const LmDev *lm, const int *gate, bool clear_state = false) {
    gr_model_step_args a{};
    a.dx = dx; a.scales = scales.p; a.bu = bu.p; a.mu = mu; a.with_backup = with_backup ? 1 : 0; a.cam_weight = cam_weight();
    a.rho_partial = rho_part; a.lm = lm; a.gate = gate; a.cam_fixed = cam_fixed_p(); a.pt_fixed = pt_fixed_p(); a.stream = stream;
    a.restore_on_hsel = (lm && sf_cont) ? 1 : 0; // (LmDev::hsel bit 1 is only meaningful in the head that goes on after a rejected step)
    if (clear_state && ctl_cap > 0) {
      const PcgState st = pcg_state();
      a.clear_ptr[0] = ctl.p; a.clear_bytes[0] = (int64_t)(ctl.n * sizeof(double));
      a.clear_ptr[1] = ctl_i.p; a.clear_bytes[1] = (int64_t)(ctl_i.n * sizeof(int));
      a.clear_ptr[2] = st.iters; a.clear_bytes[2] = (int64_t)sizeof(int);
      a.inf_word = st.rz0;
    }
    hook_begin();
    hook_end(model->step(model->ctx, &a), "step");
  }
  void backup() override { // graph.hpp:302-309
    if (model) { hook_begin(); hook_end(model->backup(model->ctx, stream), "backup"); return; }
    GR_HIP(hipMemcpyAsync(cams_bak.p, cams.p, cams.n * sizeof(T), hipMemcpyDeviceToDevice, stream));
    GR_HIP(hipMemcpyAsync(pts_bak.p, pts.p, pts.n * sizeof(T), hipMemcpyDeviceToDevice, stream));
  }
  void revert() override { // graph.hpp:311-318
    if (model) { hook_begin(); hook_end(model->revert(model->ctx, stream), "revert"); return; }
    // one launch: vertices <- backups and the camera packs (the matrix-free operator recomputes J from the pack: kept in step)
    k_revert_pack<T><<<cdiv(Nc, 28) + cdiv(3 * (size_t)Np, TPB), TPB, 0, stream>>>((unsigned)pose_dim, (unsigned)(3 * Np), cdiv(Nc, 28), cams.p, pts.p, cams_bak.p, pts_bak.p, pack.p);
    xp_valid = false;
  }
  void apply_update_dev(const T *dx, bool with_backup = false) { // graph.hpp:292-300, ops/update.hpp:11-31
    if (model) { model_step(dx, with_backup, 0.0, nullptr, nullptr, nullptr); return; }
    campack(dx, with_backup ? cams_bak.p : nullptr); // cameras: x += dx .* s fused with backup + pack rebuild
    k_apply_update<T><<<cdiv(3 * Np, TPB), TPB, 0, stream>>>((unsigned)(3 * Np), pts.p, dx + pose_dim, scales.p + pose_dim, with_backup ? pts_bak.p : nullptr);
    xp_valid = false;
  }
  void apply_update(const void *dx) override {
    GR_HIP(hipMemcpyAsync(v_dx.p, dx, pose_dim * sizeof(T), hipMemcpyDefault, stream));
    points_in(static_cast<const T *>(dx) + pose_dim, v_dx.p + pose_dim, 3);
    apply_update_dev(v_dx.p);
  }

  // ---- Solver interface ---------------------------------------------------------
  // per-block partials of the Schur-PCG scalars (kernels.hpp PcgScalars): one slot per block of the largest producer grid
  int sc_np() const { return (std::max({cdiv(Nc, 64), cdiv(pose_dim, 252), cdiv(Nc, 4)}) + 63) / 64 * 64; }
  void ensure_scalars(int max_iter) {
    const int cap = max_iter + 2;
    if (cap > sc_cap) { sc_cap = cap; sc_d.alloc((2 * (size_t)sc_np() + 1) * cap); sc_i.alloc((size_t)cap + 1); }
    alloc_pinned(cap);
  }
  PcgScalars scalars() {
    PcgScalars sc;
    sc.np = sc_np();
    const size_t blk = (size_t)sc_cap * sc.np;
    sc.rz = sc_d.p; sc.den = sc_d.p + blk; sc.rz0 = sc_d.p + 2 * blk;
    sc.done = sc_i.p; sc.iters = sc_i.p + sc_cap;
    sc.hflag = flags(); sc.hiters = h_iters();
    return sc;
  }

  void solver_update_structure(int solver) override {
    if (solver == GR_SOLVER_PCG_SCHUR) { build_schur_structure(); want_hcp = true; }
    else if (solver == GR_SOLVER_DENSE_SCHUR) { ensure_chol(); want_hcp = true; }
    else if (solver == GR_SOLVER_PCG_SCHUR_IMPLICIT) {
      no_model("GR_SOLVER_PCG_SCHUR_IMPLICIT (its kernels recompute the built-in model's Jacobian; use GR_SOLVER_PCG_SCHUR)");
      want_hcp = false; ensure_implicit_schur();
      tiling_for_pcg = false; if (!tiling_tuned) tune_tiling();
      if (want_g3_gather() && !g3_obs_order) build_g3_gather(); // pass 1's output in observation order, gathered by k_is_points
    }
    else {
      want_hcp = false;
      lazy_cfg = tune.pcg_lazy < 0 ? -1 : (tune.pcg_lazy != 0 ? 1 : 0);
      cg_cfg = tune.pcg_single_reduction < 0 ? -1 : (tune.pcg_single_reduction != 0 ? 1 : 0);
      v_r.alloc(n); v_p.alloc(n); v_z.alloc(n); v_xb.alloc(n); v_ps.alloc(n); v_diag.alloc(n);
      MinvC.alloc(81 * (size_t)Nc); MinvP.alloc(9 * (size_t)Np);
      tiling_for_pcg = true; if (!tiling_tuned) tune_tiling();
      if (want_g3_gather() && !g3_obs_order) build_g3_gather(); // rebuilt whenever the observation order has changed
      if (!records_tuned) tune_point_records();
      if (comm && !diag_running) agree_on_fusion(); // collective: every rank of a sharded problem calls solver_update_structure
    }
  }
  // H.update_values / preconditioner->update_values: the blocks are produced by
  // linearize() already (fused); only the Hcp blocks may have to be (re)built.
  void solver_update_values(int solver) override {
    if ((solver == GR_SOLVER_PCG_SCHUR || solver == GR_SOLVER_DENSE_SCHUR) && !hcp_valid) linearize_impl(true);
  }
  void solver_set_damping(int solver, double mu, bool use_identity) override {
    damping = mu; damping_identity = use_identity;
    if (solver == GR_SOLVER_PCG || solver == GR_SOLVER_PCG_IDENTITY) {
      // BlockJacobiPreconditioner::set_damping_factor (block_jacobi.hpp:120-172); the identity
      // variant only needs the clamped diagonal (pcg.hpp:93-103)
      // one launch: camera inverses, point inverses and (when the loop state exists) its reset
      PcgState st{};
      if (ctl_cap > 0) { ensure_point_records(); st = pcg_state(); } // records first: the fused PCG start below writes their zs half (single-reduction form)
      const int nbc = cdiv(Nc, 64), nbp = cdiv(Np, 64);
      const bool fuse = lm_x && ctl_cap > 0 && state_clean_cap == ctl_cap;
      if (fuse) {
        k_block_jacobi<T><<<nbc + nbp, 64, 0, stream>>>((int)Nc, (int)Np, nbc, nbp, Hcc.p, Hll.p, scales.p, mu, use_identity ? 1 : 0, MinvC.p, MinvP.p, v_diag.p, st, ctl_cap, nullptr,
                                                         bu.p, lm_x, v_r.p, v_z.p, solver == GR_SOLVER_PCG_IDENTITY ? 1 : 0, cam_weight());
        update0_done = true; update0_identity = solver == GR_SOLVER_PCG_IDENTITY;
        state_clean_cap = -1; state_fresh_cap = -1;
      } else {
        k_block_jacobi<T><<<nbc + nbp + 1, 64, 0, stream>>>((int)Nc, (int)Np, nbc, nbp, Hcc.p, Hll.p, scales.p, mu, use_identity ? 1 : 0, MinvC.p, MinvP.p, v_diag.p, st, ctl_cap);
        state_fresh_cap = ctl_cap;
        update0_done = false;
      }
    }
  }

  // SchurComplement::update_values (schur.hpp:227-235)
  void schur_update_values() override { schur_update_values_impl(false); }
  // for_solve: the PCG scalars are reset by the first kernel and b_S is left to k_schur_pcg_prepare
  void schur_update_values_impl(bool for_solve) {
    build_schur_structure();
    if (!hcp_valid) linearize_impl(true);
    const int ui = damping_identity ? 1 : 0;
    const T *hcc_w = cam_weight() ? Hcc.p : nullptr; // the (global) camera blocks enter the all-reduced S once
    if (for_solve) k_point_prepare<T><<<cdiv(Np, TPB) + 1, TPB, 0, stream>>>((int)Np, (int)Nc, Hll.p, bl.p, scales.p, damping, ui, Hll_inv.p, Mp.p, vl.p, scalars(), sc_cap, pt_fixed_p());
    else k_point_prepare<T><<<cdiv(Np, TPB), TPB, 0, stream>>>((int)Np, (int)Nc, Hll.p, bl.p, scales.p, damping, ui, Hll_inv.p, Mp.p, vl.p, PcgScalars{}, 0, pt_fixed_p());
    if (tune.schur_fused != 0) launch_schur_reduce(hcc_w, damping, ui, nullptr); // S (multi-item blocks by their last arriver) + the b_S chunk partials
    else {
    if (nmulti) k_schur_multi<T, 0><<<cdiv(9 * (size_t)nmulti, TPB), TPB, 0, stream>>>(nmulti, multi_blk.p, S_rowi.p, S_coli.p, hcc_w, scales.p, damping, ui, S.p);
    {
      Scope sc(this, "schur_products", nprod * (54.0 * w() + 8) + 9.0 * Np * w() + 81.0 * nnzb * w(), nprod * 342.0);
      k_schur_products<T><<<cdiv(nitems, 4), TPB, 0, stream>>>(nitems, item_blk.p, item_beg.p, item_end.p, item_single.p, prod_a.p, prod_b.p, S_rowi.p, S_coli.p, pt_pm.p, Hcp.p, Mp.p, hcc_w, scales.p, damping, ui, S.p);
    }
    if (nmulti) k_schur_multi<T, 1><<<cdiv(9 * (size_t)nmulti, TPB), TPB, 0, stream>>>(nmulti, multi_blk.p, S_rowi.p, S_coli.p, hcc_w, scales.p, damping, ui, S.p);
    {
      Scope sc(this, "b_schur", No * (27.0 * w() + 8) + 3.0 * Np * w(), No * 54.0);
      k_bschur_partial<T><<<cdiv(nch, 4), TPB, 0, stream>>>(nch, chunk_beg.p, pt_cm.p, pos_cm.p, Hcp.p, vl.p, part9.p);
    }
    }
    if (!for_solve || comm) k_bschur_finalize<T><<<cdiv(pose_dim, TPB), TPB, 0, stream>>>((int)Nc, cam_chunk_ptr.p, part9.p, bc.p, scales.p, b_schur.p, cam_weight());
    if (comm) {
      // every rank reduced over its own points (rank 0 also carries the damped Hcc and bc, which are global already):
      // one grouped all-reduce makes S and b_S global, the solve that follows is replicated
      Scope sc(this, "allreduce_schur", 0, 0);
      group_start();
      allreduce_T(S.p, 81 * (size_t)nnzb);
      allreduce_T(b_schur.p, (size_t)pose_dim);
      group_end();
    }
  }
  void launch_schur_reduce(const T *hcc_w, double mu, int ui, const LmDev *lm) {
    Scope sc(this, "schur_products", nprod * (54.0 * w() + 8) + 9.0 * Np * w() + 81.0 * nnzb * w() + No * (27.0 * w() + 8) + 3.0 * Np * w(), nprod * 342.0 + No * 54.0, true);
    const int nwg_items = cdiv(nitems, 4);
    launch(k_schur_reduce<T>, nwg_items + cdiv(nch, 4), nitems, nwg_items, item_blk.p, item_beg.p, item_end.p, item_multi.p, multi_first.p, multi_n.p, multi_cnt.p, slab.p,
           prod_a.p, prod_b.p, S_rowi.p, S_coli.p, prod_pm.p, Hcp.p, Mp.p, hcc_w, scales.p, mu, ui, S.p, nch, chunk_beg.p, pt_cm.p, pos_cm.p, vl.p, part9.p, lm, sf_cont ? Hcp1.p : nullptr, sf_cont ? lmdev.p : nullptr);
  }
  // replicated reduced solves: rank 0's camera step is the one every rank applies (the per-rank copies agree
  // only up to the order of the atomic dot-product partials, which must not leak into the replicated cameras)
  void broadcast_camera_step(T *x) {
    if (!comm) return;
    if (comm->rank != 0) GR_HIP(hipMemsetAsync(x, 0, pose_dim * sizeof(T), stream));
    allreduce_T(x, (size_t)pose_dim);
  }
  void schur_matvec_dev(const T *x, T *y, int k) {
    Scope sc(this, "schur_matvec", (2.0 * nnzb - Nc) * 81.0 * w(), (2.0 * nnzb - Nc) * 162.0);
    k_schur_matvec<T><<<cdiv(Nc, 4), TPB, 0, stream>>>((int)Nc, row_ptr.p, row_blk.p, row_col.p, S.p, x, y, scalars(), k);
  }
  void schur_matvec(const void *x, void *y) override {
    build_schur_structure();
    ensure_scalars(1);
    GR_HIP(hipMemcpyAsync(v_p.p, x, pose_dim * sizeof(T), hipMemcpyDefault, stream));
    schur_matvec_dev(v_p.p, v_Ap.p, -1);
    GR_HIP(hipMemcpyAsync(y, v_Ap.p, pose_dim * sizeof(T), hipMemcpyDefault, stream));
    GR_HIP(hipStreamSynchronize(stream));
  }
  void landmark_update_dev(const T *xp, T *xl) {
    {
      Scope sc(this, "backsub", No * (27.0 * w() + 8) + Np * 15.0 * w(), No * 54.0);
      k_backsub<T><<<nb_pm, TPB, 0, stream>>>((int)No, (int)Nc, pt_ptr.p, cam_pm.p, pt_pm.p, Hcp.p, Hll_inv.p, bl.p, scales.p, xp, xl, xl_acc.p, boundary_flag.p);
    }
    k_backsub_fixup<T><<<cdiv(Np, TPB), TPB, 0, stream>>>((int)Np, (int)Nc, Hll_inv.p, bl.p, scales.p, xl, xl_acc.p, boundary_flag.p);
  }
  void landmark_update(const void *xp, void *xl) override {
    GR_HIP(hipMemcpyAsync(v_p.p, xp, pose_dim * sizeof(T), hipMemcpyDefault, stream));
    landmark_update_dev(v_p.p, v_dx.p + pose_dim);
    points_out(v_dx.p + pose_dim, xl, 3);
  }
  void schur_structure(int64_t *nb, int64_t *colptr, int64_t *rowidx) override {
    build_schur_structure();
    if (nb) *nb = nnzb;
    if (colptr) for (int64_t c = 0; c <= Nc; ++c) colptr[c] = h_S_colptr[c];
    if (rowidx) for (int64_t q = 0; q < nnzb; ++q) rowidx[q] = h_S_rowi[q];
  }

  // PCGSchurSolver::solve (solver/pcg_schur.hpp:79-168); device-resident scalars,
  // no host round trip inside the loop.
  // host side of the device-resident PCG loops: enqueue with one iteration of look-ahead and stop
  // once the direction kernel has flagged (pinned memory) that the loop has left
  // Host side of the device-resident PCG loops.  Iteration k + 1 is enqueued BEFORE the flag of iteration k
  // is read only while k + 1 is still below the iteration count of the previous solve (the loop is expected
  // to continue: no bubble); past that the host first waits for the flag, so a loop that leaves where the last
  // one did costs neither a bubble nor no-op launches.  Returns the number of enqueued iterations that ran as
  // no-ops (the loop had already left).
  int predicted_iters = 0, last_active = 0;
  // LM host loop, speculative trial step: at the iteration where the loop is PREDICTED to leave, the trial-step
  // kernels (backup/update/rho, linearise, finalize) are enqueued before the exit flag has been seen, gated on the
  // device by PcgState::left.  Right prediction: the round trip between the last direction kernel and the
  // trial step disappears; wrong: three kernels return at once and the trial step is enqueued again later.
  std::function<void(const int *gate)> trial_hook;
  bool trial_done = false;
  int trial_part = 3; // linearize_impl's `part` for the trial enqueued by the LM loop
  DevBuf<int> loop_left;
  template <typename Enqueue> int run_pcg_iterations(int max_iter, Enqueue &&enqueue, int pre_enqueued = 0) {
    int enqueued = 0, hook_at = -1;
    bool left = false;
    trial_done = false;
    if (max_iter > 0) { if (!pre_enqueued) enqueue(0); ++enqueued; }
    for (int k = 0; k < max_iter; ++k) {
      if (k + 1 < max_iter && k + 1 < predicted_iters && enqueued == k + 1) { enqueue(k + 1); ++enqueued; }
      if (trial_hook && hook_at < 0 && predicted_iters > 0 && k + 1 >= predicted_iters && enqueued == k + 1) {
        trial_hook(k + 1 < max_iter ? loop_left.p : nullptr); // the iteration cap ends the loop whatever the flag says
        hook_at = k;
      }
      spin_until([&] { return __atomic_load_n(const_cast<const int *>(&flags()[k]), __ATOMIC_ACQUIRE) != 0; });
      if (flags()[k] == 2) { left = true; break; }
      if (k + 1 < max_iter && enqueued == k + 1) { enqueue(k + 1); ++enqueued; }
    }
    if (hook_at >= 0) {
      const int last = left ? std::min((int)*h_iters(), enqueued) - 1 : enqueued - 1; // last iteration that ran
      trial_done = (left && last <= hook_at) || hook_at + 1 == max_iter;
      if (!trial_done) note_noop({"linearize", "linearize_finalize"}, 1);
    }
    const int active = left ? std::min((int)*h_iters(), enqueued) : enqueued;
    // next solve's inner iteration count: the last one, extrapolated when the count is falling (10 -> 8 -> 2 -> 1 on the
    // bench line: "same as last time" enqueued one iteration too many in each of those solves, three no-op launches of 4.7 us)
    predicted_iters = std::max(1, active + std::min(0, active - last_active));
    last_active = active;
    return enqueued - active;
  }
  void note_noop(std::initializer_list<const char *> names, int count) {
    if (!profiling || count <= 0) return;
    for (const char *nm : names) {
      auto it = prof.find(nm);
      if (it != prof.end()) it->second.noop += count;
    }
  }
  int solve_pcg_schur(int max_iter, double tol, double rej, T *x) {
    build_schur_structure();
    ensure_scalars(max_iter);
    PcgScalars sc = scalars();
    for (int k = 0; k < max_iter + 1; ++k) flags()[k] = 0;
    *h_iters() = 0;
    schur_update_values_impl(true);
    if (comm) k_schur_pcg_prepare<T, 1><<<cdiv(Nc, 64), 64, 0, stream>>>((int)Nc, S.p, S_diag.p, cam_chunk_ptr.p, part9.p, bc.p, scales.p, b_schur.p, MinvS.p, v_r.p, v_z.p, v_p.p, x, nullptr, sc);
    else k_schur_pcg_prepare<T, 0><<<cdiv(Nc, 64), 64, 0, stream>>>((int)Nc, S.p, S_diag.p, cam_chunk_ptr.p, part9.p, bc.p, scales.p, b_schur.p, MinvS.p, v_r.p, v_z.p, v_p.p, x, nullptr, sc);
    const int noop = run_pcg_iterations(max_iter, [&](int k) {
      schur_matvec_dev(v_p.p, v_Ap.p, k);
      k_pcgs_update<T><<<cdiv(pose_dim, 252), TPB, 0, stream>>>((int)Nc, x, v_xb.p, v_r.p, v_z.p, v_p.p, v_Ap.p, MinvS.p, sc, k);
      k_pcgs_direction<T><<<cdiv(pose_dim, TPB), TPB, 0, stream>>>((int)Nc, x, v_xb.p, v_p.p, v_z.p, nullptr, nullptr, sc, k, tol, rej);
    });
    note_noop({"schur_matvec"}, noop);
    broadcast_camera_step(x);
    landmark_update_dev(x, x + pose_dim);
    return 0;
  }

  // Direct solve of the reduced camera system (chol.hpp): the role of EigenSchurLDLTSolver::solve
  // (solver/eigen_schur.hpp:71-108) / cudssSchurSolver::solve (cudss_schur.hpp:190-234), then
  // compute_landmark_update (schur.hpp:279-302).
  DenseChol<T> chol;
  bool chol_ready = false;
  struct CholSink : CholProfSink {
    Engine *e; std::vector<std::unique_ptr<Scope>> open;
    explicit CholSink(Engine *e_) : e(e_) {}
    void begin(const char *name, double bytes, double flops) override { open.emplace_back(new Scope(e, name, bytes, flops)); }
    void end() override { open.pop_back(); }
  };
  std::unique_ptr<CholSink> chol_sink;
  // sparse direct path (sparse_chol.hpp): nested dissection + level-scheduled tile Cholesky; used when the camera graph
  // dissects (the tile elimination tree is clearly shorter than the chain of tile columns), DenseChol otherwise.
  // GR_SPARSE_CHOL=0 / 1 forces the choice.
  SparseChol<T> spchol;
  bool spchol_overlap() const { return tune.spchol_overlap != 0; } // forward substitution beside the factorisation
  bool use_spchol = false;
  void ensure_chol() {
    build_schur_structure();
    if (chol_ready) return;
    {
      const int force = tune.sparse_cholesky;
      spchol.fuse_potrf = tune.spchol_fuse != 0; spchol.fuse_quads = tune.spchol_fuse >= 2; spchol.bwd_chain = tune.spchol_bwd_chain != 0; spchol.slice = std::max(1, tune.spchol_slice); spchol.overlap_form = tune.spchol_overlap == 2 ? 2 : 1;
      chol.fuse_potrf = tune.chol_fuse != 0; chol.pin_variant = tune.chol_pin;
      // what either form may take: 3/4 of the free HBM (the factor is the largest single allocation of the direct solvers)
      size_t mem_free = 0, mem_total = 0;
      GR_HIP(hipMemGetInfo(&mem_free, &mem_total));
      const size_t budget = mem_free / 4 * 3;
      const bool dense_fits = DenseChol<T>::bytes_needed((int64_t)pose_dim) <= budget;
      if (force != 0 && spchol.set_structure((int)Nc, h_S_rowi, h_S_coli, stream)) {
        // tile-sparse storage: chosen when the elimination tree is clearly shorter than the chain of tile columns, and whenever the
        // padded dense triangle would not fit but the factor's own tiles do (memory follows nnz(L))
        use_spchol = force == 1 || 2 * spchol.nlevels < spchol.nt || !dense_fits;
        if (spchol.bytes() > budget) use_spchol = false;
        if (tune.verbose)
          std::fprintf(stderr, "[graphite-mi355x] sparse Cholesky: %d supernodes, %d tile columns (padded n = %d), elimination-tree height %d, %lld factor tiles "
                               "= %.3f GB (dense triangle: %.3f GB) -> %s\n",
                       spchol.nsuper, spchol.nt, spchol.npad, spchol.nlevels, (long long)spchol.factor_tiles, spchol.bytes() / 1e9, spchol.dense_bytes() / 1e9,
                       use_spchol ? "nested dissection, level-scheduled, tile-sparse storage" : "dense tile Cholesky");
      }
      if (use_spchol) {
        spchol.allocate();
        chol_sink.reset(new CholSink(this));
        spchol.sink = chol_sink.get();
        chol_ready = true;
        return;
      }
      if (!dense_fits)
        throw std::invalid_argument("direct solve of the reduced camera system: neither the tile-sparse factor nor the padded dense triangle fits in the free HBM");
    }
    const int nt = (int)((pose_dim + CH_NB - 1) / CH_NB);
    std::vector<char> tz((size_t)nt * nt, 0);
    for (int64_t q = 0; q < nnzb; ++q) {
      const int i = h_S_rowi[q], j = h_S_coli[q]; // upper block (i <= j) -> lower rows 9j.., cols 9i..
      const int r0 = 9 * j / CH_NB, r1 = (9 * j + 8) / CH_NB, c0 = 9 * i / CH_NB, c1 = (9 * i + 8) / CH_NB;
      for (int r = r0; r <= r1; ++r)
        for (int c = c0; c <= c1; ++c) if (r >= c) tz[(size_t)r * nt + c] = 1;
    }
    chol.set_structure((int)pose_dim, std::move(tz), stream);
    chol_sink.reset(new CholSink(this));
    chol.sink = chol_sink.get();
    chol_ready = true;
  }
  void direct_solver_info(gr_direct_solver_info &o) override {
    if (!chol_ready) throw std::logic_error("gr_bal_direct_solver_info: call gr_bal_solver_update_structure(GR_SOLVER_DENSE_SCHUR) first");
    std::memset(&o, 0, sizeof(o));
    if (use_spchol) {
      o.sparse = 1; o.tile_columns = spchol.nt; o.levels = spchol.nlevels; o.supernodes = spchol.nsuper; o.padded_n = spchol.npad;
      o.factor_tiles = spchol.nz_tiles; o.factor_bytes = (int64_t)spchol.bytes(); o.dense_bytes = (int64_t)spchol.dense_bytes();
    } else {
      o.sparse = 0; o.tile_columns = chol.nt; o.levels = chol.nt; o.supernodes = 1; o.padded_n = chol.npad;
      o.factor_tiles = chol.nz_tiles; o.dense_bytes = (int64_t)chol.npad * chol.npad * (int64_t)sizeof(T);
      o.factor_bytes = o.dense_bytes + (int64_t)chol.nt * CH_NB * CH_NB * (int64_t)sizeof(T);
    }
  }
  bool solve_dense_schur(T *x) {
    ensure_chol();
    schur_update_values();
    if (use_spchol) {
      spchol.load(nnzb, S_rowi.p, S_coli.p, S.p);
      if (spchol_overlap()) spchol.factor_solve(b_schur.p, x);
      else { spchol.factor(); spchol.solve(b_schur.p, x); }
      broadcast_camera_step(x);
      landmark_update_dev(x, x + pose_dim);
      *h_iters() = 0;
      return spchol.ok();
    }
    chol.clear();
    k_chol_scatter<T><<<cdiv(81 * (size_t)nnzb, TPB), TPB, 0, stream>>>(nnzb, S_rowi.p, S_coli.p, S.p, chol.A.p, chol.ld());
    chol.factor();
    chol.solve(b_schur.p, x);
    broadcast_camera_step(x);
    landmark_update_dev(x, x + pose_dim);
    *h_iters() = 0;
    return chol.ok();
  }

  // Implicit Schur PCG (kernels_is.hpp): same iterates, S never formed, Jacobians recomputed.
  DevBuf<T> Sdiag, zl, v_q, is_raw;
  void ensure_implicit_schur() {
    Hll_inv.alloc(9 * (size_t)Np); Mp.alloc(9 * (size_t)Np); vl.alloc(3 * (size_t)Np); zl.alloc(3 * (size_t)Np);
    Sdiag.alloc(81 * (size_t)Nc); MinvS.alloc(81 * (size_t)Nc); b_schur.alloc(pose_dim); v_q.alloc(pose_dim);
    v_r.alloc(n); v_p.alloc(n); v_z.alloc(n); v_Ap.alloc(n); v_xb.alloc(n);
  }
  // k_is_prepare takes 242 VGPRs in fp64 (two workgroups per CU): on the common grid of four per CU half of its persistent
  // workgroups waited for a slot
  int grid_is_prepare() {
    static thread_local int per_cu = 0;
    if (per_cu == 0) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_is_prepare<T>), TPB, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 2; }
      per_cu = std::min(nb, 4);
    }
    return std::max(8, std::min(grid_obs, num_cu * per_cu) & ~7);
  }
  void solve_pcg_schur_implicit(int max_iter, double tol, double rej, T *x) {
    ensure_implicit_schur();
    const int ui = damping_identity ? 1 : 0;
    ensure_scalars(max_iter);
    PcgScalars sc = scalars();
    k_point_prepare<T><<<cdiv(Np, TPB) + 1, TPB, 0, stream>>>((int)Np, (int)Nc, Hll.p, bl.p, scales.p, damping, ui, Hll_inv.p, Mp.p, vl.p, sc, sc_cap, pt_fixed_p());
    {
      Scope s0(this, "is_prepare", No * (2 * w() + 12.0) + (24.0 * Nc + 15.0 * Np) * w() + 54.0 * nseg * w(), No * 700.0);
      if (jac32) { k_is_prepare<T, float><<<grid_is_prepare(), TPB, 0, stream>>>((int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, Mp.p, vl.p, cam_partial.p); } else { k_is_prepare<T><<<grid_is_prepare(), TPB, 0, stream>>>((int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, Mp.p, vl.p, cam_partial.p); }
    }
    if (comm) { // diagonal blocks of S and b_S: this shard's sums, all-reduced, then combined with the global Hcc, bc
      is_raw.alloc(90 * (size_t)Nc);
      k_is_finalize<T><<<cdiv(90 * (size_t)Nc, TPB), TPB, 0, stream>>>((int)Nc, cam_seg_ptr.p, cam_partial.p, Hcc.p, bc.p, scales.p, damping, ui, Sdiag.p, b_schur.p, nullptr, is_raw.p, cam_fixed_p());
      allreduce_T(is_raw.p, 90 * (size_t)Nc);
      k_is_finalize<T><<<cdiv(90 * (size_t)Nc, TPB), TPB, 0, stream>>>((int)Nc, cam_seg_ptr.p, cam_partial.p, Hcc.p, bc.p, scales.p, damping, ui, Sdiag.p, b_schur.p, is_raw.p, nullptr, cam_fixed_p());
    } else
      k_is_finalize<T><<<cdiv(90 * (size_t)Nc, TPB), TPB, 0, stream>>>((int)Nc, cam_seg_ptr.p, cam_partial.p, Hcc.p, bc.p, scales.p, damping, ui, Sdiag.p, b_schur.p, nullptr, nullptr, cam_fixed_p());
    for (int k = 0; k < max_iter + 1; ++k) flags()[k] = 0;
    *h_iters() = 0;
    k_schur_pcg_prepare<T, 2><<<cdiv(Nc, 64), 64, 0, stream>>>((int)Nc, Sdiag.p, nullptr, nullptr, nullptr, nullptr, scales.p, b_schur.p, MinvS.p, v_r.p, v_z.p, v_p.p, x, v_q.p, sc);
    const double pass_bytes = No * (2 * w() + 12.0) + (24.0 * Nc + 3.0 * Np + 9.0 * Nc) * w() + 3.0 * No * w();
    const int noop = run_pcg_iterations(max_iter, [&](int k) {
      {
        Scope s1(this, "is_pass1", pass_bytes, No * 290.0);
        if (jac32) { k_is_pass1<T, float><<<grid_obs, TPB, 0, stream>>>((int)No, o_ntiles(), o_cam(), o_pt(), g3_pos(), o_obs(), pts.p, pack.p, loss_kind, loss_delta, v_q.p, g3.p, sc, k); } else { k_is_pass1<T><<<grid_obs, TPB, 0, stream>>>((int)No, o_ntiles(), o_cam(), o_pt(), g3_pos(), o_obs(), pts.p, pack.p, loss_kind, loss_delta, v_q.p, g3.p, sc, k); }
      }
      k_is_points<T, 0><<<is_points_blocks(), TPB, 0, stream>>>((int)Np, (int)Nc, pt_ptr.p, g3.p, Mp.p, Hll_inv.p, bl.p, scales.p, zl.p, sc, k, g3_gather());
      {
        Scope s2(this, "is_pass2", No * (2 * w() + 12.0) + (24.0 * Nc + 6.0 * Np) * w() + 9.0 * nseg * w(), No * 290.0);
        if (jac32) { k_is_pass2<T, float><<<grid_obs, TPB, 0, stream>>>((int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, zl.p, op_partial.p, sc, k); } else { k_is_pass2<T><<<grid_obs, TPB, 0, stream>>>((int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, zl.p, op_partial.p, sc, k); }
      }
      if (comm) { // one all-reduce of the 9 Nc vector per iteration; r, z, p are replicated, so no dot crosses ranks
        k_cam_rows<T><<<cdiv(pose_dim, TPB), TPB, 0, stream>>>((int)Nc, cam_seg_ptr.p, op_partial.p, raw_c.p, sc.done, k);
        allreduce_T(raw_c.p, pose_dim);
      }
      k_is_apply<T><<<cdiv(pose_dim, 252), TPB, 0, stream>>>((int)Nc, cam_seg_ptr.p, op_partial.p, Hcc.p, scales.p, v_p.p, v_q.p, damping, ui, v_Ap.p, sc, k, comm ? raw_c.p : nullptr, cam_fixed_p());
      k_pcgs_update<T><<<cdiv(pose_dim, 252), TPB, 0, stream>>>((int)Nc, x, v_xb.p, v_r.p, v_z.p, v_p.p, v_Ap.p, MinvS.p, sc, k);
      k_pcgs_direction<T><<<cdiv(pose_dim, TPB), TPB, 0, stream>>>((int)Nc, x, v_xb.p, v_p.p, v_z.p, v_q.p, scales.p, sc, k, tol, rej);
    });
    note_noop({"is_pass1", "is_pass2"}, noop);
    // back-substitution x_l = Hll^-1 (b_l - Hpl^T x_p): pass 1 with q = s_c .* x_c, then the per-point solve
    k_mul<T><<<cdiv(pose_dim, TPB), TPB, 0, stream>>>((unsigned)pose_dim, v_q.p, scales.p, x);
    if (jac32) { k_is_pass1<T, float><<<grid_obs, TPB, 0, stream>>>((int)No, o_ntiles(), o_cam(), o_pt(), g3_pos(), o_obs(), pts.p, pack.p, loss_kind, loss_delta, v_q.p, g3.p, sc, -1); } else { k_is_pass1<T><<<grid_obs, TPB, 0, stream>>>((int)No, o_ntiles(), o_cam(), o_pt(), g3_pos(), o_obs(), pts.p, pack.p, loss_kind, loss_delta, v_q.p, g3.p, sc, -1); }
    k_is_points<T, 1><<<is_points_blocks(), TPB, 0, stream>>>((int)Np, (int)Nc, pt_ptr.p, g3.p, Mp.p, Hll_inv.p, bl.p, scales.p, x + pose_dim, sc, 0, g3_gather());
  }

  // ---- landmark shards: the inner iteration's message fused into the operator / update launches (kernels_mf.hpp ShardPush,
  // comm.hpp IpcFused).  Needs the IPC mailbox transport, the single-reduction PCG form, the plain camera-major order (a
  // workgroup's observations are one contiguous camera range) and a message that fits a mailbox slot.
  DevBuf<int> sp_cam_wg, sp_empty;
  DevBuf<unsigned> sp_cam_cnt;
  int sp_grid = -1, sp_nempty = 0;
  int64_t fused_messages = 0; // pushed by operator launches (host count of the enqueued ones)
  size_t shard_dots_off() const { return (9 * (size_t)Nc * sizeof(T) + 15) / 16 * 16; }
  IpcComm *ipc_comm() const { return dynamic_cast<IpcComm *>(comm.get()); }
  // The observation order is a per-rank, TIMED choice (tune_tiling): whether the inner iteration's message can be fused is
  // therefore AGREED between the ranks (solver_update_structure: one host all-reduce of this rank's answer) — a rank that pushed
  // from its operator launch while a peer waited in a mailbox kernel would hang both.
  bool shard_fused_local() const {
    IpcComm *ic = ipc_comm();
    if (!ic || tune.shard_fused == 0 || (ic->size < 2 && tune.shard_fused != 1) || tiled) return false;
    ic->virtual_ranks = tune.shard_virtual_ranks;
    return shard_dots_off() + NSLOT * sizeof(double) <= ic->fused_slot_bytes();
  }
  bool fused_agreed = false;
  // which ranks hold observations of which camera (IpcFused::contrib): bit r of entry c.  Agreed with the fusion itself (one host
  // all-reduce of 2^rank per held camera: exact in a double up to 32 ranks); gr_bal_comm_set_contributors overrides it (projection
  // runs, where one rank plays all: tools/shard_projection.py hands in the masks of the real partition)
  DevBuf<unsigned> d_contrib;
  bool contrib_forced = false;
  void agree_on_fusion() {
    fused_agreed = false;
    if (!ipc_comm()) return;
    double mine = shard_fused_local() ? 1.0 : 0.0;
    if (comm->size > 1) allreduce_host(&mine, 1);
    fused_agreed = mine == (double)comm->size;
    if (contrib_forced) return;
    d_contrib.release();
    if (comm->size > 32) return; // no masks: every rank pushes every row
    std::vector<double> m(Nc);
    for (int64_t c = 0; c < Nc; ++c) m[c] = h_cam_ptr[c + 1] > h_cam_ptr[c] ? (double)(1u << comm->rank) : 0.0;
    if (comm->size > 1) allreduce_host(m.data(), m.size());
    std::vector<unsigned> u(Nc);
    for (int64_t c = 0; c < Nc; ++c) u[c] = (unsigned)m[c];
    d_contrib.upload(u, stream);
  }
  void set_contributors(const unsigned *mask, int64_t count) override {
    if (count != Nc) throw std::invalid_argument("contributor masks: one entry per camera");
    std::vector<unsigned> u(mask, mask + count);
    d_contrib.upload(u, stream);
    contrib_forced = true;
  }
  IpcFused fused_fz() {
    ipc_comm()->virtual_ranks = tune.shard_virtual_ranks;
    IpcFused f = ipc_comm()->fused();
    f.contrib = d_contrib.n == (size_t)Nc ? d_contrib.p : nullptr;
    return f;
  }
  // never inside a rank-local timing run (diag_time): a fused launch pushes a mailbox message and waits for the peers', and the ranks
  // time different numbers of launches (the g3 layouts are only timed by ranks above a size threshold)
  bool shard_fused() const { return fused_agreed && !diag_running && tune.shard_fused != 0 && !tiled && pcg_mode() == 2; }
  // ... and the linearisation's camera-space sums [Hcc 81 Nc | bc 9 Nc | chi2, rho denominator] pushed by k_linearize_finalize
  size_t shard_lin_scal_off() const { return (90 * (size_t)Nc * sizeof(T) + 15) / 16 * 16; }
  bool shard_fused_lin() const {
    IpcComm *ic = ipc_comm();
    if (!ic || tune.shard_fused == 0 || (ic->size < 2 && tune.shard_fused != 1)) return false;
    ic->virtual_ranks = tune.shard_virtual_ranks;
    return shard_lin_scal_off() + 2 * sizeof(double) <= ic->fused_slot_bytes();
  }
  ShardPush shard_push() {
    if (sp_grid != grid_op) { // per camera: how many workgroups of the operator grid hold observations of it (xcd_tile_range, plain form)
      std::vector<int> wg(Nc, 0), empty;
      const std::vector<int> h_cam = cam_cm.download(stream);
      const int nb = grid_op >> 3, nblk = (int)((No + 63) >> 6); // as xcd_obs_range (kernels_mf.hpp), plain form: ranges of 64-observation blocks
      for (int b = 0; b < grid_op; ++b) {
        const int x = b & 7, bi = b >> 3;
        const int x0 = (int)((long long)x * nblk / 8), x1 = (int)((long long)(x + 1) * nblk / 8);
        const int b0 = x0 + (int)((long long)bi * (x1 - x0) / nb), b1 = x0 + (int)((long long)(bi + 1) * (x1 - x0) / nb);
        const long long j0 = (long long)b0 << 6, j1 = std::min<long long>((long long)b1 << 6, (long long)No);
        if (j0 >= j1) continue;
        for (int c = h_cam[j0]; c <= h_cam[j1 - 1]; ++c) if (h_cam_ptr[c + 1] > h_cam_ptr[c]) wg[c]++;
      }
      for (int64_t c = 0; c < Nc; ++c) if (h_cam_ptr[c + 1] == h_cam_ptr[c]) empty.push_back((int)c);
      sp_nempty = (int)empty.size();
      if (empty.empty()) empty.push_back(0);
      sp_cam_wg.upload(wg, stream); sp_empty.upload(empty, stream);
      sp_cam_cnt.alloc(Nc); sp_cam_cnt.zero(stream);
      GR_HIP(hipStreamSynchronize(stream));
      sp_grid = grid_op;
    }
    ShardPush sp;
    sp.fz = fused_fz();
    sp.cam_seg_ptr = cam_seg_ptr.p; sp.cam_wg = sp_cam_wg.p; sp.cam_cnt = sp_cam_cnt.p; sp.empty = sp_empty.p; sp.n_empty = sp_nempty;
    sp.dots_off = shard_dots_off();
    return sp;
  }
  template <typename JT> void launch_operator_j(PcgState st, int k, const T *rec, const LmDev *lm, double mu) {
#ifdef GR_DIAG
    { // diagnostic builds: GR_OP_VAR=1|2|3 runs an ablated operator INSIDE the solve (wrong numbers, real cache state)
      static const int var = getenv("GR_OP_VAR") ? atoi(getenv("GR_OP_VAR")) : 0;
#define GR_OPV(V) launch(k_pcg_operator<T, V, JT>, grid_op, (int)No, (int)Nc, o_ntiles(), o_cam(), o_pt(), g3_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, v_ps.p, g3.p, op_partial.p, mu, st, k, rec, lm, ShardPush{})
      if (var == 1) { GR_OPV(1); return; }
      if (var == 2) { GR_OPV(2); return; }
      if (var == 3) { GR_OPV(3); return; }
#undef GR_OPV
    }
#endif
    // operator forms: (LAZY, FUSE, REC) by the PCG form / shard message / record layout
#define GR_OP_ARGS(REC_P, SP) grid_op, (int)No, (int)Nc, o_ntiles(), o_cam(), o_pt(), g3_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, v_ps.p, g3.p, op_partial.p, mu, st, k, REC_P, lm, SP
#define GR_OP_FORM(LZ, FU, RC, REC_P, SP) launch(k_pcg_operator<T, 0, JT, LZ, FU, RC>, GR_OP_ARGS(REC_P, SP))
    if (st.lazy == 2 && shard_fused()) { if (rec) GR_OP_FORM(2, true, true, rec, shard_push()); else GR_OP_FORM(2, true, false, nullptr, shard_push()); }
    else if (st.lazy == 2) { if (rec) GR_OP_FORM(2, false, true, rec, ShardPush{}); else GR_OP_FORM(2, false, false, nullptr, ShardPush{}); }
    else if (st.lazy) GR_OP_FORM(1, false, false, nullptr, ShardPush{});
    else if (rec) GR_OP_FORM(0, false, true, rec, ShardPush{});
    else GR_OP_FORM(0, false, false, nullptr, ShardPush{});
#undef GR_OP_FORM
#undef GR_OP_ARGS
  }
  // user-traits problems: the library's operator on the stored weighted Jacobian, or the user-side one that recomputes the blocks
  template <typename SJ> void launch_operator_stored(PcgState st, int k, const LmDev *lm) {
    const SJ *j = reinterpret_cast<const SJ *>(jst.p);
#define GR_OPS(DC) launch(k_pcg_operator_stored<T, SJ, DC>, grid_op, (int)No, (int)Nc, o_ntiles(), o_cam(), o_pt(), g3_pos(), blk_seg.p, seg_slot.p, j, (long long)jst_stride, v_ps.p, g3.p, op_partial.p, st, k, lm)
    const int dc = model->pose_dim;
    if (dc <= 6) GR_OPS(6); else if (dc == 7) GR_OPS(7); else GR_OPS(9);
#undef GR_OPS
  }
  void launch_operator_model(PcgState st, int k, const LmDev *lm) {
    if (st.lazy) throw std::invalid_argument("user-traits problems run the direction-kernel form of the PCG");
    if (model->store_jacobians) {
      if (!jst.p) throw std::invalid_argument("stored-Jacobian operator before the first linearisation");
      if constexpr (sizeof(T) == 8) { if (model->storage_dtype == GR_F32) { launch_operator_stored<float>(st, k, lm); return; } }
      launch_operator_stored<T>(st, k, lm);
      return;
    }
    if (!model->op) throw std::invalid_argument("gr_model_ops.op is NULL and store_jacobians is 0");
    gr_model_op_args a{};
    a.No = (int)No; a.Nc = (int)Nc; a.ntiles = o_ntiles(); a.grid = grid_op; a.cam = o_cam(); a.pt = o_pt(); a.pos = g3_pos();
    a.blk_seg = blk_seg.p; a.seg_slot = seg_slot.p; a.ps = v_ps.p; a.g3 = g3.p; a.op_partial = op_partial.p;
    a.den_slots = st.slots(k, DEN); a.done = st.done + k; a.lm = lm; a.stream = stream;
    hook_begin();
    hook_end(model->op(model->ctx, &a), "op");
  }
  void launch_operator(PcgState st, int k, const T *rec, const LmDev *lm = nullptr, double mu = 0.0) {
    if (model) { launch_operator_model(st, k, lm); return; }
    if constexpr (sizeof(T) == 8) { if (jac32) { launch_operator_j<float>(st, k, rec, lm, mu); return; } }
    launch_operator_j<T>(st, k, rec, lm, mu);
  }
  // k_pcg_update is persistent: a grid beyond what is resident at once runs its excess workgroups as a second, nearly empty round
  // (the single-reduction form takes 73 VGPRs in fp64 = 6 workgroups per CU, the first-iteration form 68 = 7, against the 8 the
  // common grid assumes: 512 of a Final-13682 shard's 2 048 workgroups waited for a slot).  Capped per kernel variant, once.
  template <int MODE, bool IDENTITY, int LZ> int update_grid(int blocks) {
    static thread_local int per_cu = 0; // one per instantiation (= kernel variant); the same for every problem of the process
    if (per_cu == 0) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_pcg_update<T, MODE, IDENTITY, LZ>), TPB, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 4; }
      per_cu = std::min(nb, 8);
    }
    int b = std::min(blocks, num_cu * per_cu);
    if (g3_obs_order) b = std::max(8, b / 8 * 8); // the point sweep deals workgroups to XCDs by blockIdx % 8
    return std::max(1, b);
  }
  template <int MODE, bool IDENTITY> void launch_update(int blocks, T *x, const T *rawc, int cw, int ui, PcgState st, int k, int nc = -1, int np = -1, const LmDev *lm = nullptr, bool first_lazy = false) {
    const int nc_v = nc < 0 ? (int)Nc : nc, np_v = np < 0 ? (int)Np : np;
#define GR_UPD(LZ, LM_P, FZ, OFF) launch(k_pcg_update<T, MODE, IDENTITY, LZ>, update_grid<MODE, IDENTITY, LZ>(blocks), nc_v, np_v, bu.p, scales.p, x, v_xb.p, v_r.p, v_z.p, v_p.p, g3.p, pt_ptr.p, op_partial.p, cam_seg_ptr.p, rawc, cw, v_diag.p, damping, ui, MinvC.p, MinvP.p, st, k, LM_P, g3_gather(), FZ, OFF)
    if (st.lazy == 2) {
      const bool fz_on = MODE == 1 && shard_fused();
      GR_UPD(2, nullptr, fz_on ? fused_fz() : IpcFused{}, (unsigned long long)shard_dots_off());
    }
    else if (st.lazy) GR_UPD(1, nullptr, IpcFused{}, 0ull);
    else if (first_lazy) GR_UPD(3, lm, IpcFused{}, 0ull);
    else GR_UPD(0, lm, IpcFused{}, 0ull);
#undef GR_UPD
  }
  // bytes one matrix-free operator launch has to move at minimum (J recomputed, every array
  // touched once): obs + 3 index streams, ps, packs, points, g3 out, segment partials
  double operator_bytes() const {
    if (model && model->store_jacobians) // the stored weighted Jacobian + 3 index streams, ps, g3 out, segment partials
      return No * ((2.0 * std::max(6, (int)model->pose_dim == 7 ? 7 : ((int)model->pose_dim <= 6 ? 6 : 9)) + 6.0) * (model->storage_dtype == GR_F64 ? 8.0 : 4.0) + 12.0) + n * (double)sizeof(T) + 3.0 * No * sizeof(T) + 9.0 * nseg * sizeof(T);
    return No * (2.0 * sizeof(T) + 12.0) + (n + 24.0 * Nc + 3.0 * Np) * sizeof(T) + 3.0 * No * sizeof(T) + 9.0 * nseg * sizeof(T);
  }
  void ensure_ctl(int max_iter) {
    const int cap = max_iter + 2;
    if (cap > ctl_cap) { ctl_cap = cap; ctl.alloc(((size_t)NSLOT * NSW + 4) * cap); ctl_i.alloc(cap); }
    alloc_pinned(cap);
  }
  PcgState pcg_state() {
    PcgState st;
    const size_t blk = (size_t)ctl_cap * NSLOT * NSW;
    st.acc = ctl.p; st.pdp = ctl.p + blk; st.rz0 = ctl.p + blk + ctl_cap;
    st.done = ctl_i.p; st.iters = pcg_iters.p; st.hflag = flags(); st.hiters = h_iters();
    loop_left.alloc(1); st.left = loop_left.p;
    st.beta = ctl.p + blk + 2 * (size_t)ctl_cap; st.scale = ctl.p + blk + 3 * (size_t)ctl_cap;
    st.lazy = pcg_mode();
    if (st.lazy) v_zs.alloc(n);
    if (st.lazy == 2) v_sv.alloc(n);
    st.ps = v_ps.p; st.zs = v_zs.p; st.sv = v_sv.p;
    st.zrec = (st.lazy == 2 && use_records && xp.n == 8 * (size_t)Np) ? xp.p : nullptr;
    st.x = nullptr; st.xb = v_xb.p; st.n = (unsigned)n; st.tol = 0.0; st.rej = 1e30; // set per solve (solve_pcg)
    st.ts = (lm_fused && h_ts) ? h_ts + 2 * ts_slot : nullptr;
    return st;
  }
  // PCGSolver::solve (solver/pcg.hpp:61-232).  Scalars stay on the device; the host only
  // watches a pinned flag per iteration so that it stops enqueueing once the loop has left
  // (one iteration of look-ahead keeps the queue full).
  template <bool IDENTITY> void solve_pcg(int max_iter, double tol, double rej, T *x) {
    const int cap_before = ctl_cap;
    ensure_ctl(max_iter);
    const bool started = update0_done && ctl_cap == cap_before && x == lm_x && update0_identity == IDENTITY; // by k_block_jacobi
    update0_done = false;
    state_clean_cap = -1;
    ensure_point_records();
    T *rec = use_records ? xp.p : nullptr;
    PcgState st = pcg_state();
    st.x = x; st.tol = tol; st.rej = rej;
    const int ui = damping_identity ? 1 : 0;
    for (int k = 0; k < max_iter + 1; ++k) flags()[k] = 0;
    *h_iters() = 0;
    if (!started && state_fresh_cap != ctl_cap) k_pcg_state_init<<<1, TPB, 0, stream>>>(st, ctl_cap); // else reset by k_block_jacobi
    state_fresh_cap = -1;
    const int ublocks = update_blocks();
    const T *rawc = comm ? raw_c.p : nullptr;
    const int cw = cam_weight();
    if (!started) launch_update<0, IDENTITY>(ublocks, x, rawc, cw, ui, st, 0);
    if (comm && st.lazy != 2) allreduce_d(st.acc, 4 * (size_t)NSW); // record 0: RZP, RR, PDZ, ZDZ (single-reduction form: rides in the first iteration's message)
    auto enqueue_operator = [&](int k) {
      Scope s1(this, "pcg_operator", operator_bytes(), No * 340.0, true);
      launch_operator(st, k, rec, nullptr, damping);
    };
    auto enqueue_update = [&](int k) {
      if (comm) { // camera rows + the p.A.p partials, summed over the landmark shards
        k_cam_rows<T><<<cdiv(pose_dim, TPB), TPB, 0, stream>>>((int)Nc, cam_seg_ptr.p, op_partial.p, raw_c.p, nullptr, k);
        group_start();
        allreduce_T(raw_c.p, pose_dim);
        allreduce_d(st.acc + ((size_t)k * NSLOT + DEN) * NSW, NSW);
        group_end();
      }
      {
        Scope s2(this, "pcg_update", 9.0 * n * sizeof(T) + (81.0 * Nc + 9.0 * Np) * sizeof(T) + 9.0 * nseg * sizeof(T) + 3.0 * No * sizeof(T), 16.0 * n + 2.0 * (81.0 * Nc + 9.0 * Np), true);
        launch_update<1, IDENTITY>(ublocks, x, rawc, cw, ui, st, k);
      }
      if (comm) allreduce_d(st.acc + (size_t)(k + 1) * NSLOT * NSW, 4 * (size_t)NSW);
    };
    if (st.lazy == 2) {
      // SINGLE-REDUCTION form: operator k on z', ONE grouped all-reduce (camera rows + the five dot records of iteration
      // k), update k, whose prologue decides about iteration k - 1 and raises its host flag.  The record-0 dots of the
      // start ride in the first message.  Reaching the iteration cap costs one more (scalar) all-reduce + a closing launch.
      bool left = false;
      int hook_at = -1, ran = 0;
      trial_done = false;
      const bool fz_on = comm && shard_fused();
      auto iteration = [&](int k) {
        enqueue_operator(k);
        if (fz_on) { ++coll_count; ++fused_messages; } // the operator launch pushed the message, the update launch awaits it: no kernel in between
        else if (comm) {
          k_cam_rows<T><<<cdiv(pose_dim, TPB), TPB, 0, stream>>>((int)Nc, cam_seg_ptr.p, op_partial.p, raw_c.p, nullptr, k);
          ++launch_count;
          group_start();
          allreduce_T(raw_c.p, pose_dim);
          allreduce_d(st.acc + (size_t)k * NSLOT * NSW, (size_t)NSLOT * NSW); // RZP, RR, PDZ, ZDZ, DEN of record k
          group_end();
        }
        Scope s2(this, "pcg_update", 9.0 * n * sizeof(T) + (81.0 * Nc + 9.0 * Np) * sizeof(T) + 9.0 * nseg * sizeof(T) + 3.0 * No * sizeof(T), 16.0 * n + 2.0 * (81.0 * Nc + 9.0 * Np), true);
        launch_update<1, IDENTITY>(ublocks, x, rawc, cw, ui, st, k);
      };
      int enq = 0;
      if (max_iter > 0) { iteration(0); ++enq; }
      for (int k = 0; k < max_iter; ++k) { // waits for the flag of iteration k, raised by update k + 1 (or the closing launch)
        if (k + 1 < max_iter) { iteration(k + 1); ++enq; }
        else {
          if (comm) allreduce_d(st.acc + (size_t)(k + 1) * NSLOT * NSW, 4 * (size_t)NSW);
          k_pcg_close_cg<T><<<64, TPB, 0, stream>>>(st, k + 1, damping);
        }
        if (trial_hook && hook_at < 0 && predicted_iters > 0 && k + 1 >= predicted_iters) {
          trial_hook(k + 1 < max_iter ? loop_left.p : nullptr);
          hook_at = k;
        }
        spin_until([&] { return __atomic_load_n(const_cast<const int *>(&flags()[k]), __ATOMIC_ACQUIRE) != 0; });
        ran = k + 1;
        if (flags()[k] == 2) { left = true; break; }
      }
      const int active = left ? std::min((int)*h_iters(), ran) : ran;
      if (hook_at >= 0) {
        trial_done = (left && ran - 1 <= hook_at) || hook_at + 1 == max_iter;
        if (!trial_done) note_noop({"linearize", "linearize_finalize"}, 1);
      }
      predicted_iters = std::max(active, 1);
      note_noop({"pcg_operator", "pcg_update"}, enq - active);
      return;
    }
    if (st.lazy) {
      // LAZY form: no direction kernel.  The decision that follows update k (and the host flag of iteration k) comes from
      // the prologue of operator k + 1, so the pair (update k, operator k + 1) is enqueued together and the host sees the
      // flag while that operator is still running: time enough to enqueue the next pair without a prediction.
      int enq_ops = 0;
      bool left = false;
      int hook_at = -1, ran = 0;
      trial_done = false;
      if (max_iter > 0) { enqueue_operator(0); ++enq_ops; }
      for (int k = 0; k < max_iter; ++k) {
        enqueue_update(k);
        if (k + 1 < max_iter) { enqueue_operator(k + 1); ++enq_ops; }
        else k_pcg_close<T><<<64, TPB, 0, stream>>>(st, k + 1);
        if (trial_hook && hook_at < 0 && predicted_iters > 0 && k + 1 >= predicted_iters) {
          trial_hook(k + 1 < max_iter ? loop_left.p : nullptr); // the iteration cap ends the loop whatever the flag says
          hook_at = k;
        }
        spin_until([&] { return __atomic_load_n(const_cast<const int *>(&flags()[k]), __ATOMIC_ACQUIRE) != 0; });
        ran = k + 1;
        if (flags()[k] == 2) { left = true; break; }
      }
      const int active = left ? std::min((int)*h_iters(), ran) : ran; // iterations whose operator and update did work
      if (hook_at >= 0) {
        trial_done = (left && ran - 1 <= hook_at) || hook_at + 1 == max_iter;
        if (!trial_done) note_noop({"linearize", "linearize_finalize"}, 1);
      }
      predicted_iters = std::max(active, 1);
      note_noop({"pcg_operator"}, enq_ops - active);
      note_noop({"pcg_update"}, ran - active);
      return;
    }
    launch(k_pcg_direction<T>, grid_vec, (unsigned)n, x, v_xb.p, v_p.p, v_ps.p, v_z.p, scales.p, st, -1, 0.0, 1e30, (unsigned)pose_dim, rec, nullptr, 0.0, ApplyOnExit<T>{});
    auto enqueue = [&](int k) {
      enqueue_operator(k);
      enqueue_update(k);
      {
        Scope s3(this, "pcg_direction", 5.0 * n * sizeof(T), 3.0 * n, true);
        launch(k_pcg_direction<T>, grid_vec, (unsigned)n, x, v_xb.p, v_p.p, v_ps.p, v_z.p, scales.p, st, k, tol, rej, (unsigned)pose_dim, rec, nullptr, 0.0, ApplyOnExit<T>{});
      }
    };
    note_noop({"pcg_operator", "pcg_update", "pcg_direction"}, run_pcg_iterations(max_iter, enqueue));
  }

  // Diagnostic: average device time (us) of `reps` back-to-back launches of one hot kernel
  // (which: 0 operator, 1 linearize, 2 chi2, 3 pcg_update, 4 pcg_direction, 5 linearize_finalize).
  bool diag_running = false; // diag_time is per-rank (the timed tuning choices): nothing collective may happen inside it
  double diag_time(int which, int variant, int reps) override {
    const bool was_tuned = records_tuned, was_tl = tiling_tuned;
    records_tuned = tiling_tuned = true; // no recursion through solver_update_structure
    // nothing collective in here: the timed choices are per rank, and ranks call this a different number of times (a rank that
    // stays in the plain order also times the g3 layouts) — the linearisation below keeps its camera-space sums rank-local
    struct Running { bool &f; explicit Running(bool &f_) : f(f_) { f = true; } ~Running() { f = false; } } running(diag_running);
    solver_update_structure(GR_SOLVER_PCG);
    records_tuned = was_tuned; tiling_tuned = was_tl;
    linearize_impl(false);
    solver_set_damping(GR_SOLVER_PCG, 1e-4, false);
    ensure_ctl(4);
    ensure_point_records();
    PcgState st = pcg_state();
    const int ui = 0;
    const int ublocks = update_blocks();
    k_pcg_state_init<<<1, TPB, 0, stream>>>(st, ctl_cap);
    st.x = v_dx.p;
    launch_update<0, false>(ublocks, v_dx.p, nullptr, 1, ui, st, 0);
    if (st.lazy) launch_operator(st, 0, nullptr, nullptr, damping); // publishes beta[0], scale[0], pdp[0]
    else k_pcg_direction<T><<<grid_vec, TPB, 0, stream>>>((unsigned)n, v_dx.p, v_xb.p, v_p.p, v_ps.p, v_z.p, scales.p, st, -1, 0.0, 1e30, (unsigned)pose_dim, use_records ? xp.p : nullptr);
    hipEvent_t a, b;
    GR_HIP(hipEventCreate(&a)); GR_HIP(hipEventCreate(&b));
    auto one_launch = [&] {
      switch (which) {
      case 0:
#define GR_OP(V) k_pcg_operator<T, V><<<grid_op, TPB, 0, stream>>>((int)No, (int)Nc, o_ntiles(), o_cam(), o_pt(), g3_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, v_ps.p, g3.p, op_partial.p, damping, st, 0, use_records ? xp.p : nullptr)
#ifdef GR_DIAG
        switch (variant) { case 1: GR_OP(1); break; case 2: GR_OP(2); break; case 4: GR_OP(4); break; case 7: GR_OP(7); break; case 8: GR_OP(8); break; case 15: GR_OP(15); break; case 16: GR_OP(16); break; case 31: GR_OP(31); break;
                           case 32: GR_OP(32); break; case 64: GR_OP(64); break; case 128: GR_OP(128); break; case 3: GR_OP(3); break; case 95: GR_OP(95); break; case 255: GR_OP(255); break; case 224: GR_OP(224); break; default: GR_OP(0); }
#else
        launch_operator(st, 0, use_records ? xp.p : nullptr, nullptr, damping);
#endif
        break;
      case 1:
        if (model) { launch_linearize_cam(false, g9.p, nullptr); break; }
#ifdef GR_DIAG
#define GR_LIN(V) k_linearize<T, false, T, V><<<grid_lin, TPB, 0, stream>>>((int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, g9.p, nullptr, cam_partial.p, chi2_partial.p)
        switch (variant) { case 1: GR_LIN(1); break; case 2: GR_LIN(2); break; case 3: GR_LIN(3); break; case 4: GR_LIN(4); break; case 7: GR_LIN(7); break; case 8: GR_LIN(8); break; case 15: GR_LIN(15); break; default: GR_LIN(0); }
#else
        k_linearize<T, false><<<grid_lin, TPB, 0, stream>>>((int)No, o_ntiles(), o_cam(), o_pt(), o_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, g9.p, nullptr, cam_partial.p, chi2_partial.p);
#endif
        break;
      case 2: chi2_async(nullptr, variant ? v_dx.p : nullptr, 1e-4); break;
      case 3: { // variant 1: camera part only, 2: point part only, >= 8: that many blocks
        const int nc_v = variant == 2 ? 0 : (int)Nc, np_v = variant == 1 ? 0 : (int)Np;
        int ub = variant >= 8 ? variant : std::max(1, std::min(cdiv(9 * (size_t)nc_v, 252) + cdiv(np_v, 85), num_cu * 8));
        if (g3_obs_order) ub = std::max(8, (ub + 7) / 8 * 8); // the point sweep deals workgroups to XCDs by blockIdx % 8
        if (variant == 100) { ub = update_blocks(); launch_update<1, false>(ub, v_dx.p, nullptr, 1, ui, st, 0, (int)Nc, (int)Np, nullptr, /*first_lazy=*/true); break; } // the first-iteration form (LAZY = 3)
        launch_update<1, false>(ub, v_dx.p, nullptr, 1, ui, st, 0, nc_v, np_v);
        break;
      }
      case 4: k_pcg_direction<T><<<grid_vec, TPB, 0, stream>>>((unsigned)n, v_dx.p, v_xb.p, v_p.p, v_ps.p, v_z.p, scales.p, st, -1, 0.0, 1e30, (unsigned)pose_dim, use_records ? xp.p : nullptr); break;
      case 5: { // variant 1: camera part (+ chi2 block) only, 2: point part only (the blocks of the other part return at once)
        const int nc_v = variant == 2 ? 0 : (int)Nc, np_v = variant == 1 ? 0 : (int)Np;
        k_linearize_finalize<T><<<cdiv(90 * (size_t)nc_v, TPB) + cdiv(FIN_PL * (size_t)np_v, TPB), TPB, 0, stream>>>(nc_v, np_v, 1, 1, cam_seg_ptr.p, cam_partial.p, pt_ptr.p, g9.p, Hcc.p, bc.p, Hll.p, bl.p, scales.p, grid_lin, chi2_partial.p, variant == 2 ? nullptr : dscalars.p);
        break;
      }
      case 6: { // block-Jacobi inverses: variant 1 cameras only, 2 points only, 3 with the fused PCG start
        const int nbc = variant == 2 ? 0 : cdiv(Nc, 64), nbp = variant == 1 ? 0 : cdiv(Np, 64);
        if (variant == 3) k_block_jacobi<T><<<nbc + nbp, 64, 0, stream>>>((int)Nc, (int)Np, nbc, nbp, Hcc.p, Hll.p, scales.p, 1e-4, 0, MinvC.p, MinvP.p, v_diag.p, st, ctl_cap, nullptr, bu.p, v_dx.p, v_r.p, v_z.p, 0);
        else k_block_jacobi<T><<<nbc + nbp + 1, 64, 0, stream>>>(variant == 2 ? 0 : (int)Nc, variant == 1 ? 0 : (int)Np, nbc, nbp, Hcc.p, Hll.p, scales.p, 1e-4, 0, MinvC.p, MinvP.p, v_diag.p, st, ctl_cap);
        break;
      }
      case 7: linearize_impl(false, /*pack_valid=*/true); break; // the whole linearisation (all its kernels), current path
      case 9: { // k_finalize_bj: variant bit 0 cameras only, bit 1 points only, bit 2 with the decision prologue, bit 3 one point tile per workgroup
        const int nc_v = (variant & 2) ? 0 : (int)Nc, np_v = (variant & 1) ? 0 : (int)Np;
        const int nbc = cdiv(nc_v, 28), nbp = np_v ? ((variant & 8) ? cdiv(np_v, TPB / FIN_PL) : std::max(1, std::min(cdiv(np_v, TPB / FIN_PL), num_cu * 4))) : 0;
        LmDecide dec;
        if (variant & 4) {
          ensure_lm_buffers();
          rho_partial.alloc(dir_grid());
          dec.seq = 1; dec.chi2_cur = 1e30; dec.mu_cur = 1e-4; dec.chi2_partial = chi2_partial.p; dec.n_chi2 = grid_lin; dec.rho_partial = rho_partial.p; dec.n_rho = dir_grid();
          dec.hres = h_res; dec.hres_seq = h_seq; dec.lm = lmdev.p; dec.dscal = dscalars.p;
        }
#define GR_FBJ(V) k_finalize_bj<T, V><<<nbc + nbp, TPB, 0, stream>>>(nc_v, np_v, nbc, 1, cam_seg_ptr.p, cam_partial.p, pt_ptr.p, g9.p, Hcc.p, bu.p, Hll.p, scales.p, 1e-4, 0, MinvC.p, MinvP.p, v_diag.p, st, v_dx.p, v_r.p, v_z.p, nullptr, 0, 1, dec, nullptr, nullptr)
#ifdef GR_DIAG
        switch (variant >> 4) { case 1: GR_FBJ(1); break; case 2: GR_FBJ(2); break; case 4: GR_FBJ(4); break; case 8: GR_FBJ(8); break; case 3: GR_FBJ(3); break; case 7: GR_FBJ(7); break; case 15: GR_FBJ(15); break; default: GR_FBJ(0); }
#else
        GR_FBJ(0);
#endif
#undef GR_FBJ
        break;
      }
      case 8: // one all-reduce of `variant` values (0: a camera-space vector) through the communicator, every rank calls it
        if (!comm) throw std::invalid_argument("diag_time(8): no communicator");
        comm->allreduce(tmp.p, variant > 0 ? std::min<size_t>(variant, tmp.n) : pose_dim, sizeof(T) == 8, stream);
        break;
      default: throw std::invalid_argument("diag_time: unknown kernel");
      }
    };
    if (which == 8) { ensure_tmp(); tmp.zero(stream); }
    for (int i = 0; i < 3; ++i) one_launch();
    GR_HIP(hipEventRecord(a, stream));
    for (int i = 0; i < reps; ++i) one_launch();
    GR_HIP(hipEventRecord(b, stream));
    GR_HIP(hipEventSynchronize(b));
    float ms = 0;
    GR_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms * 1e3 / reps;
  }

  int last_solver = 0;
  int last_iters() {
    int it = 0;
    if (last_solver == GR_SOLVER_DENSE_SCHUR) return 0;
    const int *src = (last_solver == GR_SOLVER_PCG_SCHUR || last_solver == GR_SOLVER_PCG_SCHUR_IMPLICIT) ? sc_i.p + sc_cap : pcg_iters.p;
    GR_HIP(hipMemcpyAsync(&it, src, sizeof(int), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
    return it;
  }
  bool solver_solve_dev(int solver, int max_iter, double tol, double rej, T *x) {
    last_solver = solver;
    switch (solver) {
    case GR_SOLVER_PCG_SCHUR: solve_pcg_schur(max_iter, tol, rej, x); return true;
    case GR_SOLVER_PCG: solve_pcg<false>(max_iter, tol, rej, x); return true;
    case GR_SOLVER_PCG_IDENTITY: solve_pcg<true>(max_iter, tol, rej, x); return true;
    case GR_SOLVER_PCG_SCHUR_IMPLICIT: solve_pcg_schur_implicit(max_iter, tol, rej, x); return true;
    case GR_SOLVER_DENSE_SCHUR: return solve_dense_schur(x);
    }
    throw std::invalid_argument("unknown solver");
  }
  bool solver_solve(int solver, int max_iter, double tol, double rej, void *dx, int *iters) override {
    const bool ok = solver_solve_dev(solver, max_iter, tol, rej, v_dx.p);
    GR_HIP(hipMemcpyAsync(dx, v_dx.p, pose_dim * sizeof(T), hipMemcpyDefault, stream));
    points_out(v_dx.p + pose_dim, static_cast<T *>(dx) + pose_dim, 3);
    const int it = last_iters();
    if (iters) *iters = it;
    return ok;
  }

  // ---- getters (scaled space, reference layouts) ------------------------------------
  void get(int which, void *out, int64_t *count) override {
    ensure_tmp();
    int64_t cnt = 0;
    const T *src = nullptr;
    std::vector<int> map;
    DevBuf<int> dmap, dmap2;
    auto launch_export = [&](size_t nb, int rows, int cols, const T *s, const T *srow, const T *scol, const int *rm, const int *cm, const int *sm) {
      k_export_blocks<T><<<cdiv(nb * rows * cols, TPB), TPB, 0, stream>>>(nb, rows, cols, s, srow, scol, rm, cm, sm, tmp.p);
    };
    switch (which) {
    case GR_GET_SCALES: cnt = (int64_t)n; src = scales.p; break;
    case GR_GET_B:
      cnt = (int64_t)n;
      k_mul<T><<<cdiv(n, TPB), TPB, 0, stream>>>((unsigned)n, tmp.p, scales.p, bu.p);
      src = tmp.p; break;
    case GR_GET_HCC: cnt = 81 * Nc; launch_export(Nc, 9, 9, Hcc.p, scales.p, scales.p, nullptr, nullptr, nullptr); src = tmp.p; break;
    case GR_GET_HLL: cnt = 9 * Np; launch_export(Np, 3, 3, Hll.p, scales.p + pose_dim, scales.p + pose_dim, nullptr, nullptr, nullptr); src = tmp.p; break;
    case GR_GET_HCP: {
      if (!hcp_valid) { Hcp.alloc(27 * (size_t)No); linearize_impl(true); }
      cnt = 27 * No;
      std::vector<int> rm(No), cmv(No);
      for (int64_t o = 0; o < No; ++o) { const int a = h_pm_of_orig[o]; rm[o] = h_cam_pm[a]; cmv[o] = h_pt_pm[a]; }
      dmap.upload(rm, stream); dmap2.upload(cmv, stream);
      DevBuf<int> dsrc; dsrc.upload(h_pm_of_orig, stream);
      launch_export(No, 9, 3, Hcp.p, scales.p, scales.p + pose_dim, dmap.p, dmap2.p, dsrc.p);
      GR_HIP(hipStreamSynchronize(stream));
      src = tmp.p; break;
    }
    case GR_GET_S: build_schur_structure(); cnt = 81 * nnzb; src = S.p; break;
    case GR_GET_B_SCHUR: build_schur_structure(); cnt = (int64_t)pose_dim; src = b_schur.p; break;
    case GR_GET_HLL_INV: build_schur_structure(); cnt = 9 * Np; src = Hll_inv.p; break;
    case GR_GET_H: {
      cnt = 81 * Nc + 27 * No + 9 * Np;
      if (out) { const std::vector<T> v = hessian_values(); GR_HIP(hipMemcpy(out, v.data(), v.size() * sizeof(T), hipMemcpyDefault)); }
      if (count) *count = cnt;
      return;
    }
    case GR_GET_RESIDUALS: {
      cnt = 2 * No;
      DevBuf<T> rpm; rpm.alloc(2 * (size_t)No);
      campack();
      wait_chi2(chi2_async(rpm.p, nullptr, 0.0));
      std::vector<T> h = rpm.download(stream), ho(2 * (size_t)No);
      for (int64_t o = 0; o < No; ++o) { ho[2 * o] = h[2 * (size_t)h_pm_of_orig[o]]; ho[2 * o + 1] = h[2 * (size_t)h_pm_of_orig[o] + 1]; }
      if (out) GR_HIP(hipMemcpy(out, ho.data(), ho.size() * sizeof(T), hipMemcpyDefault));
      if (count) *count = cnt;
      return;
    }
    default: throw std::invalid_argument("unknown array id");
    }
    if (count) *count = cnt;
    if (out && cnt) {
      if (which == GR_GET_SCALES || which == GR_GET_B) { // n-vectors: camera part as is, point part re-ordered
        GR_HIP(hipMemcpyAsync(out, src, pose_dim * sizeof(T), hipMemcpyDefault, stream));
        points_out(src + pose_dim, static_cast<T *>(out) + pose_dim, 3);
      } else if (which == GR_GET_HLL || which == GR_GET_HLL_INV) points_out(src, out, 9);
      else {
        GR_HIP(hipMemcpyAsync(out, src, (size_t)cnt * sizeof(T), hipMemcpyDefault, stream));
        GR_HIP(hipStreamSynchronize(stream));
      }
    }
  }

  // ---- Hessian in the reference's layouts (hessian.hpp:257-324, csc_utils.hpp:16-193): export API, host side ----------
  // block columns = cameras, then points in the CALLER's order; per point column the Hcp blocks by ascending camera, Hll last
  void hessian_structure(int64_t *nblocks, int64_t *colptr, int64_t *rowidx, int64_t *offsets) override {
    if (nblocks) *nblocks = Nc + No + Np;
    if (!colptr && !rowidx && !offsets) return;
    int64_t blk = 0, off = 0;
    for (int64_t c = 0; c < Nc; ++c) {
      if (colptr) colptr[c] = blk;
      if (rowidx) rowidx[blk] = c;
      if (offsets) offsets[blk] = off;
      ++blk; off += 81;
    }
    for (int64_t l = 0; l < Np; ++l) {
      const int q = h_pt_old2new[l];
      if (colptr) colptr[Nc + l] = blk;
      for (int a = h_pt_ptr[q]; a < h_pt_ptr[q + 1]; ++a) {
        if (rowidx) rowidx[blk] = h_cam_pm[a];
        if (offsets) offsets[blk] = off;
        ++blk; off += 27;
      }
      if (rowidx) rowidx[blk] = Nc + l;
      if (offsets) offsets[blk] = off;
      ++blk; off += 9;
    }
    if (colptr) colptr[Nc + Np] = blk;
  }
  std::vector<T> hessian_values() { // GR_GET_H, assembled on the host from the three scaled block arrays
    std::vector<T> hcc(81 * (size_t)Nc), hll(9 * (size_t)Np), hcp(27 * (size_t)No), v;
    int64_t cnt = 0;
    get(GR_GET_HCC, hcc.data(), &cnt); get(GR_GET_HLL, hll.data(), &cnt); get(GR_GET_HCP, hcp.data(), &cnt); // Hcp: INPUT observation order
    std::vector<int> orig_of_pm(No);
    for (int64_t o = 0; o < No; ++o) orig_of_pm[h_pm_of_orig[o]] = (int)o;
    v.reserve(81 * (size_t)Nc + 27 * (size_t)No + 9 * (size_t)Np);
    v.insert(v.end(), hcc.begin(), hcc.end());
    for (int64_t l = 0; l < Np; ++l) {
      const int q = h_pt_old2new[l];
      for (int a = h_pt_ptr[q]; a < h_pt_ptr[q + 1]; ++a) { const T *b = &hcp[27 * (size_t)orig_of_pm[a]]; v.insert(v.end(), b, b + 27); }
      v.insert(v.end(), &hll[9 * (size_t)l], &hll[9 * (size_t)l] + 9);
    }
    return v;
  }
  void export_csc(int which, int64_t *nnz, int64_t *indptr, int64_t *indices, void *values) override {
    // scalar upper CSC from a block-CSC (csc_utils.hpp:74-193): per scalar column the rows <= column of every block of its block column
    std::vector<int64_t> bcolptr, browidx, boffsets, soff;
    std::vector<T> bvalues;
    int64_t nbcol = 0;
    if (which == 0) {
      nbcol = Nc + Np;
      bcolptr.resize(nbcol + 1); browidx.resize(Nc + No + Np); boffsets.resize(Nc + No + Np);
      hessian_structure(nullptr, bcolptr.data(), browidx.data(), boffsets.data());
      if (values) bvalues = hessian_values();
      soff.resize(nbcol + 1);
      for (int64_t b = 0; b <= nbcol; ++b) soff[b] = b <= Nc ? 9 * b : 9 * Nc + 3 * (b - Nc);
    } else if (which == 1) {
      build_schur_structure();
      nbcol = Nc;
      bcolptr.assign(h_S_colptr.begin(), h_S_colptr.end()); browidx.assign(h_S_rowi.begin(), h_S_rowi.end());
      boffsets.resize(nnzb);
      for (int64_t q = 0; q < nnzb; ++q) boffsets[q] = 81 * q;
      if (values) bvalues = S.download(stream);
      soff.resize(nbcol + 1);
      for (int64_t b = 0; b <= nbcol; ++b) soff[b] = 9 * b;
    } else throw std::invalid_argument("gr_bal_export_csc: which = 0 (H) or 1 (S)");
    int64_t count = 0;
    T *vout = static_cast<T *>(values);
    for (int64_t bc = 0; bc < nbcol; ++bc) {
      const int64_t ncols = soff[bc + 1] - soff[bc];
      for (int64_t cin = 0; cin < ncols; ++cin) {
        const int64_t col = soff[bc] + cin;
        if (indptr) indptr[col] = count;
        for (int64_t b = bcolptr[bc]; b < bcolptr[bc + 1]; ++b) {
          const int64_t br = browidx[b], nr = soff[br + 1] - soff[br], r0 = soff[br];
          for (int64_t r = 0; r < nr && r0 + r <= col; ++r, ++count) {
            if (indices) indices[count] = r0 + r;
            if (vout) vout[count] = bvalues[boffsets[b] + cin * nr + r];
          }
        }
      }
    }
    if (indptr) indptr[soff[nbcol]] = count;
    if (nnz) *nnz = count;
  }

  // ---- device-decided LM iteration of PCGSchurSolver on small reduced systems (kernels_sf.hpp) ------------------------------------
  DevBuf<unsigned> coop_bar;
  DevBuf<double> coop_part;
  DevBuf<double> coop_recs;   // k_schur_pcg_coop's flagged records (coop_exchange): [2][Nc + 2] x 16 bytes
  unsigned coop_seq = 0;      // launches of the cooperative PCG so far: the records' tag
  DevBuf<int> coop_iters, coop_fail_dev;
  volatile int *h_coop_fail = nullptr; // pinned, sticky
  bool sf_active = false;              // armed by lm()
  // ... and in that form (built-in model, gr_bal_tuning.schur_fused = 2 / auto) a rejected step does not stop the head: RejectCont (kernels_sf.hpp)
  bool sf_cont = false;
  double sf_nu = 2.0;                  // the factor the NEXT rejection multiplies the damping with (the host's nu)
  DevBuf<T> Hcp1;                      // second buffer of camera-point blocks (LmDev::hsel)
  DevBuf<double> sf_vsum;              // per-point sums of the current linearisation
  bool schur_fused_ok(int max_iter) const { return !comm && max_iter >= 1 && tune.schur_fused != 0 && tune.lm_fused != 0; }
  // the whole PCG on S in one cooperative launch: one wave per camera row, all of them resident at once — checked against the
  // launch's own residency (occupancy query x CUs), not assumed from the camera count; a handle whose barrier has
  // timed out once (CUs held by another stream or process) keeps the per-iteration launches from then on (sf_disabled)
  bool sf_disabled = false;
  int coop_capacity = -1;
  int coop_test_timeouts = 0; // GR_TEST_COOP_TIMEOUT = N: the first N cooperative launches of this handle wait for a workgroup that is not there
  bool schur_coop() {
    if (sf_disabled || Nc > 2 * (int64_t)num_cu) return false;
    if (coop_capacity < 0) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_schur_pcg_coop<T>), 64, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 0; }
      coop_capacity = nb * num_cu;
      coop_test_timeouts = env_int("GR_TEST_COOP_TIMEOUT", 0);
    }
    return Nc <= (int64_t)coop_capacity;
  }
  void ensure_coop(int max_iter) {
    coop_bar.alloc(1); coop_iters.alloc(1);
    if (!coop_fail_dev.n) { coop_fail_dev.alloc(1); coop_fail_dev.zero(stream); }
    coop_part.alloc((3 * (size_t)max_iter + 4) * (size_t)Nc);
    if (!coop_recs.n) { coop_recs.alloc(2 * 2 * ((size_t)Nc + 2)); coop_recs.zero(stream); }
    if (!h_coop_fail) {
      void *q = nullptr;
      GR_HIP(hipHostMalloc(&q, 64, hipHostMallocCoherent | hipHostMallocMapped));
      std::memset(q, 0, 64);
      h_coop_fail = static_cast<volatile int *>(q);
    }
    alloc_pinned(max_iter + 2);
  }
  // finalisation (+ accept decision of the pending trial step), S and b_S partials, the whole PCG on S, back-substitution + trial step
  void enqueue_schur_head(LmDecide dec, double mu, bool use_identity, int max_iter, double tol, double rej, int time_slot) {
    ts_slot = time_slot;
    flag_bank ^= 1;
    *h_iters() = 0;
    damping = mu; damping_identity = use_identity;
    const LmDev *lm = (dec.seq && !dec.report_only) ? lmdev.p : nullptr;
    const int ui = use_identity ? 1 : 0;
    CoopState cs{};
    cs.barrier = coop_bar.p; cs.part = coop_part.p; cs.iters = coop_iters.p; cs.hiters = h_iters(); cs.fail = h_coop_fail; cs.fail_dev = coop_fail_dev.p; cs.ts = h_ts ? h_ts + 2 * ts_slot : nullptr;
    cs.recs = coop_recs.p; cs.rec_cap = (int)Nc + 2; cs.launch_id = ++coop_seq;
    if (coop_test_timeouts > 0) { --coop_test_timeouts; cs.absent = 1; }
    const int nbc = cdiv(Nc, 28), nbp = std::max(1, std::min(cdiv(Np, TPB / FIN_PL), num_cu * 4));
    const bool coop = schur_coop();
    if (!coop) { ensure_scalars(max_iter); for (int k = 0; k < max_iter + 1; ++k) flags()[k] = 0; }
    {
      Scope sc(this, "finalize_schur", 8.0 * No * w() + 54.0 * nseg * w() + (90.0 * Nc + 36.0 * Np) * w(), 9.0 * No + 54.0 * nseg + 120.0 * Np, true);
      RejectCont<T> rc{};
      if (sf_cont) {
        if (model) rc.model_cont = 1; // (the user's vertices: restored by gr_model_ops.step, LmDev::hsel bit 1)
        else { rc.cams = cams.p; rc.pts = pts.p; rc.cams_bak = cams_bak.p; rc.pts_bak = pts_bak.p; rc.pack = pack.p; }
        rc.vsum = sf_vsum.p; rc.lm = lmdev.p; rc.nu_cur = sf_nu;
      }
      launch(k_finalize_schur<T>, nbc + nbp + (coop ? 0 : 1), (int)Nc, (int)Np, nbc, scale_system ? 1 : 0, cam_seg_ptr.p, cam_partial.p, pt_ptr.p, g9.p, Hcc.p, bu.p, Hll.p, scales.p, mu, ui,
             Hll_inv.p, Mp.p, vl.p, dec, cam_fixed_p(), pt_fixed_p(), cs, coop ? PcgScalars{} : scalars(), coop ? 0 : sc_cap, rc);
    }
    launch_schur_reduce(Hcc.p, mu, ui, lm);
    if (!coop) {
      // large reduced system: the per-iteration PCG kernels, each gated on the decision; the host follows the exit flags as before.
      // solve_seconds: HIP events only with options->profile (an event between two launches is a bubble)
      PcgScalars sc = scalars();
      k_schur_pcg_prepare<T, 0><<<cdiv(Nc, 64), 64, 0, stream>>>((int)Nc, S.p, S_diag.p, cam_chunk_ptr.p, part9.p, bc.p, scales.p, b_schur.p, MinvS.p, v_r.p, v_z.p, v_p.p, v_dx.p, nullptr, sc, lm);
      ++launch_count;
      const int noop = run_pcg_iterations(max_iter, [&](int k) {
        { Scope s1(this, "schur_matvec", (2.0 * nnzb - Nc) * 81.0 * w(), (2.0 * nnzb - Nc) * 162.0);
          k_schur_matvec<T><<<cdiv(Nc, 4), TPB, 0, stream>>>((int)Nc, row_ptr.p, row_blk.p, row_col.p, S.p, v_p.p, v_Ap.p, sc, k, lm); }
        k_pcgs_update<T><<<cdiv(pose_dim, 252), TPB, 0, stream>>>((int)Nc, v_dx.p, v_xb.p, v_r.p, v_z.p, v_p.p, v_Ap.p, MinvS.p, sc, k, lm);
        k_pcgs_direction<T><<<cdiv(pose_dim, TPB), TPB, 0, stream>>>((int)Nc, v_dx.p, v_xb.p, v_p.p, v_z.p, nullptr, nullptr, sc, k, tol, rej, lm);
        launch_count += 3;
      });
      note_noop({"schur_matvec"}, noop);
    } else
    {
      Scope sc(this, "schur_pcg_coop", (2.0 * nnzb - Nc) * 81.0 * w(), (2.0 * nnzb - Nc) * 162.0, true);
      ++launch_count;
      hipEvent_t ea = ext_a, eb = ext_b;
      ext_a = ext_b = nullptr;
      if (ea) hipExtLaunchKernelGGL(k_schur_pcg_coop<T>, dim3((unsigned)Nc), dim3(64), 0, stream, ea, eb, 0, (int)Nc, row_ptr.p, row_blk.p, row_col.p, S.p, S_diag.p, cam_chunk_ptr.p, part9.p, bc.p, scales.p, b_schur.p,
                                    v_p.p, v_dx.p, max_iter, tol, rej, cs, lm);
      else hipLaunchKernelGGL(k_schur_pcg_coop<T>, dim3((unsigned)Nc), dim3(64), 0, stream, (int)Nc, row_ptr.p, row_blk.p, row_col.p, S.p, S_diag.p, cam_chunk_ptr.p, part9.p, bc.p, scales.p, b_schur.p,
                              v_p.p, v_dx.p, max_iter, tol, rej, cs, lm);
    }
    {
      const int nct = cdiv(pose_dim, 252);
      rho_blocks = model ? model->step_blocks : nct + nbp;
      rho_partial.alloc(rho_blocks);
      Scope sc(this, "backsub_apply", No * (27.0 * w() + 4) + (9.0 * Np + 3.0 * n + 24.0 * Nc) * w(), No * 54.0, true);
      if (model) launch(k_backsub_apply<T, false>, nct + nbp, (int)Nc, (int)Np, nct, pt_ptr.p, cam_pm.p, Hcp.p, Hll_inv.p, bu.p, scales.p, v_dx.p, nullptr, nullptr, nullptr, nullptr, nullptr, mu, nullptr, lm, sf_cont ? Hcp1.p : nullptr, sf_cont ? lmdev.p : nullptr);
      else launch(k_backsub_apply<T>, rho_blocks, (int)Nc, (int)Np, nct, pt_ptr.p, cam_pm.p, Hcp.p, Hll_inv.p, bu.p, scales.p, v_dx.p, cams.p, pts.p, cams_bak.p, pts_bak.p, pack.p, mu, rho_partial.p, lm, sf_cont ? Hcp1.p : nullptr, sf_cont ? lmdev.p : nullptr);
    }
    // user-traits problems: the trial step through Traits::update, under the same decision (the damping of its rho partials from LmDev)
    if (model) model_step(v_dx.p, /*with_backup=*/true, mu, rho_partial.p, lm, nullptr);
    xp_valid = false;
  }
  // the trial linearisation of the Schur solvers: k_linearize with the camera-point blocks, finalised by the next head
  void linearize_hcp_deferred() {
    const double bytes = No * (2 * w() + 12.0) + (24.0 * Nc + 3.0 * Np) * w() + 8.0 * No * w() + 54.0 * nseg * w() + 27.0 * No * w() + model_jac_bytes();
    Scope sc(this, "linearize_hcp", bytes, No * (250.0 + 48 + 117 + 81.0), true);
    launch_linearize_cam(true, g9.p, nullptr);
    hcp_valid = true;
  }

  // ---- optimizer::levenberg_marquardt (optimizer/levenberg_marquardt.hpp:110-242) -----
  DevBuf<LmDev> lmdev;

  // ---- LM loop on the matrix-free PCG in its direction-kernel form: fused head, fused trial step ---------------------------
  // One accepted LM iteration = k_finalize_bj (finalisation of the linearisation + block-Jacobi + PCG start + the accept
  // decision about the PREVIOUS trial step), first direction, k x (operator, update, direction) — the direction launch that
  // ends the PCG loop applies the trial step itself (ApplyOnExit) — and k_linearize at the trial point.  On an accept streak
  // the head of iteration i + 1 (k_finalize_bj, first direction, first PCG iteration) is enqueued while k_linearize of
  // iteration i is still running; the device takes the accept decision, the host only observes it (LmDev, LmDecide).
  long long *h_ts = nullptr;    // pinned: 4 (start, end) pairs of device wall-clock stamps around the PCG loops (solve_seconds without events)
  int ts_slot = 0;
  double wall_clock_hz = 1e8;
  void ensure_lm_buffers() {
    if (!h_ts) {
      void *q = nullptr;
      GR_HIP(hipHostMalloc(&q, 8 * sizeof(long long), hipHostMallocCoherent | hipHostMallocMapped));
      h_ts = static_cast<long long *>(q);
      std::memset(h_ts, 0, 8 * sizeof(long long));
      int khz = 0;
      if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) wall_clock_hz = 1e3 * khz;
    }
    // (GR_TEST_POISON_LMDEV: tests/test_engine_model.py fills the record with 0xff instead — no loop may depend on what the allocation held)
    if (!lmdev.n) { lmdev.alloc(1); GR_HIP(hipMemsetAsync(lmdev.p, env_int("GR_TEST_POISON_LMDEV", 0) ? 0xff : 0, sizeof(LmDev), stream)); }
  }
  DevBuf<double> fbj_part;      // PcgState::part0
  bool lm_fused = false;        // armed by lm() for GR_SOLVER_PCG / _IDENTITY without a communicator, PCG mode 0, max_iter >= 1
  bool fin_pending = false;     // k_linearize ran, its finalisation is still to come (k_finalize_bj, or flush_finalize)
  bool pcg_state_clean = false; // the PCG loop state has been cleared since the last solve (k_linearize's reset block)
  int dir_grid() const { return std::max(grid_vec, cdiv(pose_dim, 252) + 1); }
  ApplyOnExit<T> apply_args(int k, int max_iter) {
    ApplyOnExit<T> ap;
    if (!lm_fused) return ap;
    ap.at_cap = (k + 1 == max_iter) ? 1 : 0;
    if (model) return ap; // the loop-ending launch only closes the loop; the step is gr_model_ops.step's (fused_iteration)
    rho_blocks = dir_grid();
    rho_partial.alloc(rho_blocks);
    ap.cams = cams.p; ap.pts = pts.p; ap.cams_bak = cams_bak.p; ap.pts_bak = pts_bak.p; ap.bu = bu.p;
    ap.rho_partial = rho_partial.p; ap.pack = pack.p; ap.xp = (use_records && xp.n && xp_valid) ? xp.p : nullptr;
    ap.cam_weight = 1; ap.at_cap = (k + 1 == max_iter) ? 1 : 0;
    return ap;
  }
  // ---- resident PCG (kernels_rp.hpp): the whole inner solve + the trial step as ONE launch, one 512-thread workgroup per CU ----
  DevBuf<T> rp_rec6, rp_crec, rp_g4;
  DevBuf<unsigned> rp_bar;
  DevBuf<long long> rp_dbg; // GR_RP_DEBUG=1: phase stamps of the last resident launch (gr_test_rp_stamps)
  volatile int *h_rp_fail = nullptr; // pinned: a grid barrier of the resident launch timed out (workgroups not co-resident)
  bool lm_resident = false;          // armed by lm()
  bool rp_disabled = false;          // a barrier timed out once on this handle: the multi-launch form from then on
  int rp_fit = -1;                   // -1 not decided, 0 / 1 (structure, tuning and device dependent: re-decided by apply_tuning)
  int rp_grid() const { return num_cu & ~7; }
  bool resident_fits() {
    if (rp_fit >= 0) return rp_fit != 0;
    rp_fit = 0;
    if (model || comm || tiled || jac32 || tune.lm_fused == 2 || tune.lm_fused == 0) return false;
    const int G = rp_grid();
    if (G < 8 || No <= 0 || 4 * No >= (int64_t)1 << 29 || 6 * Np >= (int64_t)1 << 28) return false; // 32-bit byte offsets of the buffer accesses
    // the kernel's own partition, evaluated for every workgroup
    const int nblk = (int)((No + 63) >> 6), nb = G >> 3;
    const int cam_tiles = cdiv(pose_dim, 9 * RP_CT), ntile = cam_tiles + cdiv(Np, RP_PT);
    for (int b = 0; b < G; ++b) {
      const int x = b & 7, bi = b >> 3;
      const int x0 = (int)((long long)x * nblk / 8), x1 = (int)((long long)(x + 1) * nblk / 8);
      const int b0 = x0 + (int)((long long)bi * (x1 - x0) / nb), b1 = x0 + (int)((long long)(bi + 1) * (x1 - x0) / nb);
      if ((long long)(b1 - b0) * 64 > (long long)RP_RO * RTPB) return false;
      const int ut0 = (int)((long long)b * ntile / G), ut1 = (int)((long long)(b + 1) * ntile / G);
      if (ut1 - ut0 > RP_RV) return false;
    }
    // one workgroup per CU must be resident: LDS image + the occupancy query
    const void *fn = reinterpret_cast<const void *>(&k_pcg_resident<T, T>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RpLds<T>::bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    int nbk = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, fn, RTPB, RpLds<T>::bytes) != hipSuccess || nbk < 1) { (void)hipGetLastError(); return false; }
    if (!h_rp_fail) {
      void *q = nullptr;
      GR_HIP(hipHostMalloc(&q, 64, hipHostMallocCoherent | hipHostMallocMapped));
      std::memset(q, 0, 64);
      h_rp_fail = static_cast<volatile int *>(q);
    }
    rp_rec6.alloc(6 * (size_t)Np); rp_crec.alloc(18 * (size_t)Nc); rp_g4.alloc(4 * (size_t)No); rp_bar.alloc(RP_BAR_WORDS);
    GR_HIP(hipMemsetAsync(rp_bar.p, 0, sizeof(unsigned) * RP_BAR_WORDS, stream));
    rp_fit = 1;
    if (tune.verbose) std::fprintf(stderr, "[graphite-mi355x] resident PCG: %d workgroups x %d threads, %zu bytes of LDS each\n", G, RTPB, (size_t)RpLds<T>::bytes);
    return true;
  }
  // auto = OFF: measured on Ladybug-1723 fp64 the resident launch is 0.85-0.91 x the three launches it replaces (profiles/r06_v1_resident_pcg_*:
  // the grid barriers cost 2-3 us each, but the operator phase is issue / latency bound at the same per-SIMD rate with two waves per SIMD
  // where the stand-alone kernel has three, and the owners' registers spill); kept as a tested form behind gr_bal_tuning.pcg_resident = 1
  bool resident_wanted() { return tune.pcg_resident > 0 && !rp_disabled && resident_fits(); }
  // algorithmic bytes of one resident launch: what it reads once (the lane's streams, packs, points, the finalisation's vectors and
  // inverses) and leaves (dx, the stepped vertices, their backups, the packs) + per inner iteration the exchange arrays touched once
  // each way
  double resident_fixed_bytes() const { return No * (2.0 * w() + 12.0) + (24.0 * Nc + 3.0 * Np) * w() + 5.0 * n * w() + (81.0 * Nc + 9.0 * Np) * w() + 5.0 * n * w() + 24.0 * Nc * w(); }
  double resident_iter_bytes() const { return 2.0 * (6.0 * Np + 18.0 * Nc) * w() + 2.0 * 3.0 * No * w() + 2.0 * 9.0 * nseg * w() + (24.0 * Nc + 81.0 * Nc + 9.0 * Np + 2.0 * n) * w(); }
  void launch_resident(PcgState st, const LmDev *lm, double mu, bool use_identity, bool identity_precond, int max_iter, double tol, double rej) {
    RpParams<T> P{};
    P.No = (int)No; P.Nc = (int)Nc; P.Np = (int)Np;
    P.loss_kind = loss_kind; P.loss_delta = loss_delta;
    P.cam_fixed = cam_fixed_p(); P.pt_fixed = pt_fixed_p();
    P.rec6 = rp_rec6.p; P.crec = rp_crec.p; P.g4 = rp_g4.p; P.op_partial = op_partial.p;
    P.max_iter = max_iter; P.use_identity = use_identity ? 1 : 0; P.identity_precond = identity_precond ? 1 : 0;
    P.tol = tol; P.rej = rej; P.mu = mu;
    P.st = st; P.bar = rp_bar.p; P.fail = h_rp_fail; P.dx = v_dx.p; P.lm = lm;
    P.var = env_int("GR_RP_VAR", 0);
    if (env_int("GR_RP_DEBUG", 0)) { rp_dbg.alloc(64 * (size_t)rp_grid()); P.dbg = rp_dbg.p; }
    rho_blocks = rp_grid();
    rho_partial.alloc(rho_blocks);
    ApplyOnExit<T> ap;
    ap.cams = cams.p; ap.pts = pts.p; ap.cams_bak = cams_bak.p; ap.pts_bak = pts_bak.p; ap.bu = bu.p;
    ap.rho_partial = rho_partial.p; ap.pack = pack.p; ap.xp = (use_records && xp.n && xp_valid) ? xp.p : nullptr;
    ap.cam_weight = 1;
    P.ap = ap;
    Scope sc(this, "pcg_resident", resident_fixed_bytes(), 0.0, true);
    ++launch_count;
    hipEvent_t ea = ext_a, eb = ext_b;
    ext_a = ext_b = nullptr;
#define GR_RP_ARGS P, cam_cm.p, pt_cm.p, pos_cm.p, obs_cm.p, blk_seg.p, seg_slot.p, pt_ptr.p, cam_seg_ptr.p, pts.p, pack.p, scales.p, v_diag.p, MinvC.p, MinvP.p, v_r.p, v_z.p, v_zs.p
    if (ea) hipExtLaunchKernelGGL((k_pcg_resident<T, T>), dim3((unsigned)rp_grid()), dim3(RTPB), (unsigned)RpLds<T>::bytes, stream, ea, eb, 0, GR_RP_ARGS);
    else hipLaunchKernelGGL((k_pcg_resident<T, T>), dim3((unsigned)rp_grid()), dim3(RTPB), (unsigned)RpLds<T>::bytes, stream, GR_RP_ARGS);
#undef GR_RP_ARGS
    GR_HIP(hipGetLastError());
  }
  void launch_direction(PcgState st, int k, double tol, double rej, T *x, T *rec, const LmDev *lm, double mu, int max_iter) {
    Scope s3(this, "pcg_direction", 5.0 * n * sizeof(T), 3.0 * n, true);
    launch(k_pcg_direction<T>, lm_fused ? dir_grid() : grid_vec, (unsigned)n, x, v_xb.p, v_p.p, v_ps.p, v_z.p, scales.p, st, k, tol, rej, (unsigned)pose_dim, rec, lm, mu, apply_args(k, max_iter));
  }
  // k_linearize alone (LM loop, fused form): the finalisation follows in k_finalize_bj; its last workgroup clears the PCG loop state
  void linearize_deferred(const int *gate) {
    const double bytes = No * (2 * w() + 12.0) + (24.0 * Nc + 3.0 * Np) * w() + 8.0 * No * w() + 54.0 * nseg * w() + model_jac_bytes();
    Scope sc(this, "linearize", bytes, No * (250.0 + 48 + 117), true);
    lin_reset = ctl_cap > 0;
    launch_linearize_cam(false, g9.p, gate);
    lin_reset = false;
  }
  // finalisation of a pending linearisation on its own (loop exit, last iteration): k_linearize_finalize
  void flush_finalize(int spec_seq = 0) {
    Scope sc(this, "linearize_finalize", 8.0 * No * w() + 54.0 * nseg * w() + (90.0 * Nc + 15.0 * Np) * w(), 9.0 * No + 54.0 * nseg, true);
    launch(k_linearize_finalize<T>, cdiv(90 * (size_t)Nc, TPB) + cdiv(FIN_PL * (size_t)Np, TPB), (int)Nc, (int)Np, scale_system ? 1 : 0, 1, cam_seg_ptr.p, cam_partial.p, pt_ptr.p, g9.p, Hcc.p, bc.p, Hll.p, bl.p, scales.p, grid_lin, chi2_partial.p, dscalars.p,
                                                                                                       spec_seq ? rho_partial.p : nullptr, spec_seq ? rho_blocks : 0, spec_seq ? h_res : nullptr, h_seq, spec_seq, nullptr, cam_fixed_p(), pt_fixed_p(), IpcFused{}, 0ull);
    fin_pending = false; hcp_valid = false;
  }
  // head of one LM iteration; dec.seq != 0: with the accept decision about the trial step just linearised (then every kernel
  // of the head is gated on that decision and takes the damping from LmDev)
  template <bool IDENTITY> void enqueue_head(LmDecide dec, double mu, bool use_identity, int max_iter, double tol, double rej, int time_slot) {
    ensure_ctl(max_iter);
    ensure_point_records();
    // the first PCG iteration without its direction launch (operator and update in their lazy forms).  With point records too: the
    // record layout is a compile-time form of the operator, iteration 0 runs the plain form on zs / pts and the direction launch that
    // ends it fills the [X Y Z | s.p] sectors for iteration 1
    const bool first_lazy = max_iter > 0 && tune.lm_fused != 2 && !model; // (the lazy first operator is a form of the built-in kernel)
    if (first_lazy) v_zs.alloc(n);
    ts_slot = time_slot;
    flag_bank ^= 1; // the previous user of this bank is the solve before the last one: complete
    for (int k = 0; k < max_iter + 1; ++k) flags()[k] = 0;
    *h_iters() = 0;
    damping = mu; damping_identity = use_identity;
    T *rec = use_records ? xp.p : nullptr;
    PcgState st = pcg_state();
    st.x = v_dx.p; st.tol = tol; st.rej = rej; st.ts_op = first_lazy ? 1 : 0;
    // The point workgroups are persistent.  Few camera workgroups (Ladybug-1723: 62 of 1 024 slots): the whole grid is resident at
    // once — a point workgroup that has to wait for a camera workgroup's slot finishes a camera part + a full point share late
    // (30.9 vs 26.4 us).  Many (Final-13682: 489): they are short-lived, and leaving their slots to them alone starves the point
    // part (1 024 vs 840 us), so the point workgroups fill every slot and start as the camera workgroups retire.
    const int fbj_nbc = cdiv(Nc, 28), fbj_slots = num_cu * fbj_per_cu;
    const int fbj_nbp = std::max(1, std::min(cdiv(Np, TPB / FIN_PL), 4 * fbj_nbc < fbj_slots ? fbj_slots - fbj_nbc : fbj_slots));
    if (first_lazy) { fbj_part.alloc(3 * (size_t)(fbj_nbc + fbj_nbp)); st.part0 = fbj_part.p; st.n_part0 = fbj_nbc + fbj_nbp; }
    const bool resident = lm_resident && first_lazy;
    if (resident) { st.bar = rp_bar.p; st.bar_words = RP_BAR_WORDS; } // cleared by k_finalize_bj's first workgroup
    if (!pcg_state_clean) k_pcg_state_init<<<1, TPB, 0, stream>>>(st, ctl_cap);
    pcg_state_clean = false;
    const LmDev *lm = (dec.seq && !dec.report_only) ? lmdev.p : nullptr;
    const int ui = use_identity ? 1 : 0;
    {
      const int nbc = fbj_nbc, nbp = fbj_nbp;
      Scope sc(this, "finalize_bj", 8.0 * No * w() + 54.0 * nseg * w() + (2 * 90.0 * Nc + 36.0 * Np + 27.0 * Nc) * w(), 9.0 * No + 54.0 * nseg + 900.0 * Nc + 60.0 * Np, true);
      launch(k_finalize_bj<T>, nbc + nbp, (int)Nc, (int)Np, nbc, scale_system ? 1 : 0, cam_seg_ptr.p, cam_partial.p, pt_ptr.p, g9.p, Hcc.p, bu.p, Hll.p, scales.p, mu, ui, MinvC.p, MinvP.p, v_diag.p, st,
                                                      v_dx.p, v_r.p, v_z.p, first_lazy ? v_zs.p : nullptr, IDENTITY ? 1 : 0, 1, dec, cam_fixed_p(), pt_fixed_p());
    }
    fin_pending = false; hcp_valid = false;
    if (resident) { // every inner iteration and the trial step: one launch
      launch_resident(st, lm, mu, use_identity, IDENTITY, max_iter, tol, rej);
      return;
    }
    if (!first_lazy) launch_direction(st, -1, 0.0, 1e30, v_dx.p, rec, lm, mu, max_iter);
    if (max_iter > 0) {
      {
        Scope s1(this, "pcg_operator", operator_bytes(), No * 340.0, true);
        if (first_lazy) launch_operator_first(st, lm, mu); // lazy form on zs: no first direction launch
        else launch_operator(st, 0, rec, lm, mu);
      }
      {
        Scope s2(this, "pcg_update", 9.0 * n * sizeof(T) + (81.0 * Nc + 9.0 * Np) * sizeof(T) + 9.0 * nseg * sizeof(T) + 3.0 * No * sizeof(T), 16.0 * n + 2.0 * (81.0 * Nc + 9.0 * Np), true);
        launch_update<1, IDENTITY>(update_blocks(), v_dx.p, nullptr, 1, ui, st, 0, -1, -1, lm, first_lazy);
      }
      launch_direction(st, 0, tol, rej, v_dx.p, rec, lm, mu, max_iter);
    }
  }
  // operator of iteration 0 on the UN-normalised direction s .* z' that k_finalize_bj left in zs (A is linear; the update kernel of
  // iteration 0 applies sigma = 1 / |r|): the plain kernel, no decision prologue, no first direction launch
  void launch_operator_first(PcgState st, const LmDev *lm, double mu) {
#define GR_OP_FIRST(JT_) launch(k_pcg_operator<T, 0, JT_>, grid_op, (int)No, (int)Nc, o_ntiles(), o_cam(), o_pt(), g3_pos(), o_obs(), blk_seg.p, seg_slot.p, pts.p, pack.p, loss_kind, loss_delta, v_zs.p, g3.p, op_partial.p, mu, st, 0, nullptr, lm, ShardPush{})
    if constexpr (sizeof(T) == 8) {
      if (jac32) { GR_OP_FIRST(float); return; }
    }
    GR_OP_FIRST(T);
#undef GR_OP_FIRST
  }
  // iterations 1 ... of the solve whose head is in the stream; returns once the host has seen the loop leave
  template <bool IDENTITY> void continue_pcg(int max_iter, double tol, double rej) {
    T *rec = use_records ? xp.p : nullptr;
    PcgState st = pcg_state();
    st.x = v_dx.p; st.tol = tol; st.rej = rej;
    const int ui = damping_identity ? 1 : 0;
    auto enqueue = [&](int k) {
      { Scope s1(this, "pcg_operator", operator_bytes(), No * 340.0, true); launch_operator(st, k, rec, nullptr, damping); }
      {
        Scope s2(this, "pcg_update", 9.0 * n * sizeof(T) + (81.0 * Nc + 9.0 * Np) * sizeof(T) + 9.0 * nseg * sizeof(T) + 3.0 * No * sizeof(T), 16.0 * n + 2.0 * (81.0 * Nc + 9.0 * Np), true);
        launch_update<1, IDENTITY>(update_blocks(), v_dx.p, nullptr, 1, ui, st, k);
      }
      launch_direction(st, k, tol, rej, v_dx.p, rec, nullptr, damping, max_iter);
    };
    note_noop({"pcg_operator", "pcg_update", "pcg_direction"}, run_pcg_iterations(max_iter, enqueue, /*pre_enqueued=*/1));
  }

  // ---- optimizer::levenberg_marquardt (optimizer/levenberg_marquardt.hpp:110-242) -----
  void lm(const gr_lm_options &opt, gr_lm_stats &st, double *chi2_trace, double *lambda_trace) override {
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    profiling = opt.profile != 0;
    if (profiling) { flush_prof(); prof.clear(); }
    std::memset(&st, 0, sizeof(st));
    struct LmScope { // the fused PCG start (solver_set_damping) is only armed inside this loop
      Engine *e;
      explicit LmScope(Engine *e_) : e(e_) { e->lm_x = e->v_dx.p; e->state_clean_cap = -1; e->update0_done = false; e->lm_fused = false; e->lm_resident = false; e->fin_pending = false; e->sf_active = false; e->sf_cont = false; }
      ~LmScope() { e->lm_x = nullptr; e->state_clean_cap = -1; e->update0_done = false; e->lm_fused = false; e->lm_resident = false; e->fin_pending = false; e->sf_active = false; e->sf_cont = false; e->profiling = false; }
    } lm_scope(this);
    struct EventPairs { // solve_seconds: events around every solve, read one iteration late, outside the decision -> launch path
      hipEvent_t ev[3][2];
      EventPairs() { for (auto &pr : ev) for (auto &e : pr) GR_HIP(hipEventCreate(&e)); }
      ~EventPairs() { for (auto &pr : ev) for (auto &e : pr) (void)hipEventDestroy(e); }
    } evp;
    T mu = (T)opt.initial_damping;
    T nu = 2;
    solver_update_structure(opt.solver);
    const bool pcg_solver = opt.solver == GR_SOLVER_PCG || opt.solver == GR_SOLVER_PCG_IDENTITY;
    // (user-traits problems too: their trial step goes through the user's Traits::update — gr_model_ops.step, a launch of its own behind the
    // loop-ending direction launch, gated on the loop's exit word like the trial linearisation — and their heads run the direction-kernel form)
    lm_fused = pcg_solver && !comm && pcg_mode() == 0 && opt.pcg_max_iter >= 1 && tune.lm_fused != 0;
    lm_resident = lm_fused && opt.iterations > 0 && resident_wanted();
    // (larger reduced systems keep the host-driven loop: measured on Ladybug-1723, 1 723 cameras, the merged finalisation and
    // back-substitution launches are no faster than their parts there — 1 716 vs 1 737 LM it/s — the shared S / b_S launch is what pays)
    sf_active = opt.solver == GR_SOLVER_PCG_SCHUR && schur_fused_ok(opt.pcg_max_iter) && schur_coop() && opt.iterations > 0;
    bool head_enqueued = false; // the head of the NEXT iteration is already in the stream (device-decided accept)
    T chi2v = 0;
    sf_cont = sf_active && tune.schur_fused != 1; // (1: the head stops on a rejected step and the host runs the rejection)
    if (sf_active) {
      // Schur solver, small reduced system: the first linearisation is finalised by the head of iteration 0 like every other one,
      // and that launch reports the initial chi2
      ensure_lm_buffers(); ensure_coop(opt.pcg_max_iter); ensure_scalars(opt.pcg_max_iter); // (pinned words are sized before the first head uses them)
      if (sf_cont) {
        Hcp1.alloc(27 * (size_t)No); sf_vsum.alloc(9 * (size_t)Np);
        const LmDev init{(double)mu, 0, 1}; // hsel = 1: the first linearisation writes buffer 0, the first head flips to it
        GR_HIP(hipMemcpyAsync(lmdev.p, &init, sizeof(init), hipMemcpyHostToDevice, stream));
        sf_nu = (double)nu;
      } else { // the head that stops on a rejected step reads mu / stop only — but never leave the record to what the allocation held
        const LmDev init{(double)mu, 0, 0};
        GR_HIP(hipMemcpyAsync(lmdev.p, &init, sizeof(init), hipMemcpyHostToDevice, stream));
      }
      campack();
      linearize_hcp_deferred();
      LmDecide dec;
      dec.seq = ++seq_counter; dec.report_only = 1;
      dec.chi2_partial = chi2_partial.p; dec.n_chi2 = grid_lin; dec.hres = h_res; dec.hres_seq = h_seq; dec.lm = lmdev.p; dec.dscal = dscalars.p;
      enqueue_schur_head(dec, (double)mu, opt.use_identity != 0, opt.pcg_max_iter, opt.pcg_tol, opt.pcg_rejection_ratio, 0);
      head_enqueued = true;
      wait_chi2(dec.seq);
      chi2v = (T)h_res[0];
    } else
    if (lm_fused && opt.iterations > 0) {
      // fused form: the first linearisation is finalised by the head of iteration 0 like every other one, and that launch
      // reports the initial chi2 (no separate finalize launch, no synchronising read before the loop)
      ensure_lm_buffers(); ensure_ctl(opt.pcg_max_iter);
      predicted_iters = opt.pcg_max_iter; last_active = opt.pcg_max_iter;
      campack();
      linearize_deferred(nullptr);
      fin_pending = true; pcg_state_clean = !model; // (the built-in k_linearize clears the loop state in its last workgroup)
      LmDecide dec;
      dec.seq = ++seq_counter; dec.report_only = 1;
      dec.chi2_partial = chi2_partial.p; dec.n_chi2 = grid_lin; dec.hres = h_res; dec.hres_seq = h_seq; dec.lm = lmdev.p; dec.dscal = dscalars.p;
      if (opt.solver == GR_SOLVER_PCG_IDENTITY) enqueue_head<true>(dec, (double)mu, opt.use_identity != 0, opt.pcg_max_iter, opt.pcg_tol, opt.pcg_rejection_ratio, 0);
      else enqueue_head<false>(dec, (double)mu, opt.use_identity != 0, opt.pcg_max_iter, opt.pcg_tol, opt.pcg_rejection_ratio, 0);
      head_enqueued = true;
      wait_chi2(dec.seq);
      chi2v = (T)h_res[0];
    } else {
      linearize();
      solver_update_values(opt.solver);
      chi2v = (T)read_scalar(0);
    }
    bool run = true;
    int accept_streak = 2; // consecutive accepted iterations (saturating): speculate only on a streak
    int num_bad = 0;       // levenberg_marquardt2 (:404-414): consecutive accepted iterations gaining < 0.1 %
    const bool spec_enabled = tune.lm_speculate != 0;
    const bool ahead_enabled = tune.lm_ahead != 0;
    const int64_t coll0 = coll_count, launch0 = launch_count + comm_launches, fused0 = fused_messages;
    if (chi2_trace) chi2_trace[0] = (double)chi2v;
    if (lambda_trace) lambda_trace[0] = (double)mu;
    // look-ahead predictor: nothing is known about the first solve of this call; from a fresh starting point the inner loop
    // usually runs long, so keep one iteration of look-ahead until it has ended once (a stale count from an earlier call
    // cost ~9 us of host round trip per inner iteration of the first solve, 90 us per call on the bench line)
    predicted_iters = opt.pcg_max_iter; last_active = opt.pcg_max_iter;
    auto collect_solve_time = [&](int which) {
      float ms = 0;
      (void)hipEventSynchronize(evp.ev[which][1]);
      (void)hipEventElapsedTime(&ms, evp.ev[which][0], evp.ev[which][1]);
      st.solve_seconds += ms * 1e-3;
    };
    int ev_waiting = -1; // pair whose events are recorded and not yet read
    if (!lm_fused && !sf_active) GR_HIP(hipStreamSynchronize(stream)); // fused forms: the loop's first kernels are already running
    st.setup_seconds = std::chrono::duration<double>(clk::now() - t0).count();
    auto tl = clk::now();

    int ahead_hits = 0, ahead_misses = 0, head_hits = 0;
    lm_iter_seconds.clear();
    // accept / reject bookkeeping of one trial step (levenberg_marquardt.hpp:184-233); false = leave the loop.
    // device: the decision k_finalize_bj took (and the kernels behind it follow): {new damping, accepted}
    auto decide = [&](int i, bool solve_ok, bool speculate, int it, const double *hs, const double *device) -> bool {
      st.pcg_iterations += it;
      T new_chi2 = (T)hs[0];
      if (!solve_ok) new_chi2 = std::numeric_limits<T>::max();
      T denom = solve_ok ? (T)hs[1] + (T)1.0e-3 : T(1);
      const T rho = (chi2v - new_chi2) / denom;
      const T initial_chi2 = chi2v;
      bool step_accepted = false;
      const bool accept = device ? device[1] != 0.0 : (solve_ok && std::isfinite(new_chi2) && rho > 0);
      if (accept) {
        step_accepted = true;
        if (device) mu = (T)device[0]; // the same arithmetic, taken on the device
        else {
          double alpha = 1.0 - std::pow(2.0 * rho - 1.0, 3);
          alpha = std::max(std::min(alpha, 2.0 / 3.0), 1.0 / 3.0);
          mu *= (T)alpha;
        }
        nu = 2;
        sf_nu = 2.0;
        if (!speculate) {
          if (lm_fused) { linearize_deferred(nullptr); fin_pending = true; pcg_state_clean = !model; }
          else linearize_impl(want_hcp, /*pack_valid=*/true);
        }
        solver_update_values(opt.solver);
        st.accepted++;
        accept_streak = std::min(accept_streak + 1, 2);
      } else if (device && sf_cont) {
        // the head went on from the kept point by itself (RejectCont): vertices taken back, damping raised, blocks in the other buffer
        accept_streak = 0;
        mu = (T)device[0];
        nu *= 2;
        new_chi2 = chi2v;
      } else {
        revert();
        if (speculate) { // restore H, b, scales of the kept point
          if (lm_fused) { linearize_deferred(nullptr); fin_pending = true; pcg_state_clean = !model; }
          else if (sf_active) linearize_hcp_deferred(); // finalised by the next head
          else linearize_impl(want_hcp, /*pack_valid=*/true);
        } else if (lm_fused) pcg_state_clean = false; // the old linearisation stands, the loop state is spent
        accept_streak = 0;
        // the reference recomputes error + chi2 here (:199-201); every consumer below
        // recomputes residuals from the reverted vertices, so nothing is stale.
        mu *= nu;
        nu *= 2;
        new_chi2 = chi2v;
      }
      chi2v = new_chi2;
      st.iterations_run++;
      lm_iter_seconds.push_back(std::chrono::duration<double>(clk::now() - tl).count());
      if (chi2_trace) chi2_trace[i + 1] = (double)chi2v;
      if (lambda_trace) lambda_trace[i + 1] = (double)mu;
      if (!std::isfinite(mu)) run = false;
      if (rho == 0) return false;
      if (opt.early_stop && step_accepted) {
        if (((initial_chi2 - chi2v) * 1.0e3) < initial_chi2) num_bad++;
        else num_bad = 0;
        if (num_bad >= 3) return false;
      }
      return true;
    };
    // one host-driven iteration: solve, trial step, hand-shake, decision
    auto host_iteration = [&](int i) -> bool {
      solver_set_damping(opt.solver, (double)mu, opt.use_identity != 0);
      const int pr = i % 3;
      // solve_seconds of the host-driven form comes from HIP events, and an event between two launches costs a ~6 us bubble
      // (k_finalize_bj -> first direction: 6.1 us, last direction -> k_linearize: 6.6 us in the kernel trace): they are only
      // recorded when the caller asked for timing (options->profile); the fused form stamps the device wall clock instead
      hipEvent_t ev_a = evp.ev[pr][0], ev_b = evp.ev[pr][1];
      const bool time_solve = profiling;
      if (time_solve) GR_HIP(hipEventRecord(ev_a, stream));
      if (ev_waiting >= 0) { collect_solve_time(ev_waiting); ev_waiting = -1; } // the previous iteration's pair
      const bool speculate = accept_streak >= 2 && spec_enabled;
      int seq = 0;
      // backup_parameters + apply_update + rho-denominator partials + the camera packs in one launch, then the
      // speculative linearisation; `gate` != nullptr: enqueued ahead of the PCG exit flag (run_pcg_iterations)
      auto enqueue_trial = [&](const int *gate) {
        if (time_solve) GR_HIP(hipEventRecord(ev_b, stream));
        rho_blocks = model ? model->step_blocks : cdiv(Nc, 28) + cdiv(3 * Np, TPB);
        rho_partial.alloc(rho_blocks);
        const bool clear_state = ctl_cap > 0 && pcg_solver;
        if (model) model_step(v_dx.p, /*with_backup=*/true, (double)mu, rho_partial.p, nullptr, gate, clear_state);
        else
        k_apply_update_rho<T><<<rho_blocks + (clear_state ? 1 : 0), TPB, 0, stream>>>((unsigned)n, (unsigned)pose_dim, cdiv(Nc, 28), cam_weight(), cams.p, pts.p, cams_bak.p, pts_bak.p, v_dx.p, scales.p, bu.p, (double)mu, rho_partial.p, pack.p, (use_records && xp.n && xp_valid) ? xp.p : nullptr,
                                                                                       nullptr, clear_state ? pcg_state() : PcgState{}, clear_state ? ctl_cap : 0, gate);
        seq = ++seq_counter;
        linearize_impl(want_hcp, /*pack_valid=*/true, seq, gate, trial_part);
      };
      // landmark shards: only the collective-free front of the trial (step + k_linearize) goes ahead of the exit flag
      auto enqueue_trial_front = [&](const int *gate) { trial_part = 1; enqueue_trial(gate); trial_part = 3; };
      const bool ahead = speculate && !profiling && ahead_enabled && pcg_solver;
      if (ahead) { if (comm) trial_hook = enqueue_trial_front; else trial_hook = enqueue_trial; }
      trial_done = false;
      const bool solve_ok = solver_solve_dev(opt.solver, opt.pcg_max_iter, opt.pcg_tol, opt.pcg_rejection_ratio, v_dx.p);
      trial_hook = nullptr;
      const bool trial_ahead = ahead && trial_done;
      if (ahead) (trial_ahead ? ahead_hits : ahead_misses)++;
      // Trial step.  After two accepted iterations the next one is expected to be accepted too, so the
      // trial chi2 is taken from a SPECULATIVE linearisation at the trial point (its chi2 is the same
      // sum): on acceptance the iteration is already linearised and the separate chi2 pass is saved;
      // on rejection the old linearisation is re-created from the reverted vertices (same bits: all
      // sums are fixed-order).  After a rejection the plain chi2 pass is used.
      if (speculate) {
        if (!trial_ahead) enqueue_trial(nullptr);
        else if (comm) linearize_impl(want_hcp, /*pack_valid=*/true, seq, nullptr, 2); // the loop has left: finalisation + the sums over ranks
        if (ctl_cap > 0 && pcg_solver) state_clean_cap = ctl_cap;
        if (!(use_records && xp.n && xp_valid)) xp_valid = false;
      } else {
        if (time_solve) GR_HIP(hipEventRecord(ev_b, stream));
        apply_update_dev(v_dx.p, /*with_backup=*/true);
        // trial chi2 + compute_rho denominator (:20-47) in one kernel; its last block mirrors the two
        // sums into pinned host memory, so the host polls one word instead of memcpy + stream sync
        seq = chi2_async(nullptr, v_dx.p, (double)mu);
      }
      wait_chi2(seq);
      const int it = *h_iters(); // every PCG variant mirrors its iteration count into pinned memory
      const double hs[2] = {h_res[0], h_res[1]};
      if (time_solve) ev_waiting = pr;
      return decide(i, solve_ok, speculate, it, hs, nullptr);
    };
    // the same iteration in the fused form (lm_fused); solve time from device wall-clock stamps (an event between two launches
    // costs a ~6 us bubble)
    auto fused_iteration = [&](int i) -> bool {
      const int mi = opt.pcg_max_iter;
      const bool ui = opt.use_identity != 0, ident = opt.solver == GR_SOLVER_PCG_IDENTITY;
      const double tol = opt.pcg_tol, rej = opt.pcg_rejection_ratio;
      last_solver = opt.solver;
      const int pr = i % 4;
      if (!head_enqueued) {
        if (ident) enqueue_head<true>(LmDecide{}, (double)mu, ui, mi, tol, rej, pr);
        else enqueue_head<false>(LmDecide{}, (double)mu, ui, mi, tol, rej, pr);
      } else damping = (double)mu; // the head took it from LmDev; the iterations enqueued from here on get it as an argument
      head_enqueued = false;
      ts_slot = pr;
      const bool speculate = accept_streak >= 2 && spec_enabled;
      auto enqueue_lin = [&](const int *gate) {
        if (model) { // backup + Traits::update + the rho-denominator partials, then the trial linearisation: both wait for the loop's exit word
          rho_blocks = model->step_blocks;
          rho_partial.alloc(rho_blocks);
          model_step(v_dx.p, /*with_backup=*/true, (double)mu, rho_partial.p, nullptr, gate, /*clear_state=*/true);
        }
        linearize_deferred(gate);
      };
      const bool ahead = speculate && ahead_enabled;
      if (ahead) trial_hook = enqueue_lin;
      trial_done = false;
      volatile int *const hit = h_iters(); // (the next head moves on to the other flag bank)
      if (lm_resident) { // the whole solve and the step are one launch, already in the stream: the trial linearisation simply follows it
        if (ahead) { enqueue_lin(nullptr); trial_done = true; }
      } else if (ident) continue_pcg<true>(mi, tol, rej); else continue_pcg<false>(mi, tol, rej);
      trial_hook = nullptr;
      if (ahead) (trial_done ? ahead_hits : ahead_misses)++;
      int it = lm_resident ? 0 : *hit; // (resident: known once the launch has run — read after wait_chi2)
      if (!(use_records && xp.n && xp_valid)) xp_valid = false; // the step has been applied by the last direction launch
      int seq = 0;
      if (speculate) {
        if (!(ahead && trial_done)) enqueue_lin(nullptr);
        fin_pending = true; pcg_state_clean = true; // (built-in model: k_linearize's last workgroup; user traits: the step launch)
        seq = ++seq_counter;
        if (i + 1 < opt.iterations) {
          LmDecide dec;
          dec.seq = seq; dec.chi2_cur = (double)chi2v; dec.mu_cur = (double)mu;
          dec.chi2_partial = chi2_partial.p; dec.n_chi2 = grid_lin; dec.rho_partial = rho_partial.p; dec.n_rho = rho_blocks;
          dec.hres = h_res; dec.hres_seq = h_seq; dec.lm = lmdev.p; dec.dscal = dscalars.p;
          if (ident) enqueue_head<true>(dec, (double)mu, ui, mi, tol, rej, (i + 1) % 4);
          else enqueue_head<false>(dec, (double)mu, ui, mi, tol, rej, (i + 1) % 4);
          head_enqueued = true;
          ++head_hits;
        } else flush_finalize(seq); // last iteration: only the decision is needed
      } else {
        if (model) apply_update_dev(v_dx.p, /*with_backup=*/true);
        seq = chi2_async(nullptr, v_dx.p, (double)mu); // (built-in model: the step itself was applied by the last direction launch)
      }
      wait_chi2(seq);
      if (lm_resident) {
        if (*h_rp_fail) { // not every workgroup was resident: this handle goes back to the multi-launch form; this call cannot be continued
          *h_rp_fail = 0; rp_disabled = true;
          GR_HIP(hipMemsetAsync(rp_bar.p, 0, sizeof(unsigned) * RP_BAR_WORDS, stream));
          throw HipError("resident PCG: a grid barrier timed out (workgroups not co-resident); the handle now uses the multi-launch form (gr_bal_tuning.pcg_resident = 0)");
        }
        it = *hit;
        if (profiling) { auto pf = prof.find("pcg_resident"); if (pf != prof.end()) pf->second.bytes += it * resident_iter_bytes(); }
      }
      st.solve_seconds += (double)(h_ts[2 * pr + 1] - h_ts[2 * pr]) / wall_clock_hz;
      const double hs[2] = {h_res[0], h_res[1]};
      const double dev[2] = {h_res[2], h_res[3]};
      const bool go = decide(i, true, speculate, it, hs, head_enqueued ? dev : nullptr);
      if (head_enqueued && dev[1] == 0.0) head_enqueued = false; // not accepted: the head returned at once
      return go;
    };

    // PCGSchurSolver on a small reduced system (kernels_sf.hpp): every trial step is judged by the finalisation launch of the NEXT
    // iteration's head, which the host enqueues (with the rest of that head) right behind the trial linearisation
    auto schur_fused_iteration = [&](int i) -> bool {
      const int mi = opt.pcg_max_iter;
      const bool ui = opt.use_identity != 0;
      last_solver = opt.solver;
      const int pr = i % 4;
      if (!head_enqueued) enqueue_schur_head(LmDecide{}, (double)mu, ui, mi, opt.pcg_tol, opt.pcg_rejection_ratio, pr);
      else damping = (double)mu;
      head_enqueued = false;
      const volatile int *iters_word = h_iters(); // this head's bank (the next head moves on to the other one)
      linearize_hcp_deferred(); // at the trial point the head's last launch has stepped to
      const int seq = ++seq_counter;
      if (i + 1 < opt.iterations) {
        LmDecide dec;
        dec.seq = seq; dec.chi2_cur = (double)chi2v; dec.mu_cur = (double)mu;
        dec.chi2_partial = chi2_partial.p; dec.n_chi2 = grid_lin; dec.rho_partial = rho_partial.p; dec.n_rho = rho_blocks;
        dec.hres = h_res; dec.hres_seq = h_seq; dec.lm = lmdev.p; dec.dscal = dscalars.p;
        enqueue_schur_head(dec, (double)mu, ui, mi, opt.pcg_tol, opt.pcg_rejection_ratio, (i + 1) % 4);
        head_enqueued = true;
      } else { // last iteration: only the decision is needed
        Scope sc(this, "linearize_finalize", 8.0 * No * w() + 54.0 * nseg * w() + (90.0 * Nc + 15.0 * Np) * w(), 9.0 * No + 54.0 * nseg, true);
        launch(k_linearize_finalize<T>, cdiv(90 * (size_t)Nc, TPB) + cdiv(FIN_PL * (size_t)Np, TPB), (int)Nc, (int)Np, scale_system ? 1 : 0, 1, cam_seg_ptr.p, cam_partial.p, pt_ptr.p, g9.p, Hcc.p, bc.p, Hll.p, bl.p, scales.p, grid_lin, chi2_partial.p, dscalars.p,
               rho_partial.p, rho_blocks, h_res, h_seq, seq, nullptr, cam_fixed_p(), pt_fixed_p(), IpcFused{}, 0ull);
      }
      wait_chi2(seq);
      if (*h_coop_fail) {
        // A grid barrier of the cooperative launch timed out: its workgroups were not all resident (CUs held by another stream or
        // process).  Not sticky: the flag is cleared, THIS handle keeps the per-iteration launches from now on, the stale
        // step the back-substitution launch applied is taken back and the iteration runs again on the host-driven loop.  (A head already
        // enqueued behind it times out once more: the synchronisation below waits for it.)
        GR_HIP(hipStreamSynchronize(stream));
        *h_coop_fail = 0;
        coop_fail_dev.zero(stream);
        sf_disabled = true; sf_active = false; sf_cont = false; head_enqueued = false;
        if (tune.verbose) std::fprintf(stderr, "[graphite-mi355x] cooperative PCG on S: a grid barrier timed out; this problem uses the host-driven loop from here on\n");
        revert();
        hcp_valid = false;
        linearize_impl(want_hcp);
        solver_update_values(opt.solver);
        GR_HIP(hipStreamSynchronize(stream));
        return host_iteration(i);
      }
      const int it = *iters_word;
      if (schur_coop()) st.solve_seconds += (double)(h_ts[2 * pr + 1] - h_ts[2 * pr]) / wall_clock_hz; // device stamps around the cooperative PCG
      const double hs[2] = {h_res[0], h_res[1]};
      const double dev[2] = {h_res[2], h_res[3]};
      const bool go = decide(i, true, /*speculate=*/true, it, hs, head_enqueued ? dev : nullptr);
      if (head_enqueued && dev[1] == 0.0 && !sf_cont) head_enqueued = false; // not accepted: the head returned at once (RejectCont: it went on)
      sf_nu = (double)nu;
      return go;
    };

    int i = 0;
    while (i < opt.iterations && run) {
      const bool go = sf_active ? schur_fused_iteration(i) : lm_fused ? fused_iteration(i) : host_iteration(i);
      ++i;
      if (!go) break;
      if (opt.stop_flag) { // levenberg_marquardt.hpp:233-238: polled once per iteration
        double stop = *opt.stop_flag ? 1.0 : 0.0;
        // landmark shards: a host flag is per process; the ranks leave together or not at all (one scalar all-reduce per
        // iteration, only when a flag was given), otherwise one rank would enqueue collectives its peers never join
        if (comm) allreduce_host(&stop, 1);
        if (stop != 0.0) break;
      }
    }
    if (head_enqueued) {
      // the loop ends here but the head of the next iteration is already running: if its PCG loop ended inside the head,
      // its last direction launch has applied a trial step nobody will judge — take it back
      GR_HIP(hipStreamSynchronize(stream));
      // (Schur form: an accepted head always ends with its trial step; matrix-free form on user traits: the head applies none)
      if (sf_active || (!model && (flags()[0] == 2 || opt.pcg_max_iter == 1))) revert();
      head_enqueued = false;
    }
    if (fin_pending) flush_finalize();
    if (sf_cont) hcp_valid = false; // (the current point's blocks may sit in the second buffer: whoever needs them next re-linearises)
    GR_HIP(hipStreamSynchronize(stream));
    st.loop_seconds = std::chrono::duration<double>(clk::now() - tl).count();
    if (lm_resident && rp_dbg.n && env_int("GR_RP_DEBUG", 0)) { // phase stamps of the LAST resident launch, over the workgroups (100 MHz clock)
      GR_HIP(hipStreamSynchronize(stream));
      const std::vector<long long> h = rp_dbg.download(stream);
      const int G = rp_grid();
      long long t0 = h[0];
      for (int b = 0; b < G; ++b) t0 = std::min(t0, h[(size_t)b * 64]);
      const int ns = (int)h[63];
      std::fprintf(stderr, "[rp-debug] %d stamps; per stamp: min / mean / max over %d workgroups, us since the first workgroup started\n", ns, G);
      for (int i = 0; i < ns && i < 63; ++i) {
        long long lo = h[i], hi = h[i]; double sum = 0;
        for (int b = 0; b < G; ++b) { const long long v = h[(size_t)b * 64 + i]; lo = std::min(lo, v); hi = std::max(hi, v); sum += (double)v; }
        std::fprintf(stderr, "[rp-debug]   %2d  %7.2f %7.2f %7.2f\n", i, (lo - t0) * 0.01, (sum / G - t0) * 0.01, (hi - t0) * 0.01);
      }
    }
    if (ev_waiting >= 0) collect_solve_time(ev_waiting);
    st.ok = run ? 1 : 0;
    st.final_chi2 = (double)chi2v;
    st.collectives = coll_count - coll0;
    st.kernel_launches = launch_count + comm_launches - launch0;
    st.fused_messages = fused_messages - fused0;
    check_comm("levenberg_marquardt");
    if (tune.verbose) std::fprintf(stderr, "[graphite-mi355x] LM: %s; trial linearisation enqueued ahead of the PCG exit flag in %d iterations, not ahead in %d; next head enqueued behind the trial step in %d\n",
                                   sf_active ? "Schur solver, device-decided head (finalisation, S, cooperative PCG, back-substitution, trial step)" : lm_fused ? "fused head / trial step" : "host loop", ahead_hits, ahead_misses, head_hits);
    if (profiling) flush_prof();
  }
};

} // namespace gr

// =============================================================================
// C ABI
// =============================================================================
using namespace gr;

struct gr_bal_problem {
  std::unique_ptr<EngineBase> e;
  int dtype;
};

template <typename F> static gr_status guarded(gr_bal_problem *p, F &&f) {
  if (!p || !p->e) { g_last_error = "null problem handle"; return GR_ERR_INVALID; }
  try {
    (void)hipSetDevice(p->e->device);
    f();
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const CommError &ex) { g_last_error = ex.what(); return GR_ERR_COMM; }
  catch (const std::domain_error &ex) { g_last_error = ex.what(); return GR_ERR_DUPLICATE_EDGE; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_INVALID; }
}

// standalone entry to the MFMA Cholesky (chol.hpp): A is n x n row-major with leading dimension lda,
// symmetric positive definite, only its lower triangle is read
template <typename T> static void dense_cholesky_solve_impl(int64_t n, const void *A, int64_t lda, const void *b, void *x, hipStream_t stream, double *seconds) {
  DenseChol<T> ch;
  ch.set_structure((int)n, {}, stream);
  GR_HIP(hipMemsetAsync(ch.A.p, 0, ch.A.n * sizeof(T), stream));
  ch.clear();
  GR_HIP(hipMemcpy2DAsync(ch.A.p, (size_t)ch.ld() * sizeof(T), A, (size_t)lda * sizeof(T), (size_t)n * sizeof(T), (size_t)n, hipMemcpyDefault, stream));
  DevBuf<T> db; db.alloc(n);
  GR_HIP(hipMemcpyAsync(db.p, b, (size_t)n * sizeof(T), hipMemcpyDefault, stream));
  hipEvent_t e0, e1;
  GR_HIP(hipEventCreate(&e0)); GR_HIP(hipEventCreate(&e1));
  GR_HIP(hipEventRecord(e0, stream));
  ch.factor();
  GR_HIP(hipEventRecord(e1, stream));
  ch.solve(db.p, db.p);
  const bool ok = ch.ok();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (seconds) *seconds = ms * 1e-3;
  GR_HIP(hipMemcpyAsync(x, db.p, (size_t)n * sizeof(T), hipMemcpyDefault, stream));
  GR_HIP(hipStreamSynchronize(stream));
  if (!ok) throw std::range_error("matrix is not positive definite");
}
// the engine's camera model on caller-supplied (camera, point, observation) triples: what the hand-written kernels compute
// per observation (bal_device.hpp), one thread per triple
template <typename T>
__global__ void k_model_evaluate(int n, const T *__restrict__ cams, const T *__restrict__ pts, const T *__restrict__ obs,
                                 T *__restrict__ res, T *__restrict__ Jc, T *__restrict__ Jp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T cam[9], pk[PACK], e0, e1, jc[18], jp[6];
#pragma unroll
  for (int k = 0; k < 9; ++k) cam[k] = cams[9 * (size_t)i + k];
  make_campack(cam, pk);
  bal_linearize<T>(pk, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], obs[2 * (size_t)i], obs[2 * (size_t)i + 1], e0, e1, jc, jp);
  if (res) { res[2 * (size_t)i] = e0; res[2 * (size_t)i + 1] = e1; }
  if (Jc) {
#pragma unroll
    for (int k = 0; k < 18; ++k) Jc[18 * (size_t)i + k] = jc[k];
  }
  if (Jp) {
#pragma unroll
    for (int k = 0; k < 6; ++k) Jp[6 * (size_t)i + k] = jp[k];
  }
}
template <typename T> static void model_evaluate_impl(int64_t n, const void *cams, const void *pts, const void *obs, void *res, void *Jc, void *Jp, hipStream_t stream) {
  DevBuf<T> dc, dp, dob, dr, djc, djp;
  dc.alloc(9 * (size_t)n); dp.alloc(3 * (size_t)n); dob.alloc(2 * (size_t)n); dr.alloc(2 * (size_t)n); djc.alloc(18 * (size_t)n); djp.alloc(6 * (size_t)n);
  GR_HIP(hipMemcpyAsync(dc.p, cams, 9 * (size_t)n * sizeof(T), hipMemcpyDefault, stream));
  GR_HIP(hipMemcpyAsync(dp.p, pts, 3 * (size_t)n * sizeof(T), hipMemcpyDefault, stream));
  GR_HIP(hipMemcpyAsync(dob.p, obs, 2 * (size_t)n * sizeof(T), hipMemcpyDefault, stream));
  k_model_evaluate<T><<<cdiv(n, 64), 64, 0, stream>>>((int)n, dc.p, dp.p, dob.p, dr.p, djc.p, djp.p);
  if (res) GR_HIP(hipMemcpyAsync(res, dr.p, 2 * (size_t)n * sizeof(T), hipMemcpyDefault, stream));
  if (Jc) GR_HIP(hipMemcpyAsync(Jc, djc.p, 18 * (size_t)n * sizeof(T), hipMemcpyDefault, stream));
  if (Jp) GR_HIP(hipMemcpyAsync(Jp, djp.p, 6 * (size_t)n * sizeof(T), hipMemcpyDefault, stream));
  GR_HIP(hipStreamSynchronize(stream));
}
// ---- gr_spchol: the nested-dissection tile Cholesky on any block-sparse SPD matrix (include/graphite_mi355x.h) -----------------------
struct gr_spchol {
  gr_dtype dtype = GR_F64;
  int device = 0;
  int64_t nnzb = 0;
  SparseChol<double> d;
  SparseChol<float> f;
  DevBuf<int> rowi, coli;
  DevBuf<double> stage_d; // host-pointer callers: staged copies of blocks | b (x goes back from there)
  DevBuf<float> stage_f;
  int64_t n = 0, bb = 0;  // scalar columns, scalars per block
};
// true: p can be dereferenced by a kernel (device or managed memory, or registered / pinned host memory the device maps)
static bool spchol_device_visible(const void *p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}
template <typename T> static void spchol_create_impl(gr_spchol *h, SparseChol<T> &sc, int64_t nodes, int bs, int64_t nnzb, const int64_t *brow, const int64_t *bcol, hipStream_t stream) {
  std::vector<int> r((size_t)nnzb), c((size_t)nnzb);
  std::vector<char> has_diag((size_t)nodes, 0);
  for (int64_t q = 0; q < nnzb; ++q) {
    if (brow[q] < 0 || bcol[q] >= nodes || brow[q] > bcol[q]) throw std::invalid_argument("gr_spchol_create: block (" + std::to_string(brow[q]) + ", " + std::to_string(bcol[q]) + ") is not an upper block of the matrix");
    r[(size_t)q] = (int)brow[q]; c[(size_t)q] = (int)bcol[q];
    if (brow[q] == bcol[q]) has_diag[(size_t)brow[q]] = 1;
  }
  for (int64_t i = 0; i < nodes; ++i) if (!has_diag[(size_t)i]) throw std::invalid_argument("gr_spchol_create: node " + std::to_string(i) + " has no diagonal block");
  if (!sc.set_structure((int)nodes, r, c, stream, bs)) throw std::range_error("gr_spchol_create: the node graph does not dissect (one supernode): use gr_dense_cholesky_solve");
  sc.allocate();
  h->rowi.alloc((size_t)nnzb); h->coli.alloc((size_t)nnzb);
  GR_HIP(hipMemcpyAsync(h->rowi.p, r.data(), (size_t)nnzb * sizeof(int), hipMemcpyHostToDevice, stream));
  GR_HIP(hipMemcpyAsync(h->coli.p, c.data(), (size_t)nnzb * sizeof(int), hipMemcpyHostToDevice, stream));
  GR_HIP(hipStreamSynchronize(stream));
}

extern "C" {

const char *gr_version(void) { return "graphite-mi355x 0.2 (gfx950)"; } // 0.2: sizeof(gr_bal_tuning) = 100, sizeof(gr_lm_stats) = 80 (both grew in round 4), gr_model_ops
const char *gr_last_error_string(void) { return g_last_error.c_str(); }
int gr_device_count(void) {
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess) return 0;
  return nd;
}

__global__ void k_warm_up() {}
gr_status gr_warm_up(int device) {
  try {
    GR_HIP(hipSetDevice(device));
    k_warm_up<<<1, 1>>>();
    // ... and the runtime's own lazily built machinery (fill / copy blit kernels, staging buffers of pageable copies in both
    // directions, the pinned-memory path): a 4 MB round trip through each, once per process (20 ms on first use, measured
    // inside gr_bal_create before this)
    {
      static bool done = false;
      if (!done) {
        done = true;
        const size_t nb = (size_t)4 << 20;
        std::vector<char> h(nb, 1);
        DevBuf<char> d;
        d.alloc(nb);
        GR_HIP(hipMemsetAsync(d.p, 0, nb, nullptr));
        GR_HIP(hipMemcpyAsync(d.p, h.data(), nb, hipMemcpyHostToDevice, nullptr));
        GR_HIP(hipMemcpyAsync(h.data(), d.p, nb, hipMemcpyDeviceToHost, nullptr));
        void *pin = nullptr;
        GR_HIP(hipHostMalloc(&pin, 1 << 16, hipHostMallocCoherent | hipHostMallocMapped));
        GR_HIP(hipStreamSynchronize(nullptr));
        (void)hipHostFree(pin);
        UploadRing::get().ensure(); // this thread's pinned staging ring (common.hpp)
      }
    }
    GR_HIP(hipDeviceSynchronize());
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
}

static gr_status create_impl(gr_bal_problem **out, gr_dtype dtype, int64_t nc, int64_t np, int64_t no,
                             const void *cameras, const void *points, const void *observations,
                             const int32_t *cam_idx, const int32_t *pt_idx, int device, void *stream, bool shard, const gr_model_ops *model = nullptr) {
  if (model && (model->pose_dim < 1 || model->pose_dim > 9 || model->landmark_dim < 1 || model->landmark_dim > 3 || model->error_dim < 1 || model->error_dim > 2 ||
                !model->linearize || !model->chi2 || !model->step || !model->backup || !model->revert || (!model->store_jacobians && !model->op) || model->step_blocks < 1 ||
                (model->storage_dtype != GR_F32 && model->storage_dtype != GR_F64) || (model->storage_dtype == GR_F64 && dtype == GR_F32))) {
    g_last_error = "gr_bal_create_model: the launcher table does not describe a (<= 9, <= 3) -> <= 2 model";
    return GR_ERR_INVALID;
  }
  if (!out || nc <= 0 || np <= 0 || no <= 0 || (!model && (!cameras || !points || !observations)) || !cam_idx || !pt_idx ||
      no >= (int64_t)std::numeric_limits<int>::max() / 27) {
    g_last_error = "gr_bal_create: bad argument";
    return GR_ERR_INVALID;
  }
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= device || device < 0) {
    g_last_error = "no HIP device: the MI355X path has no CPU fallback";
    return GR_ERR_NO_DEVICE;
  }
  try {
    GR_HIP(hipSetDevice(device));
    auto *p = new gr_bal_problem();
    p->dtype = dtype;
    if (dtype == GR_F32) p->e.reset(new Engine<float>(nc, np, no, cameras, points, observations, cam_idx, pt_idx, device, (hipStream_t)stream, shard, model));
    else if (dtype == GR_F64) p->e.reset(new Engine<double>(nc, np, no, cameras, points, observations, cam_idx, pt_idx, device, (hipStream_t)stream, shard, model));
    else { delete p; g_last_error = "bad dtype"; return GR_ERR_INVALID; }
    *out = p;
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const std::domain_error &ex) { g_last_error = ex.what(); return GR_ERR_DUPLICATE_EDGE; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_INVALID; }
}
gr_status gr_bal_create_model(gr_bal_problem **out, gr_dtype dtype, int64_t nc, int64_t np, int64_t no, const int32_t *pose_idx, const int32_t *landmark_idx,
                              const gr_model_ops *ops, int device, void *stream) {
  if (!ops) { g_last_error = "gr_bal_create_model: ops is NULL"; return GR_ERR_INVALID; }
  return create_impl(out, dtype, nc, np, no, nullptr, nullptr, nullptr, pose_idx, landmark_idx, device, stream, false, ops);
}
gr_status gr_bal_model_orders(gr_bal_problem *p, int32_t *obs_order, int32_t *landmark_order) {
  return guarded(p, [&] { p->e->orders(obs_order, landmark_order); });
}
gr_status gr_bal_create(gr_bal_problem **out, gr_dtype dtype, int64_t nc, int64_t np, int64_t no,
                        const void *cameras, const void *points, const void *observations,
                        const int32_t *cam_idx, const int32_t *pt_idx, int device, void *stream) {
  return create_impl(out, dtype, nc, np, no, cameras, points, observations, cam_idx, pt_idx, device, stream, false);
}
gr_status gr_bal_create_shard(gr_bal_problem **out, gr_dtype dtype, int64_t nc, int64_t np, int64_t no,
                              const void *cameras, const void *points, const void *observations,
                              const int32_t *cam_idx, const int32_t *pt_idx, int device, void *stream) {
  return create_impl(out, dtype, nc, np, no, cameras, points, observations, cam_idx, pt_idx, device, stream, true);
}
gr_status gr_bal_destroy(gr_bal_problem *p) {
  if (!p) return GR_ERR_INVALID;
  (void)hipSetDevice(p->e->device);
  delete p;
  return GR_OK;
}
void gr_bal_tuning_default(gr_bal_tuning *t) { if (t) tuning_default(*t); }
gr_status gr_bal_set_tuning(gr_bal_problem *p, const gr_bal_tuning *t) {
  if (!t) { g_last_error = "null tuning"; return GR_ERR_INVALID; }
  return guarded(p, [&] { p->e->tune = *t; p->e->tune.grid_mult = std::max(0, t->grid_mult); p->e->tune.vec_per_thread = std::max(1, t->vec_per_thread); p->e->apply_tuning(); });
}
gr_status gr_bal_get_tuning(gr_bal_problem *p, gr_bal_tuning *t) {
  if (!t) { g_last_error = "null tuning"; return GR_ERR_INVALID; }
  return guarded(p, [&] { *t = p->e->tune; });
}
gr_status gr_bal_set_loss(gr_bal_problem *p, gr_loss kind, double delta) { return guarded(p, [&] { p->e->set_loss(kind, delta); }); }
gr_status gr_bal_set_scale_system(gr_bal_problem *p, int enable) { return guarded(p, [&] { p->e->set_scale_system(enable != 0); }); }
gr_status gr_bal_set_fixed(gr_bal_problem *p, const unsigned char *cam_fixed, const unsigned char *pt_fixed) { return guarded(p, [&] { p->e->set_fixed(cam_fixed, pt_fixed); }); }
gr_status gr_bal_set_jacobian_precision(gr_bal_problem *p, gr_dtype dtype) { return guarded(p, [&] { p->e->set_jacobian_precision((int)dtype); }); }
gr_status gr_bal_set_params(gr_bal_problem *p, const void *c, const void *q) { return guarded(p, [&] { p->e->set_params(c, q); }); }
gr_status gr_bal_get_params(gr_bal_problem *p, void *c, void *q) { return guarded(p, [&] { p->e->get_params(c, q); }); }
gr_status gr_bal_linearize(gr_bal_problem *p) { return guarded(p, [&] { p->e->linearize(); }); }
gr_status gr_bal_chi2(gr_bal_problem *p, double *chi2) { return guarded(p, [&] { const double v = p->e->chi2(); if (chi2) *chi2 = v; }); }
gr_status gr_bal_backup_parameters(gr_bal_problem *p) { return guarded(p, [&] { p->e->backup(); }); }
gr_status gr_bal_revert_parameters(gr_bal_problem *p) { return guarded(p, [&] { p->e->revert(); }); }
gr_status gr_bal_apply_update(gr_bal_problem *p, const void *dx) { return guarded(p, [&] { p->e->apply_update(dx); }); }
gr_status gr_bal_solver_update_structure(gr_bal_problem *p, gr_solver s) { return guarded(p, [&] { p->e->solver_update_structure(s); }); }
gr_status gr_bal_solver_update_values(gr_bal_problem *p, gr_solver s) { return guarded(p, [&] { p->e->solver_update_values(s); }); }
gr_status gr_bal_solver_set_damping(gr_bal_problem *p, gr_solver s, double mu, int use_identity) { return guarded(p, [&] { p->e->solver_set_damping(s, mu, use_identity != 0); }); }
gr_status gr_bal_solver_solve(gr_bal_problem *p, gr_solver s, int max_iter, double tol, double rej, void *dx, int *iters) {
  bool ok = true;
  const gr_status st = guarded(p, [&] { ok = p->e->solver_solve(s, max_iter, tol, rej, dx, iters); });
  if (st == GR_OK && !ok) return GR_ERR_SOLVE_FAILED;
  return st;
}
gr_status gr_dense_cholesky_solve(gr_dtype dtype, int64_t n, const void *A, int64_t lda, const void *b, void *x, int device, void *stream, double *factor_seconds) {
  if (n <= 0 || lda < n || !A || !b || !x || n > (1 << 20)) { g_last_error = "gr_dense_cholesky_solve: bad argument"; return GR_ERR_INVALID; }
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= device || device < 0) { g_last_error = "no HIP device: the MI355X path has no CPU fallback"; return GR_ERR_NO_DEVICE; }
  try {
    GR_HIP(hipSetDevice(device));
    if (dtype == GR_F64) dense_cholesky_solve_impl<double>(n, A, lda, b, x, static_cast<hipStream_t>(stream), factor_seconds);
    else if (dtype == GR_F32) dense_cholesky_solve_impl<float>(n, A, lda, b, x, static_cast<hipStream_t>(stream), factor_seconds);
    else { g_last_error = "bad dtype"; return GR_ERR_INVALID; }
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const std::range_error &ex) { g_last_error = ex.what(); return GR_ERR_SOLVE_FAILED; }
  catch (const CommError &ex) { g_last_error = ex.what(); return GR_ERR_COMM; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_INVALID; }
}
gr_status gr_spchol_create(gr_spchol **out, gr_dtype dtype, int64_t num_nodes, int32_t block_size, int64_t num_blocks, const int64_t *block_row, const int64_t *block_col, int device, void *stream) {
  if (!out || num_nodes <= 0 || block_size <= 0 || block_size > 128 || num_blocks < num_nodes || !block_row || !block_col || num_nodes * (int64_t)block_size > (1 << 22) || (dtype != GR_F64 && dtype != GR_F32)) {
    // (the symbolic phase keeps a dense map of 128-column tiles: 2^22 columns = 1 GB of host memory for it)
    g_last_error = "gr_spchol_create: bad argument (at most 2^22 columns, block size 1 .. 128, every diagonal block listed)"; return GR_ERR_INVALID;
  }
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= device || device < 0) { g_last_error = "no HIP device: the MI355X path has no CPU fallback"; return GR_ERR_NO_DEVICE; }
  gr_spchol *h = nullptr;
  try {
    GR_HIP(hipSetDevice(device));
    h = new gr_spchol();
    h->dtype = dtype; h->device = device; h->nnzb = num_blocks; h->n = num_nodes * (int64_t)block_size; h->bb = (int64_t)block_size * block_size;
    if (dtype == GR_F64) spchol_create_impl<double>(h, h->d, num_nodes, block_size, num_blocks, block_row, block_col, static_cast<hipStream_t>(stream));
    else spchol_create_impl<float>(h, h->f, num_nodes, block_size, num_blocks, block_row, block_col, static_cast<hipStream_t>(stream));
    *out = h;
    return GR_OK;
  } catch (const HipError &ex) { delete h; g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const std::range_error &ex) { delete h; g_last_error = ex.what(); return GR_ERR_SOLVE_FAILED; }
  catch (const std::exception &ex) { delete h; g_last_error = ex.what(); return GR_ERR_INVALID; }
}
gr_status gr_spchol_factor_solve(gr_spchol *h, const void *blocks, const void *b, void *x) {
  if (!h || !blocks || !b || !x) { g_last_error = "gr_spchol_factor_solve: bad argument"; return GR_ERR_INVALID; }
  try {
    GR_HIP(hipSetDevice(h->device));
    auto run = [&](auto &sc, auto &stage) {
      using T = typename std::remove_pointer<decltype(stage.p)>::type;
      hipStream_t st = sc.stream;
      const size_t nv = (size_t)h->nnzb * (size_t)h->bb, n = (size_t)h->n;
      const T *bl = static_cast<const T *>(blocks), *rhs = static_cast<const T *>(b);
      T *sol = static_cast<T *>(x);
      const bool dev_bl = spchol_device_visible(blocks), dev_b = spchol_device_visible(b), dev_x = spchol_device_visible(x);
      if (!dev_bl || !dev_b || !dev_x) { // host-pointer callers (the Python binding, a CPU-side client): staged through the handle's own buffers
        stage.alloc(nv + 2 * n);
        if (!dev_bl) { GR_HIP(hipMemcpyAsync(stage.p, blocks, nv * sizeof(T), hipMemcpyHostToDevice, st)); bl = stage.p; }
        if (!dev_b) { GR_HIP(hipMemcpyAsync(stage.p + nv, b, n * sizeof(T), hipMemcpyHostToDevice, st)); rhs = stage.p + nv; }
        if (!dev_x) sol = stage.p + nv + n;
      }
      sc.load(h->nnzb, h->rowi.p, h->coli.p, bl);
      sc.factor_solve(rhs, sol); // (a level's forward substitution rides in its update launch, the backward one is one dependency-driven launch: §5)
      const bool ok = sc.ok();
      if (!dev_x) { GR_HIP(hipMemcpyAsync(x, sol, n * sizeof(T), hipMemcpyDeviceToHost, st)); GR_HIP(hipStreamSynchronize(st)); }
      return ok;
    };
    const bool ok = h->dtype == GR_F64 ? run(h->d, h->stage_d) : run(h->f, h->stage_f);
    if (!ok) { g_last_error = "gr_spchol_factor_solve: a pivot is not positive"; return GR_ERR_SOLVE_FAILED; }
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_INVALID; }
}
gr_status gr_spchol_info(const gr_spchol *h, gr_direct_solver_info *o) {
  if (!h || !o) { g_last_error = "gr_spchol_info: bad argument"; return GR_ERR_INVALID; }
  std::memset(o, 0, sizeof(*o));
  auto fill = [&](const auto &sc) {
    o->sparse = 1; o->tile_columns = sc.nt; o->levels = sc.nlevels; o->supernodes = sc.nsuper; o->padded_n = sc.npad;
    o->factor_tiles = sc.nz_tiles; o->factor_bytes = (int64_t)sc.bytes(); o->dense_bytes = (int64_t)sc.dense_bytes();
  };
  if (h->dtype == GR_F64) fill(h->d); else fill(h->f);
  return GR_OK;
}
void gr_spchol_destroy(gr_spchol *h) { delete h; }

gr_status gr_bal_model_evaluate(gr_dtype dtype, int64_t n, const void *cameras, const void *points, const void *observations,
                                void *residuals, void *Jc, void *Jp, int device, void *stream) {
  if (n <= 0 || n > (1 << 24) || !cameras || !points || !observations) { g_last_error = "gr_bal_model_evaluate: bad argument"; return GR_ERR_INVALID; }
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= device || device < 0) { g_last_error = "no HIP device: the MI355X path has no CPU fallback"; return GR_ERR_NO_DEVICE; }
  try {
    GR_HIP(hipSetDevice(device));
    if (dtype == GR_F64) model_evaluate_impl<double>(n, cameras, points, observations, residuals, Jc, Jp, static_cast<hipStream_t>(stream));
    else if (dtype == GR_F32) model_evaluate_impl<float>(n, cameras, points, observations, residuals, Jc, Jp, static_cast<hipStream_t>(stream));
    else { g_last_error = "bad dtype"; return GR_ERR_INVALID; }
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_INVALID; }
}
gr_status gr_bal_schur_update_values(gr_bal_problem *p) { return guarded(p, [&] { p->e->schur_update_values(); }); }
gr_status gr_bal_schur_matvec(gr_bal_problem *p, const void *x, void *y) { return guarded(p, [&] { p->e->schur_matvec(x, y); }); }
gr_status gr_bal_landmark_update(gr_bal_problem *p, const void *xp, void *xl) { return guarded(p, [&] { p->e->landmark_update(xp, xl); }); }
gr_status gr_bal_schur_structure(gr_bal_problem *p, int64_t *nnzb, int64_t *colptr, int64_t *rowidx) { return guarded(p, [&] { p->e->schur_structure(nnzb, colptr, rowidx); }); }
gr_status gr_bal_get(gr_bal_problem *p, gr_bal_array which, void *out, int64_t *count) { return guarded(p, [&] { p->e->get(which, out, count); }); }
gr_status gr_bal_hessian_structure(gr_bal_problem *p, int64_t *nblocks, int64_t *colptr, int64_t *rowidx, int64_t *offsets) {
  return guarded(p, [&] { p->e->hessian_structure(nblocks, colptr, rowidx, offsets); });
}
gr_status gr_bal_export_csc(gr_bal_problem *p, int which, int64_t *nnz, int64_t *indptr, int64_t *indices, void *values) {
  return guarded(p, [&] { p->e->export_csc(which, nnz, indptr, indices, values); });
}
gr_status gr_bal_levenberg_marquardt(gr_bal_problem *p, const gr_lm_options *opt, gr_lm_stats *stats, double *chi2_trace, double *lambda_trace) {
  if (!opt || !stats) { g_last_error = "null options/stats"; return GR_ERR_INVALID; }
  return guarded(p, [&] { p->e->lm(*opt, *stats, chi2_trace, lambda_trace); });
}
gr_status gr_bal_kernel_stats(gr_bal_problem *p, gr_kernel_stat *out, int cap, int *n) {
  return guarded(p, [&] { const int k = p->e->kernel_stats(out, cap); if (n) *n = k; });
}
gr_status gr_bal_lm_iteration_seconds(gr_bal_problem *p, double *seconds, int cap, int *n) {
  return guarded(p, [&] {
    const auto &v = p->e->lm_iter_seconds;
    if (n) *n = (int)v.size();
    for (int i = 0; seconds && i < cap && i < (int)v.size(); ++i) seconds[i] = v[i] - (i ? v[i - 1] : 0.0);
  });
}
gr_status gr_bal_direct_solver_info(gr_bal_problem *p, gr_direct_solver_info *info) {
  if (!info) { g_last_error = "gr_bal_direct_solver_info: bad argument"; return GR_ERR_INVALID; }
  return guarded(p, [&] { p->e->direct_solver_info(*info); });
}
// diagnostic (not part of the drop-in surface): mean device time in us of one hot kernel
double gr_bal_diag_time(gr_bal_problem *p, int which, int variant, int reps) {
  double us = -1;
  guarded(p, [&] { us = p->e->diag_time(which, variant, reps); });
  return us;
}
gr_status gr_comm_unique_id(void *unique_id_128) {
  try {
    RcclApi &api = RcclApi::get();
    if (!api.ok()) { g_last_error = "librccl.so.1 not found"; return GR_ERR_COMM; }
    api.check(api.GetUniqueId(unique_id_128), "ncclGetUniqueId");
    return GR_OK;
  } catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_COMM; }
}
gr_status gr_bal_comm_init(gr_bal_problem *p, const void *unique_id_128, int rank, int world_size) {
  if (!p || !p->e || !unique_id_128 || rank < 0 || rank >= world_size) { g_last_error = "gr_bal_comm_init: bad argument"; return GR_ERR_INVALID; }
  try {
    GR_HIP(hipSetDevice(p->e->device));
    p->e->set_comm(std::unique_ptr<Comm>(new RcclComm(unique_id_128, rank, world_size)));
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_COMM; }
}
gr_status gr_bal_comm_info(gr_bal_problem *p, gr_comm_info *info) {
  if (!info) { g_last_error = "gr_bal_comm_info: null info"; return GR_ERR_INVALID; }
  return guarded(p, [&] { p->e->comm_info(*info); });
}
gr_status gr_bal_comm_set_contributors(gr_bal_problem *p, const uint32_t *mask, int64_t count) {
  if (!p || !p->e || !mask) { g_last_error = "gr_bal_comm_set_contributors: bad argument"; return GR_ERR_INVALID; }
  try {
    GR_HIP(hipSetDevice(p->e->device));
    p->e->set_contributors(mask, count);
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_INVALID; }
}
gr_status gr_bal_comm_ipc_mailbox(gr_bal_problem *p, size_t slot_bytes, int world_size, void *handle_64) {
  if (!p || !p->e || !handle_64 || world_size < 1 || world_size > 64 || slot_bytes < 1024) { g_last_error = "gr_bal_comm_ipc_mailbox: bad argument"; return GR_ERR_INVALID; }
  try {
    GR_HIP(hipSetDevice(p->e->device));
    slot_bytes = (slot_bytes + 255) / 256 * 256;
    const size_t bytes = IpcComm::mailbox_bytes(world_size, slot_bytes);
    char *box = nullptr;
    // FINE-GRAINED device memory: peers write it over xGMI while this GPU's kernels read it; coarse-grained (plain hipMalloc)
    // lines may sit stale in this GPU's L2s across launches, fine-grained ones are coherent at the memory side
    GR_HIP(hipExtMallocWithFlags(reinterpret_cast<void **>(&box), bytes, hipDeviceMallocFinegrained));
    GR_HIP(hipMemset(box, 0, bytes));
    GR_HIP(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    GR_HIP(hipIpcGetMemHandle(&h, box));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    std::memcpy(handle_64, &h, 64);
    if (p->e->ipc_box) (void)hipFree(p->e->ipc_box);
    p->e->ipc_box = box; p->e->ipc_slot = slot_bytes; p->e->ipc_world = world_size;
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_COMM; }
}
gr_status gr_bal_comm_init_ipc(gr_bal_problem *p, const void *handles, int rank, int world_size, const void *unique_id_128, int *used_ipc) {
  if (!p || !p->e || !handles || rank < 0 || rank >= world_size || !p->e->ipc_box || p->e->ipc_world != world_size) { g_last_error = "gr_bal_comm_init_ipc: bad argument (call gr_bal_comm_ipc_mailbox first)"; return GR_ERR_INVALID; }
  try {
    GR_HIP(hipSetDevice(p->e->device));
    // the fallback first: it is also what the ranks agree through, so nothing below may leave a rank before it has
    // taken part in the agreement (a rank that returned early would leave its peers waiting in that all-reduce)
    std::unique_ptr<Comm> rccl;
    if (unique_id_128) rccl.reset(new RcclComm(unique_id_128, rank, world_size));
    std::unique_ptr<IpcComm> ipc;
    bool ok = true;
    std::string why;
    try {
      std::vector<char *> boxes(world_size, nullptr);
      std::vector<bool> opened(world_size, false);
      // gr_bal_tuning.comm_transport: 0 = RCCL only (nothing is opened), 2 = fault injection: the mapping is refused
      if (p->e->tune.comm_transport == 0) { ok = false; why = "gr_bal_tuning.comm_transport = 0: RCCL only"; }
      for (int r = 0; r < world_size && ok; ++r) {
        if (r == rank) { boxes[r] = p->e->ipc_box; continue; }
        if (p->e->tune.comm_transport == 2) { ok = false; why = "peer mapping refused (gr_bal_tuning.comm_transport = 2, fault injection)"; break; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const char *>(handles) + 64 * (size_t)r, 64);
        void *q = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); ok = false; why = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e); break; }
        boxes[r] = static_cast<char *>(q); opened[r] = true;
      }
      if (ok && p->e->tune.comm_transport == 2 && world_size == 1) { ok = false; why = "peer mapping refused (gr_bal_tuning.comm_transport = 2, fault injection)"; }
      if (ok) {
        ipc.reset(new IpcComm(rank, world_size, p->e->ipc_slot, boxes, opened));
        p->e->ipc_box = nullptr; // owned by the communicator from here on
        // the caller's wait bound also governs the start-up self-test, with a floor of 5 s: this is the one collective at
        // which ranks arrive after unequal host-side set-up work (code-object loading, first launches)
        ipc->set_timeout_ms(std::max(5000, p->e->tune.ipc_timeout_ms));
        // start-up verification on the real topology: known values, three sizes, both mailbox sets
        const size_t maxn = ipc->slot_bytes / sizeof(double);
        const double S = 0.5 * world_size * (world_size + 1);
        for (size_t nn : {(size_t)3, std::min<size_t>(1000, maxn), maxn}) {
          std::vector<double> h(nn);
          for (size_t i = 0; i < nn; ++i) h[i] = (double)(rank + 1) * (double)(i % 7 + 1);
          DevBuf<double> d;
          d.upload(h, nullptr);
          ipc->allreduce(d.p, nn, true, nullptr);
          h = d.download(nullptr);
          for (size_t i = 0; i < nn && ok; ++i) ok = h[i] == S * (double)(i % 7 + 1);
          if (ipc->failed()) ok = false;
          if (!ok) { why = "self-test: wrong sum or time-out"; break; }
        }
      } else {
        for (int r = 0; r < world_size; ++r) if (opened[r]) (void)hipIpcCloseMemHandle(boxes[r]);
      }
    } catch (const std::exception &ex) { ok = false; why = ex.what(); }
    if (rccl) { // agree: everybody uses the mailboxes, or nobody does
      DevBuf<double> flag;
      flag.upload(std::vector<double>{ok ? 1.0 : 0.0}, nullptr);
      rccl->allreduce(flag.p, 1, true, nullptr);
      ok = flag.download(nullptr)[0] == (double)world_size;
      if (!ok) {
        ipc.reset();
        if (p->e->tune.verbose) std::fprintf(stderr, "[graphite-mi355x] rank %d: IPC mailboxes not used (%s): every rank drops to RCCL\n", rank, why.empty() ? "a peer refused" : why.c_str());
        p->e->set_comm(std::move(rccl));
        if (used_ipc) *used_ipc = 0;
        return GR_OK;
      }
      ipc->fallback = std::move(rccl);
    } else if (!ok) throw std::runtime_error("IPC all-reduce unavailable (" + why + ") and there is no fallback communicator");
    p->e->set_comm(std::move(ipc));
    if (used_ipc) *used_ipc = 1;
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_COMM; }
}
gr_status gr_bal_comm_allreduce_host(gr_bal_problem *p, double *v, size_t n) {
  if (!p || !p->e || !v) { g_last_error = "gr_bal_comm_allreduce_host: bad argument"; return GR_ERR_INVALID; }
  try { GR_HIP(hipSetDevice(p->e->device)); p->e->allreduce_host(v, n); return GR_OK; }
  catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const CommError &ex) { g_last_error = ex.what(); return GR_ERR_COMM; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_COMM; }
}
// TEST ONLY: the DPP lane exchanges of common.hpp (lane_xor<1 | 2 | 4 | 8>) against __shfl_xor on every lane, 32- and 64-bit;
// *mismatches = number of (offset, lane, width) triples that differ
__global__ void k_test_lane_xor(int *bad) {
  const int lane = threadIdx.x & 63;
  const unsigned u = 0x9E3779B9u * (unsigned)(lane + 1);
  const double d = 1.0 + lane * 0.125 + (double)u * 1e-12;
  int n = 0;
  n += lane_xor_u32<1>(u) != (unsigned)__shfl_xor((int)u, 1, 64); n += lane_xor_u32<2>(u) != (unsigned)__shfl_xor((int)u, 2, 64);
  n += lane_xor_u32<4>(u) != (unsigned)__shfl_xor((int)u, 4, 64); n += lane_xor_u32<8>(u) != (unsigned)__shfl_xor((int)u, 8, 64);
  n += lane_xor<1>(d) != __shfl_xor(d, 1, 64); n += lane_xor<2>(d) != __shfl_xor(d, 2, 64);
  n += lane_xor<4>(d) != __shfl_xor(d, 4, 64); n += lane_xor<8>(d) != __shfl_xor(d, 8, 64);
  double s = d;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  n += wave_allsum(d) != s;
  if (n) atomicAdd(bad, n);
}
gr_status gr_test_lane_xor(int device, int *mismatches) {
  if (!mismatches) { g_last_error = "gr_test_lane_xor: bad argument"; return GR_ERR_INVALID; }
  try {
    GR_HIP(hipSetDevice(device));
    DevBuf<int> bad; bad.alloc(1); bad.zero(nullptr);
    k_test_lane_xor<<<4, 256>>>(bad.p);
    *mismatches = bad.download(nullptr)[0];
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
}
// TEST ONLY: join `n` problems of THIS process (same GPU) into an in-process group; afterwards
// every collective call (linearize, solve, levenberg_marquardt ...) must be made concurrently,
// one host thread per problem.  Exercises the sharded algorithm on a 1-GPU box.
gr_status gr_bal_comm_init_local(gr_bal_problem **problems, int n) {
  if (!problems || n <= 0) { g_last_error = "gr_bal_comm_init_local: bad argument"; return GR_ERR_INVALID; }
  try {
    GR_HIP(hipSetDevice(problems[0]->e->device));
    auto grp = std::make_shared<LocalGroup>(n);
    for (int r = 0; r < n; ++r) problems[r]->e->set_comm(std::unique_ptr<Comm>(new LocalComm(grp, r)));
    return GR_OK;
  } catch (const HipError &ex) { g_last_error = ex.what(); return GR_ERR_HIP; }
  catch (const std::exception &ex) { g_last_error = ex.what(); return GR_ERR_COMM; }
}

} // extern "C"
