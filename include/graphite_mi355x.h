/* graphite_mi355x.h — C-ABI of libgraphite_mi355x.so
 *
 * The drop-in boundary for the hot path of sfu-rsl/graphite on MI355X
 * (gfx950): per-factor residual/Jacobian -> blocked J^T J / J^T r -> Schur
 * reduction -> PCG inner solve, inside Levenberg-Marquardt, for BAL graphs
 * (camera d=9, point d=3, reprojection factor E=2).
 *
 * The reference has no C ABI: its boundary is a C++17 template/virtual
 * interface consumed from .cu files (docs/markdown/main.md:54-55).  Each entry
 * point below names the reference member function it replaces (file:line are
 * relative to /root/reference).  The C++ mirror of that interface
 * (include/graphite/) is a thin layer over these calls; INTEGRATION.md shows
 * the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  `dtype` selects
 * the scalar type T (= S) of every `void*` array: GR_F32 or GR_F64.  Unless a
 * parameter says "host", data pointers may be host or device pointers
 * (detected with hipPointerGetAttributes).  Every function returns a
 * gr_status; nothing throws.  All work of one problem handle is issued on the
 * hipStream_t given at creation (0 = the null stream).  Functions that return
 * scalars to the host synchronise that stream; the others are asynchronous.
 */
#ifndef GRAPHITE_MI355X_H
#define GRAPHITE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GR_OK = 0,
  GR_ERR_INVALID = 1,        /* bad argument / wrong state                         */
  GR_ERR_HIP = 2,            /* a HIP runtime call failed (gr_last_error_string)    */
  GR_ERR_NO_DEVICE = 3,      /* no gfx950 device visible                            */
  GR_ERR_DUPLICATE_EDGE = 4, /* the same (camera, point) pair appears twice         */
  GR_ERR_SOLVE_FAILED = 5,   /* solver->solve() returned false                      */
  GR_ERR_COMM = 6            /* RCCL call failed                                    */
} gr_status;

typedef enum { GR_F32 = 0, GR_F64 = 1 } gr_dtype;

/* Solver<T,S> implementations (solver/solver.hpp:12-25) */
typedef enum {
  GR_SOLVER_PCG_SCHUR = 0,    /* PCGSchurSolver + BlockJacobiSchurPreconditioner (solver/pcg_schur.hpp, preconditioner/block_jacobi_schur.hpp) */
  GR_SOLVER_PCG = 1,          /* PCGSolver + BlockJacobiPreconditioner (solver/pcg.hpp, preconditioner/block_jacobi.hpp)                      */
  GR_SOLVER_PCG_IDENTITY = 2, /* PCGSolver + IdentityPreconditioner (preconditioner/identity.hpp)                                              */
  GR_SOLVER_PCG_SCHUR_IMPLICIT = 3, /* same iterates as GR_SOLVER_PCG_SCHUR, S applied as Hpp - Hpl Hll^-1 Hpl^T without forming it (kernels_is.hpp) */
  GR_SOLVER_DENSE_SCHUR = 4   /* direct solve of S x = b_S: EigenSchurLDLTSolver (solver/eigen_schur.hpp:71-108) / cudssSchurSolver
                                 (solver/cudss_schur.hpp:190-234), as a tile-sparse dense MFMA Cholesky (chol.hpp)                         */
} gr_solver;

typedef enum { GR_LOSS_DEFAULT = 0, GR_LOSS_HUBER = 1 } gr_loss; /* loss.hpp:15-51 */

/* arrays retrievable with gr_bal_get (all in the reference's column-SCALED space) */
typedef enum {
  GR_GET_SCALES = 0,   /* n        jacobian_scales                  graph.hpp:262-270            */
  GR_GET_B = 1,        /* n        b = -J^T rho' P r                ops/linearize.hpp:240-303    */
  GR_GET_HCC = 2,      /* 81 Nc    camera diagonal blocks of H      ops/hessian.hpp:10-78        */
  GR_GET_HCP = 3,      /* 27 No    camera-point blocks, INPUT observation order, 9x3 col-major   */
  GR_GET_HLL = 4,      /* 9 Np     point diagonal blocks of H                                    */
  GR_GET_S = 5,        /* 81 nnzb  Schur complement blocks (upper, column-major block order)     */
  GR_GET_B_SCHUR = 6,  /* 9 Nc     b_S                              schur.hpp:901-920            */
  GR_GET_HLL_INV = 7,  /* 9 Np     (Hll + damping)^-1               schur.hpp:1067-1114          */
  GR_GET_RESIDUALS = 8,/* 2 No     residuals, INPUT observation order ops/error.hpp:253          */
  GR_GET_H = 9         /* 81 Nc + 27 No + 9 Np   the Hessian in the reference's value layout: upper block-CSC, block columns =
                          cameras then points (caller's order), blocks sorted by row inside a column with the diagonal block
                          last, every block column-major (hessian.hpp:257-288, :123-126); structure: gr_bal_hessian_structure */
} gr_bal_array;

typedef struct gr_bal_problem gr_bal_problem; /* opaque */

/* optimizer::LevenbergMarquardtOptions (optimizer/levenberg_marquardt.hpp:52-98)
 * + the PCG constructor arguments (solver/pcg.hpp:35-40, pcg_schur.hpp:42-47). */
typedef struct {
  int32_t solver;           /* gr_solver */
  int32_t iterations;       /* default 10 */
  double initial_damping;   /* default 1e-4 */
  int32_t use_identity;     /* identity instead of diagonal-scaled damping */
  int32_t pcg_max_iter;     /* bal.cu default 10 */
  double pcg_tol;           /* bal.cu default 1.0 */
  double pcg_rejection_ratio; /* bal.cu default 5.0 */
  int32_t profile;          /* record HIP events around the dominant kernels */
  int32_t early_stop;       /* levenberg_marquardt2 (:255-418): also leave after 3 consecutive accepted
                               iterations that each lower chi2 by less than 0.1 % */
  const volatile unsigned char *stop_flag; /* LevenbergMarquardtOptions::stop_flag (a host `bool *`, may be NULL):
                               polled once per LM iteration, after it, as levenberg_marquardt.hpp:233-238 does.  On landmark
                               shards every rank must pass a flag or none; the ranks agree on it (one scalar all-reduce per
                               iteration), so they leave the loop together when ANY rank has raised its flag */
} gr_lm_options;

typedef struct {
  int32_t iterations_run;
  int32_t accepted;
  int32_t pcg_iterations;   /* total inner iterations */
  int32_t ok;               /* return value of levenberg_marquardt() */
  double setup_seconds;     /* update_structure + first linearize + update_values */
  double loop_seconds;      /* the LM for-loop (host wall clock, stream synchronised) */
  double solve_seconds;     /* device time inside solver->solve.  Matrix-free PCG in its fused form (single GPU):
                               device wall-clock stamps, always filled.  Every other solver / landmark shards: HIP
                               events around the solve, recorded only with options->profile != 0 (an event between
                               two launches costs a ~6 us bubble), 0 otherwise */
  double final_chi2;
  int64_t collectives;      /* landmark-sharded runs: collective operations (grouped calls count once; a message fused
                               into the operator / update launches counts once too) issued inside the LM loop by this
                               rank; 0 without a communicator */
  int64_t kernel_launches;  /* kernels launched inside the LM loop by the matrix-free PCG paths, the communicator's own
                               (one per mailbox message) included: what the sharded iteration costs in launch floors */
  int64_t fused_messages;   /* of `collectives`: messages that travelled inside the producing / consuming launches
                               (gr_bal_tuning.shard_fused) instead of through an all-reduce kernel of their own */
} gr_lm_stats;

/* Tuning of one problem handle.  The reference configures through option structs only (LevenbergMarquardtOptions,
 * optimizer/levenberg_marquardt.hpp:52-98; cudssSolverOptions, solver/cudss.hpp:12-31) and so does this library: every
 * choice that changes WHICH kernels run lives here (results are the same whatever is chosen: the parity suite runs the
 * combinations).  -1 = decide per problem (the default; the timed choices are made once, at solver_update_structure).
 * gr_bal_tuning_default fills the defaults and then applies the GR_* environment variables named below — a debugging
 * override read ONCE, when a problem is created or gr_bal_tuning_default is called, never inside a solve. */
typedef struct {
  int32_t point_tiles;          /* GR_PTILES        -1 auto | 0 plain camera-major order | K point tiles (Engine::build_tiled_order) */
  int32_t g3_gather;            /* GR_G3_GATHER     -1 auto (with point tiles) | 0 | 1: operator output in observation order         */
  int32_t point_records;        /* GR_POINT_RECORDS -1 auto (timed) | 0 | 1: [X Y Z | s.p] records for the operator's gathers        */
  int32_t pcg_lazy;             /* GR_PCG_LAZY      -1 auto (= 0 since round 3) | 0 | 1: no direction kernel (host-driven LM loop)   */
  int32_t pcg_single_reduction; /* GR_PCG_CG        -1 auto (with a communicator of > 1 rank) | 0 | 1                                */
  int32_t sparse_cholesky;      /* GR_SPARSE_CHOL   -1 auto (when the camera graph dissects) | 0 dense tile Cholesky | 1             */
  int32_t spchol_overlap;       /* GR_SPCHOL_OVERLAP 1: a level's forward substitution rides in its update launch | 2: second stream | 0: after */
  int32_t lm_speculate;         /* GR_LM_SPECULATE   1: trial chi2 from a speculative linearisation on accept streaks                */
  int32_t lm_ahead;             /* GR_LM_AHEAD       1: trial linearisation enqueued ahead of the PCG exit flag                      */
  int32_t lm_fused;             /* GR_LM_FUSED       1: fused iteration head (k_finalize_bj), trial step inside the last direction   */
  int32_t grid_mult;            /* GR_GRID_MULT      0: workgroups per CU of the persistent per-observation kernels; 0 = as many as
                                                       are resident at once (occupancy query per kernel)                             */
  int32_t vec_per_thread;       /* GR_VEC_PER_THREAD 2: elements per thread of the light vector kernels                             */
  int32_t schur_item;           /* GR_SCHUR_ITEM     56: products per work item of the explicit Schur reduction                     */
  int32_t verbose;              /* GR_VERBOSE        0: report the per-problem choices on stderr                                    */
  int32_t ipc_timeout_ms;       /* GR_IPC_TIMEOUT_MS 30000: bound of one IPC-mailbox all-reduce wait (gr_bal_comm_init_ipc)         */
  int32_t shard_fused;          /* GR_SHARD_FUSED   -1 auto (IPC mailboxes, single-reduction PCG, plain observation order) | 0 | 1:
                                   the per-inner-iteration message is pushed by the operator launch's finishing workgroups, its
                                   last workgroup waits for the peers' messages, the update launch sums the camera rows from the
                                   mailbox (2 launches per inner iteration instead of 4).  0 keeps a kernel of its own for the
                                   all-reduce                                                                                    */
  int32_t shard_virtual_ranks;  /* PROJECTION ONLY (tools/shard_projection.py): on a ONE-rank mailbox communicator the fused message is
                                   pushed / awaited / summed as if V ranks took part (V slots of the own mailbox)                    */
  int32_t chol_fuse;            /* GR_CHOL_FUSE      1: dense tile Cholesky factorises the next diagonal tile inside the trailing update   */
  int32_t chol_pin;             /* GR_CHOL_PIN       1: its C tile parked in VGPRs (one workgroup per CU in fp64)                          */
  int32_t spchol_fuse;          /* GR_SPCHOL_FUSE    1: the same fusion in the nested-dissection form; 2 (default): that tile's update by quadrants on three workgroups */
  int32_t spchol_slice;         /* GR_SPCHOL_SLICE   1: tiles per work item of its triangular solves                                       */
  int32_t schur_fused;          /* GR_SCHUR_FUSED   -1 auto | 0 | 1: GR_SOLVER_PCG_SCHUR — S and the b_S partials in ONE launch whose multi-item
                                   blocks are finished by their last arriver (no float atomics), and, where the reduced system is small
                                   (cameras <= 2 x CUs), the device-decided LM iteration of kernels_sf.hpp: five launches, the whole PCG
                                   on S inside one of them — and (auto / 2, built-in model) a rejected step does not stop that head: the
                                   finalisation takes the vertices back, keeps its sums, raises the damping and the head goes on as the
                                   next iteration's (second buffer of camera-point blocks).  1: the head stops on a rejected step and the
                                   host runs the rejection.  0: the round-4 kernels and the host-driven loop                              */
  int32_t spchol_bwd_chain;     /* GR_SPCHOL_BWD_CHAIN 1: the backward substitution of the nested-dissection Cholesky as ONE dependency-driven
                                   launch (items fetch their tile into registers before waiting for x_i); 0: one launch per level       */
  int32_t pcg_resident;         /* GR_PCG_RESIDENT  -1 auto (= 0: measured slower) | 0 | 1: GR_SOLVER_PCG / _IDENTITY inside the LM loop (built-in model, one GPU, plain
                                   camera-major order, a problem whose per-lane share fits: observations <= 6 x 512 x CUs, tiles of 170
                                   points / 56 cameras <= 4 x CUs) — the WHOLE inner solve and the trial step as ONE resident launch
                                   (kernels_rp.hpp: one 512-thread workgroup per CU, the lane's observations and the thread's x, r, p, z'
                                   in LDS / VGPRs for all inner iterations, two grid barriers per iteration).  0: operator / update /
                                   direction launches                                                                                  */
  int32_t comm_transport;       /* GR_COMM_TRANSPORT -1 auto | 1: gr_bal_comm_init_ipc opens the peers' mailboxes, verifies them and keeps RCCL for
                                   what does not fit a slot | 0: RCCL only — the mailboxes are not opened, every rank agrees on the fallback
                                   communicator | 2 (fault injection): the peer mapping is treated as refused AFTER the handles were
                                   exchanged, i.e. the path a node without peer access takes: the ranks agree to drop to RCCL together.
                                   0 and 2 need the fallback communicator (unique id); without one gr_bal_comm_init_ipc fails          */
} gr_bal_tuning;
void gr_bal_tuning_default(gr_bal_tuning *t);

/* "graphite-mi355x <major.minor> (gfx950)".  ABI note: 0.2 = the layouts below (sizeof(gr_bal_tuning) == 100, sizeof(gr_lm_stats) == 80);
 * a caller built against an earlier header (80 / 64 bytes) must be rebuilt — the library writes the whole struct.  New tuning fields
 * since then took the `reserved` slots (pcg_resident, comm_transport): the size stays at 100, there are none left. */
const char *gr_version(void);
const char *gr_last_error_string(void);
/* number of visible HIP devices (0 if none); never initialises a device */
int gr_device_count(void);
/* Loads this library's code object on `device` (one empty kernel launch, synchronised): the one-time cost of the first
 * kernel launch of a process, taken where the caller wants it (graph construction) instead of inside the first
 * gr_bal_model_evaluate / gr_bal_create of an optimiser call.  Idempotent. */
gr_status gr_warm_up(int device);

/* ---- problem handle ------------------------------------------------------------
 * Replaces the construction done by examples/bal.cu:55-141 (descriptors,
 * add_vertex/add_factor) plus Graph::initialize_optimization (graph.hpp:92-167):
 * cameras take Hessian block columns 0..Nc-1, points (eliminated) Nc..Nc+Np-1.
 * cameras: Nc x 9 [r(3) t(3) f k1 k2]; points: Np x 3; obs: No x 2;
 * cam_idx/pt_idx: No int32 (host or device).  `device` is the HIP ordinal,
 * `stream` a hipStream_t (may be NULL). */
gr_status gr_bal_create(gr_bal_problem **out, gr_dtype dtype, int64_t num_cameras,
                        int64_t num_points, int64_t num_observations, const void *cameras,
                        const void *points, const void *observations, const int32_t *cam_idx,
                        const int32_t *pt_idx, int device, void *stream);
/* Same, for ONE landmark shard of a multi-GPU problem: all cameras (some of which may have no
 * observation in this shard), a subset of the points and exactly their observations
 * (point indices local to the shard).  Must be followed by gr_bal_comm_init. */
gr_status gr_bal_create_shard(gr_bal_problem **out, gr_dtype dtype, int64_t num_cameras,
                              int64_t num_points, int64_t num_observations, const void *cameras,
                              const void *points, const void *observations, const int32_t *cam_idx,
                              const int32_t *pt_idx, int device, void *stream);
gr_status gr_bal_destroy(gr_bal_problem *p);
/* replace / read the tuning of a problem (created with gr_bal_tuning_default's values).  Set it before
 * gr_bal_solver_update_structure / gr_bal_levenberg_marquardt; choices already timed are re-made. */
gr_status gr_bal_set_tuning(gr_bal_problem *p, const gr_bal_tuning *t);
gr_status gr_bal_get_tuning(gr_bal_problem *p, gr_bal_tuning *t);

/* FactorDescriptor::add_factor loss argument (factor.hpp:373-412), one loss for all factors */
gr_status gr_bal_set_loss(gr_bal_problem *p, gr_loss kind, double delta);
/* Graph::scale_system (graph.hpp:331) */
gr_status gr_bal_set_scale_system(gr_bal_problem *p, int enable);
/* Mixed precision, the reference's Graph<T = double, S = float> (examples/bal.cu:338-345, --precision FP64-FP32):
 * on a GR_F64 problem, GR_F32 makes every kernel evaluate the Jacobian entries in fp32 (residuals, all sums,
 * the PCG vectors and dot products stay fp64).  Jacobians are never stored here, so this changes arithmetic,
 * not memory traffic.  GR_F64 restores the default. */
/* VertexDescriptor::set_fixed (vertex.hpp:262-264): cam_fixed[Nc] / pt_fixed[Np] (caller's vertex order), non-zero = the
 * vertex keeps its value; NULL = none of that type.  The reference gives a fixed vertex no Hessian column and skips its
 * Jacobian blocks (ops/linearize.hpp:24, ops/hessian.hpp:95); here its column stays, empty, and its step is exactly 0.
 * Supported by every solver.
 * Host pointers. */
gr_status gr_bal_set_fixed(gr_bal_problem *p, const unsigned char *cam_fixed, const unsigned char *pt_fixed);
gr_status gr_bal_set_jacobian_precision(gr_bal_problem *p, gr_dtype dtype);

/* vertex values (user-owned Vertex* in the reference, vertex.hpp:65) */
gr_status gr_bal_set_params(gr_bal_problem *p, const void *cameras, const void *points);
gr_status gr_bal_get_params(gr_bal_problem *p, void *cameras, void *points);

/* Graph::linearize (graph.hpp:236-290) */
gr_status gr_bal_linearize(gr_bal_problem *p);
/* Graph::compute_error (graph.hpp:212-217) followed by Graph::chi2 (graph.hpp:219-225) */
gr_status gr_bal_chi2(gr_bal_problem *p, double *chi2);
/* Graph::backup_parameters / revert_parameters / apply_update (graph.hpp:292-318);
 * delta_x: n scalars in the scaled space (host or device) */
gr_status gr_bal_backup_parameters(gr_bal_problem *p);
gr_status gr_bal_revert_parameters(gr_bal_problem *p);
gr_status gr_bal_apply_update(gr_bal_problem *p, const void *delta_x);

/* Solver<T,S>::update_structure / update_values / set_damping_factor / solve
 * (solver/solver.hpp:12-25).  delta_x: n scalars (host or device), iterations:
 * number of inner PCG iterations executed (may be NULL). */
gr_status gr_bal_solver_update_structure(gr_bal_problem *p, gr_solver solver);
gr_status gr_bal_solver_update_values(gr_bal_problem *p, gr_solver solver);
gr_status gr_bal_solver_set_damping(gr_bal_problem *p, gr_solver solver, double mu, int use_identity);
gr_status gr_bal_solver_solve(gr_bal_problem *p, gr_solver solver, int max_iter, double tol,
                              double rejection_ratio, void *delta_x, int *iterations);

/* SchurComplement::update_values (schur.hpp:227-235): S, b_S, Hll^-1 with the current damping */
gr_status gr_bal_schur_update_values(gr_bal_problem *p);
/* SchurComplement::execute_schur_vector_multiply (schur.hpp:347-393): y = S x, 9 Nc scalars */
gr_status gr_bal_schur_matvec(gr_bal_problem *p, const void *x, void *y);
/* SchurComplement::compute_landmark_update (schur.hpp:279-302): xl (3 Np) from xp (9 Nc) */
gr_status gr_bal_landmark_update(gr_bal_problem *p, const void *xp, void *xl);
/* block structure of S: colptr (Nc+1) and rowidx (nnzb) int64 host arrays; pass NULL to size */
gr_status gr_bal_schur_structure(gr_bal_problem *p, int64_t *nnzb, int64_t *colptr, int64_t *rowidx);

/* Hessian::build_structure (hessian.hpp:257-288) + csc::build_block_csc_indices (csc_utils.hpp:16-50): block-CSC of the
 * upper Hessian: colptr (Nc + Np + 1), rowidx and value offsets (nblocks = Nc + No + Np each), int64 host arrays; NULL to size */
gr_status gr_bal_hessian_structure(gr_bal_problem *p, int64_t *nblocks, int64_t *colptr, int64_t *rowidx, int64_t *offsets);
/* Hessian::build_csc_structure + update_csc_values / SchurComplement::build_csc_structure + update_csc_values
 * (hessian.hpp:310-324, schur.hpp:264-277, csc_utils.hpp:74-193): scalar CSC of the UPPER triangle of H (which = 0, n columns)
 * or of S with the current damping (which = 1, 9 Nc columns; gr_bal_schur_update_values first).  indptr (columns + 1) and
 * indices (nnz): int64 host arrays; values: nnz scalars of the problem's dtype (host).  Pass NULL arrays to get *nnz. */
gr_status gr_bal_export_csc(gr_bal_problem *p, int which, int64_t *nnz, int64_t *indptr, int64_t *indices, void *values);

/* What the direct solver of the reduced camera system set up (valid after gr_bal_solver_update_structure(GR_SOLVER_DENSE_SCHUR)):
 * the counterpart of the analysis phase's report of cudssSchurSolver (solver/cudss_schur.hpp:146-170: CUDSS_PHASE_ANALYSIS) /
 * SimplicialLDLT::analyzePattern (src/eigen_solver.cpp:10-19).  factor_bytes follows the structurally non-zero 128 x 128
 * tiles of L (fill included) in the nested-dissection form; the padded dense triangle is only taken when S does not dissect. */
typedef struct {
  int32_t sparse;        /* 1: nested dissection + level-scheduled tile Cholesky on tile-sparse storage; 0: dense tile Cholesky */
  int32_t tile_columns;  /* 128-column panels after padding                                                                   */
  int32_t levels;        /* height of the tile elimination tree = length of the launch chain (dense: tile_columns)            */
  int32_t supernodes;    /* nodes of the dissection tree (dense: 1)                                                           */
  int64_t padded_n;      /* 128 * tile_columns                                                                                */
  int64_t factor_tiles;  /* structurally non-zero lower tiles of L                                                            */
  int64_t factor_bytes;  /* device memory of the factor and the per-panel inverses                                            */
  int64_t dense_bytes;   /* padded_n^2 scalars: what a dense array would take                                                 */
} gr_direct_solver_info;
gr_status gr_bal_direct_solver_info(gr_bal_problem *p, gr_direct_solver_info *info);

/* copy one array (gr_bal_array) to `out` (host or device); *count receives its length */
gr_status gr_bal_get(gr_bal_problem *p, gr_bal_array which, void *out, int64_t *count);

/* optimizer::levenberg_marquardt (optimizer/levenberg_marquardt.hpp:110-242).
 * chi2_trace / lambda_trace: host double[iterations+1] (entry 0 = initial), may be NULL. */
gr_status gr_bal_levenberg_marquardt(gr_bal_problem *p, const gr_lm_options *options,
                                     gr_lm_stats *stats, double *chi2_trace, double *lambda_trace);

/* Per-iteration wall time of the LAST gr_bal_levenberg_marquardt call: seconds[i] = host time between observing the
 * decisions of iterations i - 1 and i (the first from the start of the loop) — the "Time" column of the reference's
 * verbose table (optimizer/levenberg_marquardt.hpp:216-221).  In the fused form the host only observes decisions the
 * device has taken, so these are completion times, not enqueue times.  *n = iterations run; up to cap entries are written. */
gr_status gr_bal_lm_iteration_seconds(gr_bal_problem *p, double *seconds, int cap, int *n);

/* The camera model the engine is specialised for, on caller-supplied triples: for i < n, camera i (9 scalars
 * [angle-axis r(3), t(3), f, k1, k2]), point i (3) and observation i (2) give residual i (2), the camera block Jc
 * (2 x 9, column-major, 18 scalars) and the point block Jp (2 x 3, column-major, 6) — the values the user traits of
 * examples/bal.cuh:61-89 produce through bal_reprojection_error_simple (examples/reprojection_error.cuh:61-99) and
 * bal_jacobian_simple (examples/reprojection_error.cuh:104-126, examples/projection_jacobians.cuh:2-322).  The
 * header-only layer (include/graphite/solve.hpp) calls it to VERIFY that a user's error()/jacobian() are this model
 * before it hands a graph to the engine.  All pointers host or device; residuals / Jc / Jp may be NULL. */
gr_status gr_bal_model_evaluate(gr_dtype dtype, int64_t n, const void *cameras, const void *points,
                                const void *observations, void *residuals, void *Jc, void *Jp, int device, void *stream);

/* ---- measurement -----------------------------------------------------------------
 * HIP-event timing of the kernels of the last gr_bal_levenberg_marquardt call with
 * options->profile != 0.  Fills up to `cap` entries; returns the number of distinct
 * kernels in *n.  Times are device milliseconds summed over `launches` launches. */
typedef struct {
  char name[48];
  int64_t launches;
  double total_ms;
  double bytes_per_launch; /* algorithmic HBM bytes per launch */
  double flops_per_launch;
  int64_t active_launches; /* launches minus the look-ahead launches that found the PCG loop already finished
                              (those return at once and move no data); rooflines are priced on these */
} gr_kernel_stat;
gr_status gr_bal_kernel_stats(gr_bal_problem *p, gr_kernel_stat *out, int cap, int *n);

/* ---- multi-GPU (RCCL over xGMI) ----------------------------------------------------
 * One process per GPU.  Each rank creates its problem (gr_bal_create_shard) from ITS landmark
 * partition (all cameras, a contiguous range of points and all their observations);
 * camera-space sums are all-reduced.  Every solver runs sharded: GR_SOLVER_PCG / _IDENTITY (two small
 * all-reduces per inner iteration), GR_SOLVER_PCG_SCHUR_IMPLICIT (one all-reduce of a 9 Nc vector per inner
 * iteration), GR_SOLVER_PCG_SCHUR / GR_SOLVER_DENSE_SCHUR (S and b_S all-reduced once per LM iteration,
 * the reduced solve replicated; for small camera counts).  Calls that involve a collective
 * (linearize, solver_update_structure, schur_update_values, solver_solve, chi2, levenberg_marquardt)
 * must be made by all ranks in the same order.  unique_id: 128-byte ncclUniqueId produced on
 * rank 0 by gr_comm_unique_id and broadcast by the caller (e.g. torch.distributed). */
gr_status gr_comm_unique_id(void *unique_id_128);
gr_status gr_bal_comm_init(gr_bal_problem *p, const void *unique_id_128, int rank, int world_size);
/* One-shot peer all-reduce for the small messages of this solver (csrc/comm.hpp IpcComm): every rank owns a mailbox in its
 * HBM that every peer maps (hipIpcMemHandle) and writes into directly over xGMI; an all-reduce is one hop instead of a
 * ring.  (1) every rank calls gr_bal_comm_ipc_mailbox and gets the 64-byte handle of its mailbox; (2) the caller
 * gathers the handles of all ranks, rank order (e.g. torch.distributed.all_gather); (3) every rank calls
 * gr_bal_comm_init_ipc with the world_size x 64 bytes.  Messages larger than slot_bytes travel through RCCL
 * (unique_id_128 as for gr_bal_comm_init; NULL = no fallback, such a message is then an error).  With a fallback the
 * mailboxes are verified at start-up by all-reducing known values; if any rank sees a wrong sum or a time-out, every rank
 * drops to RCCL alone.  *used_ipc (optional) reports the outcome. */
gr_status gr_bal_comm_ipc_mailbox(gr_bal_problem *p, size_t slot_bytes, int world_size, void *handle_64);
gr_status gr_bal_comm_init_ipc(gr_bal_problem *p, const void *handles_world_x_64, int rank, int world_size,
                               const void *unique_id_128, int *used_ipc);
/* Which ranks hold observations of which camera: mask[c] bit r (world_size <= 32).  The mailbox transport agrees on these masks itself
 * (gr_bal_solver_update_structure: one host all-reduce) — a rank then pushes only the camera rows it holds and sums only the
 * contributors' slots.  This call overrides them; it exists for tools/shard_projection.py, where ONE rank plays all the ranks of a
 * message (gr_bal_tuning.shard_virtual_ranks) and must be told the masks of the real partition.  count = number of cameras. */
gr_status gr_bal_comm_set_contributors(gr_bal_problem *p, const uint32_t *mask, int64_t count);
/* What the first real multi-GPU run is audited with: which communicator this problem ended up with AFTER the start-up self-test
 * of gr_bal_comm_init_ipc (a failed mailbox verification drops every rank to RCCL), how many ranks RCCL itself sees
 * (ncclCommCount), how many peer mailboxes were mapped, and whether the ranks agreed on the fused inner-iteration message. */
typedef struct {
  int32_t rank, size;         /* of the communicator in use; size 0 = none                                                     */
  int32_t transport;          /* 0 none | 1 RCCL | 2 IPC mailboxes (RCCL behind them for large messages when rccl_ranks > 0) | 3 in-process test group */
  int32_t rccl_ranks;         /* ncclCommCount of the RCCL communicator (0: none, -1: the call failed)                         */
  int32_t mailboxes_opened;   /* peer mailboxes mapped through hipIpcOpenMemHandle: size - 1 when every peer is reachable      */
  int32_t device;             /* HIP ordinal of this rank                                                                      */
  int32_t fused_agreed;       /* the ranks agreed on the fused inner-iteration message (gr_bal_tuning.shard_fused)             */
  int32_t reserved;
  int64_t oneshot_messages;   /* mailbox all-reduces since the communicator was created                                        */
  int64_t fallback_messages;  /* ... of the collectives, those that went through the fallback (messages beyond a slot)         */
} gr_comm_info;
gr_status gr_bal_comm_info(gr_bal_problem *p, gr_comm_info *info);
/* Dense SPD solve A x = b on the MFMA Cholesky that GR_SOLVER_DENSE_SCHUR uses (the numerical role of
 * Eigen::SimplicialLDLT in src/eigen_solver.cpp:8-30 / cuDSS in solver/cudss.hpp:183-256 once S is dense).
 * A: n x n row-major, leading dimension lda, lower triangle read; A, b, x host or device pointers (x may
 * alias b).  factor_seconds (optional): device time of the factorisation alone.
 * GR_ERR_SOLVE_FAILED when a pivot is not positive. */
gr_status gr_dense_cholesky_solve(gr_dtype dtype, int64_t n, const void *A, int64_t lda, const void *b, void *x,
                                  int device, void *stream, double *factor_seconds);

/* Sparse SPD solve A x = b of a BLOCK-sparse matrix on the nested-dissection tile Cholesky that GR_SOLVER_DENSE_SCHUR uses for the reduced
 * camera system (sparse_chol.hpp) — the numerical role of Eigen::SimplicialLDLT in solver/eigen.hpp:49-98 (EigenLDLTSolver on a graph
 * without an elimination order, e.g. a pose graph) and of cuDSS in solver/cudss.hpp:183-256.  Structure once, values per solve.
 *   gr_spchol_create: num_nodes block rows / columns of block_size scalars each; num_blocks UPPER blocks (block_row[q] <= block_col[q], every
 *     diagonal block present), host arrays.  GR_ERR_SOLVE_FAILED when the node graph does not dissect into more than one supernode (then the
 *     matrix is as good as dense: use gr_dense_cholesky_solve).
 *   gr_spchol_factor_solve: blocks = num_blocks x block_size x block_size scalars, block q column-major, in the order given to create;
 *     blocks, b, x device OR host pointers, each on its own (host ones are staged through the handle; x may alias b).  GR_ERR_SOLVE_FAILED
 *     when a pivot is not positive.
 *   gr_spchol_info: sizes of the factor (gr_direct_solver_info, as gr_bal_direct_solver_info reports them). */
typedef struct gr_spchol gr_spchol;
gr_status gr_spchol_create(gr_spchol **out, gr_dtype dtype, int64_t num_nodes, int32_t block_size, int64_t num_blocks,
                           const int64_t *block_row, const int64_t *block_col, int device, void *stream);
gr_status gr_spchol_factor_solve(gr_spchol *h, const void *blocks, const void *b, void *x);
gr_status gr_spchol_info(const gr_spchol *h, gr_direct_solver_info *info);
void gr_spchol_destroy(gr_spchol *h);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHITE_MI355X_H */
