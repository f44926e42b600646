/* graphite_mi355x_model.h — the engine of libgraphite_mi355x.so on USER factor / vertex traits.
 *
 * The reference instantiates every hot kernel on the user's traits (ops/linearize.hpp:10-138, ops/error.hpp:253-323,
 * ops/product.hpp:51-103,228-292) and carries a precision matrix and a loss object PER FACTOR through all of them
 * (factor.hpp:158-174, ops/chi2.hpp:34-44, ops/linearize.hpp:283).  Here the split is:
 *
 *   user's translation unit (include/graphite/engine_model.hpp, compiled by hipcc with the user's traits)
 *       the kernels that CALL the traits: linearise (error + Jacobian blocks + loss + precision matrix), chi2,
 *       trial step (backup + Traits::update), revert — in the engine's layout: camera-major observation order,
 *       one observation per lane, wave transpose-reductions into (wave, camera) segments, point records at
 *       point-major slots;
 *   libgraphite_mi355x.so
 *       everything that does not see the traits: orderings and segments, segment / point sums, scales, block
 *       inverses, the PCG operator on the STORED weighted Jacobian (k_pcg_operator_stored), update / direction
 *       kernels, Schur reduction, tile Cholesky, the device-resident PCG loops and the LM loop.
 *
 * The library calls the user-side kernels through the table of launchers below (plain C function pointers; every
 * launcher enqueues on args->stream and returns a hipError_t as int).  A binary factor between a NON-eliminated
 * "pose" vertex type (tangent dimension <= 9) and an eliminated "landmark" vertex type (<= 3) with an error of
 * dimension <= 2 fits: smaller blocks are zero-padded to the engine's 9 / 3 / 2 layout (a padded row or column has a
 * zero Jacobian, hence a zero gradient and an exactly zero step; every other quantity is what the unpadded system gives).
 *
 * Weighted Jacobian: with W = rho'(chi2_f) P_f = L^T L (2 x 2 Cholesky factor, upper), the linearise kernel hands the
 * library Jt = L J and et = L e, so that J^T W J = Jt^T Jt and J^T W e = Jt^T et: one formulation for identity and
 * general precision matrices and for every loss.  P_f must be symmetric positive (semi-)definite.
 */
#ifndef GRAPHITE_MI355X_MODEL_H
#define GRAPHITE_MI355X_MODEL_H

#include "graphite_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GR_MODEL_JSTREAMS 24 /* 2 x 9 pose block (E x d column-major: stream 2 col + row) then 2 x 3 landmark block */

/* Graph::linearize for the factors of the model (graph.hpp:236-290): one launch over the observations in the
 * engine's order.  T = the problem's dtype. */
typedef struct {
  int32_t No, ntiles, grid;               /* observations; xcd_obs_range's ntiles; workgroups of 256 threads to launch   */
  const int32_t *cam, *pt, *pos;          /* [No] pose id, landmark id (engine order), point-major slot of observation j */
  const int32_t *blk_seg, *seg_slot;      /* (64-observation block, camera) segments: kernels_mf.hpp                     */
  void *g9;                               /* T [slot][8]: Jt_l (2 x 3 column-major), et (2)                               */
  void *Hcp;                              /* T [slot][27] (9 x 3 column-major) Jt_c^T Jt_l, or NULL                       */
  void *cam_partial;                      /* T [segment][54]: upper triangle of Jt_c^T Jt_c (45), -Jt_c^T et (9)          */
  double *chi2_partial;                   /* [grid] sum of rho(chi2_f) per workgroup                                      */
  void *jst;                              /* stored weighted Jacobian, GR_MODEL_JSTREAMS streams of jst_stride scalars of
                                             the STORAGE type (gr_model_ops.storage_dtype), observation order; NULL: none */
  int64_t jst_stride;
  const void *lm;                         /* LmDev * or NULL: return at once when lm->stop                                */
  const int32_t *gate;                    /* NULL or a device word: return at once when *gate == 0                        */
  const unsigned char *cam_fixed, *pt_fixed; /* NULL or per-vertex flags: Hcp of a fixed vertex's observation is zero     */
  void *stream;
  /* device-decided Schur loop whose rejected steps do not stop the head: two buffers of camera-point blocks; hsel (an LmDev *, bit 0 of its
     hsel field = the buffer that holds the CURRENT point's blocks: 0 = Hcp, 1 = Hcp_alt) — the lineariser writes the OTHER one.  NULL: Hcp. */
  void *Hcp_alt;
  const void *hsel;
} gr_model_lin_args;

/* Graph::compute_error + chi2 (graph.hpp:212-225) at the current vertex values */
typedef struct {
  int32_t No, grid;
  const int32_t *cam, *pt, *pos;
  double *chi2_partial;                   /* [grid]                                                                        */
  void *res_out;                          /* NULL or T [slot][2]: the residuals                                            */
  void *stream;
} gr_model_chi2_args;

/* matrix-free operator with RECOMPUTED Jacobians (set_jacobian_storage(false), factor.hpp:626-640,
 * ops/product.hpp:103,292): u = J ps, den += u^T W u, camera rows J_c^T W u per segment, landmark rows per slot */
typedef struct {
  int32_t No, Nc, ntiles, grid;
  const int32_t *cam, *pt, *pos;          /* pos == NULL: landmark rows in observation order                               */
  const int32_t *blk_seg, *seg_slot;
  const void *ps;                         /* T [9 Nc + 3 Np]: s .* p                                                       */
  void *g3;                               /* T [slot][3]                                                                   */
  void *op_partial;                       /* T [segment][9]                                                                */
  double *den_slots;                      /* 64 partial sums: atomicAdd(&den_slots[workgroup & 63], partial)               */
  const int32_t *done;                    /* device word: return at once when *done != 0 (may be NULL)                     */
  const void *lm;
  void *stream;
} gr_model_op_args;

/* Graph::backup_parameters + Graph::apply_update (graph.hpp:292-309, ops/update.hpp:11-31): vertex <- update(vertex,
 * dx .* s) through the user's Traits::update, states saved first; plus per-workgroup partials of the compute_rho
 * denominator sum dx (mu dx + s b) (optimizer/levenberg_marquardt.hpp:34-41) */
typedef struct {
  const void *dx, *scales, *bu;           /* T [9 Nc + 3 Np] (padded layout)                                               */
  double mu;
  int32_t with_backup;
  int32_t cam_weight;                     /* 0: the pose part does not count in the rho partials (landmark shards)         */
  double *rho_partial;                    /* [step_blocks] or NULL                                                         */
  const void *lm;
  const int32_t *gate;
  const unsigned char *cam_fixed, *pt_fixed;
  void *stream;
  /* the library's inner-loop state reset riding in this launch (it used to be a launch of its own behind the step): zero the ranges
     clear_ptr[i] .. + clear_bytes[i] (multiples of 4 bytes; 0 = none), except that the double at inf_word (inside one of them, or NULL)
     becomes +infinity.  Under the same gate as the step itself. */
  void *clear_ptr[3];
  int64_t clear_bytes[3];
  double *inf_word;
  /* restore_on_hsel != 0 (only the device-decided Schur loop whose rejected steps do not stop the head sets it), `lm` != NULL and
     bit 1 of its hsel field set: the vertices still sit at a rejected trial point — restore every vertex from its backup first (and
     leave the backup alone), then apply this step.  With restore_on_hsel == 0 the hsel field is never read: a head that stops on a
     rejected step passes `lm` for the gate and the damping only. */
  int32_t restore_on_hsel;
} gr_model_step_args;

typedef struct gr_model_ops {
  void *ctx;
  int32_t pose_dim, landmark_dim, error_dim; /* tangent dimensions of the two vertex types (<= 9, <= 3), error (<= 2)     */
  int32_t storage_dtype;                     /* gr_dtype of the stored Jacobian (Graph<T, S>'s S; GR_F32 under GR_F64 = mixed) */
  int32_t store_jacobians;                   /* 1: jst is written and the library's stored operator runs; 0: `op` is called */
  int32_t step_blocks;                       /* workgroups of `step` (= rho partials it writes)                            */
  int32_t lin_wg_per_cu, op_wg_per_cu;       /* workgroups of `linearize` / `op` resident per CU (occupancy query on the user
                                                side; the library sizes args->grid from them); 0 = 3                         */
  int (*linearize)(void *ctx, const gr_model_lin_args *a);
  int (*chi2)(void *ctx, const gr_model_chi2_args *a);
  int (*op)(void *ctx, const gr_model_op_args *a);       /* may be NULL when store_jacobians != 0                          */
  int (*step)(void *ctx, const gr_model_step_args *a);
  int (*backup)(void *ctx, void *stream);
  int (*revert)(void *ctx, void *stream);
} gr_model_ops;

/* A problem handle whose factor / vertex arithmetic is the user's (`ops`, which must outlive the handle): structure
 * only — local pose and landmark ids per observation (host or device).  Vertex values, observations, constraint data,
 * losses and precision matrices stay on the user side, in the order gr_bal_model_orders reports.  Everything of
 * graphite_mi355x.h that does not name the built-in camera model works on such a handle (linearize, get, solvers,
 * levenberg_marquardt, Schur / direct solvers); gr_bal_set_params / get_params / set_loss / set_jacobian_precision
 * and GR_SOLVER_PCG_SCHUR_IMPLICIT do not (GR_ERR_INVALID). */
gr_status gr_bal_create_model(gr_bal_problem **out, gr_dtype dtype, int64_t num_poses, int64_t num_landmarks,
                              int64_t num_observations, const int32_t *pose_idx, const int32_t *landmark_idx,
                              const gr_model_ops *ops, int device, void *stream);
/* The engine's orders: obs_order[j] = index (into the arrays given at creation) of the observation at position j of the
 * per-observation kernels; landmark_order[q] = caller's landmark index of engine landmark q (landmarks are renumbered by
 * first observing pose).  Host arrays of num_observations / num_landmarks int32; either may be NULL. */
gr_status gr_bal_model_orders(gr_bal_problem *p, int32_t *obs_order, int32_t *landmark_order);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHITE_MI355X_MODEL_H */
