// ORACLE — test infrastructure only (see bal_model.hpp header).
// extern "C" surface for ctypes (oracle/__init__.py).  _f32 / _f64 variants.
#include "bal_pipeline.hpp"
#include "cpu_baseline.hpp"
#include "circle_fit.hpp"

using namespace gro;

#define GRO_FOR_T(X) X(float, f32) X(double, f64)

extern "C" {

#define X(T, SFX)                                                                                   \
  void gro_bal_residual_##SFX(const T *cam, const T *pt, const T *obs, T *res) {                    \
    bal_residual<T>(cam, pt, obs, res);                                                             \
  }                                                                                                 \
  void gro_bal_residual_jacobian_##SFX(const T *cam, const T *pt, const T *obs, T *res, T *Jc,      \
                                       T *Jp) {                                                     \
    bal_residual_jacobian<T>(cam, pt, obs, res, Jc, Jp);                                            \
  }                                                                                                 \
  int gro_small_inverse_##SFX(int n, const T *A, T *Ainv) { return small_inverse<T>(n, A, Ainv); }  \
  void gro_chi2_##SFX(size_t n, int E, const T *residuals, const T *pmat, const int *loss_kind,     \
                      const T *loss_delta, T *chi2, T *dchi2) {                                     \
    chi2_kernel<T>(n, E, residuals, pmat, loss_kind, loss_delta, chi2, dchi2);                      \
  }                                                                                                 \
  void gro_scalar_diagonal_##SFX(size_t na, const size_t *active_ids, int E, int d, const T *jac,   \
                                 const size_t *ids, size_t N, size_t I, const size_t *hessian_ids,  \
                                 const uint8_t *active_state, const T *pmat, const T *dchi2,        \
                                 T *diagonal) {                                                     \
    Slot<T> s{d, jac, ids, N, I, hessian_ids, active_state};                                        \
    scalar_diagonal_kernel<T>(na, active_ids, E, s, pmat, dchi2, diagonal);                         \
  }                                                                                                 \
  void gro_block_diagonal_##SFX(size_t na, const size_t *active_ids, int E, int d, const T *jac,    \
                                const size_t *ids, size_t N, size_t I, const size_t *hessian_ids,   \
                                const uint8_t *active_state, const T *pmat, const T *dchi2,         \
                                T *blocks) {                                                        \
    Slot<T> s{d, jac, ids, N, I, hessian_ids, active_state};                                        \
    block_diagonal_kernel<T>(na, active_ids, E, s, pmat, dchi2, blocks);                            \
  }                                                                                                 \
  void gro_scale_jacobians_##SFX(size_t na, const size_t *active_ids, int E, int d, T *jac,         \
                                 const size_t *ids, size_t N, size_t I, const size_t *hessian_ids,  \
                                 const uint8_t *active_state, const T *scales) {                    \
    scale_jacobians_kernel<T>(na, active_ids, E, d, jac, ids, N, I, hessian_ids, active_state,      \
                              scales);                                                              \
  }                                                                                                 \
  void gro_compute_b_##SFX(size_t na, const size_t *active_ids, int E, int d, const T *jac,         \
                           const size_t *ids, size_t N, size_t I, const size_t *hessian_ids,        \
                           const uint8_t *active_state, const T *residuals, const T *pmat,          \
                           const T *dchi2, T *b) {                                                  \
    Slot<T> s{d, jac, ids, N, I, hessian_ids, active_state};                                        \
    compute_b_kernel<T>(na, active_ids, E, s, residuals, pmat, dchi2, b);                           \
  }                                                                                                 \
  void gro_Jv_##SFX(size_t na, const size_t *active_ids, int E, int d, const T *jac,                \
                    const size_t *ids, size_t N, size_t I, const size_t *hessian_ids,               \
                    const uint8_t *active_state, const T *x, T *y) {                                \
    Slot<T> s{d, jac, ids, N, I, hessian_ids, active_state};                                        \
    Jv_kernel<T>(na, active_ids, E, s, x, y);                                                       \
  }                                                                                                 \
  void gro_JtPv_##SFX(size_t na, const size_t *active_ids, int E, int d, const T *jac,              \
                      const size_t *ids, size_t N, size_t I, const size_t *hessian_ids,             \
                      const uint8_t *active_state, const T *pmat, const T *dchi2, const T *x,       \
                      T *y) {                                                                       \
    Slot<T> s{d, jac, ids, N, I, hessian_ids, active_state};                                        \
    JtPv_kernel<T>(na, active_ids, E, s, pmat, dchi2, x, y);                                        \
  }                                                                                                 \
  void gro_hessian_block_##SFX(size_t na, const size_t *active_ids, int E, int di, const T *jac_i,  \
                               size_t Ii, const size_t *hid_i, const uint8_t *act_i, int dj,        \
                               const T *jac_j, size_t Ij, const size_t *hid_j,                      \
                               const uint8_t *act_j, const size_t *ids, size_t N,                   \
                               const size_t *block_offsets, const T *pmat, const T *dchi2,          \
                               T *hessian) {                                                        \
    Slot<T> si{di, jac_i, ids, N, Ii, hid_i, act_i};                                                \
    Slot<T> sj{dj, jac_j, ids, N, Ij, hid_j, act_j};                                                \
    hessian_block_kernel<T>(na, active_ids, E, si, sj, block_offsets, pmat, dchi2, hessian);        \
  }                                                                                                 \
  void gro_augment_block_diagonal_##SFX(size_t nv, int D, T *blocks, const T *scalar_diag, T mu,    \
                                        int use_identity, const uint8_t *active_state) {            \
    augment_block_diagonal_kernel<T>(nv, D, blocks, scalar_diag, mu, use_identity != 0,             \
                                     active_state);                                                 \
  }                                                                                                 \
  void gro_apply_block_jacobi_##SFX(size_t nv, int D, T *z, const T *r, const T *blocks,            \
                                    const size_t *hessian_ids, const uint8_t *active_state) {       \
    apply_block_jacobi_kernel<T>(nv, D, z, r, blocks, hessian_ids, active_state);                   \
  }                                                                                                 \
  void gro_apply_update_##SFX(size_t nv, int D, T *params, const T *delta_x, const T *scales,       \
                              const size_t *hessian_ids, const uint8_t *active_state) {             \
    apply_update_kernel<T>(nv, D, params, delta_x, scales, hessian_ids, active_state);              \
  }                                                                                                 \
  /* ---- BAL pipeline handle ------------------------------------------------------------- */      \
  void *gro_bal_create_##SFX(size_t nc, size_t np, size_t no, const T *cams, const T *pts,          \
                             const T *obs, const int32_t *cam_idx, const int32_t *pt_idx) {         \
    auto *o = new BalOracle<T>();                                                                   \
    o->init(nc, np, no, cams, pts, obs, cam_idx, pt_idx);                                           \
    return o;                                                                                       \
  }                                                                                                 \
  void gro_bal_destroy_##SFX(void *h) { delete static_cast<BalOracle<T> *>(h); }                    \
  void gro_bal_set_loss_##SFX(void *h, int kind, T delta) {                                         \
    auto *o = static_cast<BalOracle<T> *>(h);                                                       \
    o->loss_kind = kind; o->loss_delta = delta;                                                     \
  }                                                                                                 \
  /* per-factor precision matrices [No][4] row-major, loss kinds / deltas [No], constraint data [No][4]; NULL = none */ \
  void gro_bal_set_factor_tables_##SFX(void *h, const T *pmat, const int *loss_kinds, const T *loss_deltas, const T *fdata) { \
    auto *o = static_cast<BalOracle<T> *>(h);                                                       \
    o->pmat.clear(); o->loss_kinds.clear(); o->loss_deltas.clear(); o->fdata.clear();               \
    if (pmat) o->pmat.assign(pmat, pmat + 4 * o->No);                                               \
    if (loss_kinds) { o->loss_kinds.assign(loss_kinds, loss_kinds + o->No); o->loss_deltas.assign(loss_deltas, loss_deltas + o->No); } \
    if (fdata) o->fdata.assign(fdata, fdata + 4 * o->No);                                           \
  }                                                                                                 \
  void gro_bal_set_model_##SFX(void *h, int kind) { static_cast<BalOracle<T> *>(h)->model_kind = kind; } \
  void gro_bal_set_jacobian_storage_##SFX(void *h, int mode) { static_cast<BalOracle<T> *>(h)->jac_storage = mode; } \
  void gro_bal_set_scale_system_##SFX(void *h, int on) { static_cast<BalOracle<T> *>(h)->scale_system = on != 0; } \
  void gro_bal_set_fixed_##SFX(void *h, const uint8_t *cf, const uint8_t *pf) { static_cast<BalOracle<T> *>(h)->set_fixed(cf, pf); } \
  void gro_bal_set_pcg_single_reduction_##SFX(void *h, int on) { static_cast<BalOracle<T> *>(h)->pcg_single_reduction = on != 0; } \
  void gro_bal_set_params_##SFX(void *h, const T *cams, const T *pts) {                             \
    auto *o = static_cast<BalOracle<T> *>(h);                                                       \
    o->cams.assign(cams, cams + 9 * o->Nc); o->pts.assign(pts, pts + 3 * o->Np);                    \
  }                                                                                                 \
  void gro_bal_get_params_##SFX(void *h, T *cams, T *pts) {                                         \
    auto *o = static_cast<BalOracle<T> *>(h);                                                       \
    std::copy(o->cams.begin(), o->cams.end(), cams); std::copy(o->pts.begin(), o->pts.end(), pts);  \
  }                                                                                                 \
  void gro_bal_compute_error_##SFX(void *h) { static_cast<BalOracle<T> *>(h)->compute_error(); }    \
  double gro_bal_chi2_##SFX(void *h) { return (double)static_cast<BalOracle<T> *>(h)->chi2(); }     \
  void gro_bal_linearize_##SFX(void *h) { static_cast<BalOracle<T> *>(h)->linearize(); }            \
  void gro_bal_hessian_update_##SFX(void *h) { static_cast<BalOracle<T> *>(h)->hessian_update_values(); } \
  void gro_bal_apply_damping_##SFX(void *h, T mu, int use_identity) {                               \
    static_cast<BalOracle<T> *>(h)->apply_damping(mu, use_identity != 0);                           \
  }                                                                                                 \
  void gro_bal_schur_update_##SFX(void *h) { static_cast<BalOracle<T> *>(h)->schur_update_values(); } \
  size_t gro_bal_nnzb_schur_##SFX(void *h) { return static_cast<BalOracle<T> *>(h)->nnzb_S(); }     \
  /* which: 0 res(2No) 1 Jc(18No) 2 Jp(6No) 3 scales(n) 4 b(n) 5 Hcc(81Nc) 6 Hcp(27No) 7 Hll(9Np)   \
     8 S(81 nnzb) 9 b_schur(9Nc) 10 Hll_inv(9Np) 11 chi2_vec(No) 12 dchi2(No) 13 prev_diag(n) */     \
  size_t gro_bal_get_##SFX(void *h, int which, T *out) {                                            \
    auto *o = static_cast<BalOracle<T> *>(h);                                                       \
    const std::vector<T> *v = nullptr;                                                              \
    switch (which) {                                                                                \
    case 0: v = &o->res; break; case 1: v = &o->Jc; break; case 2: v = &o->Jp; break;               \
    case 3: v = &o->scales; break; case 4: v = &o->b; break; case 5: v = &o->Hcc; break;            \
    case 6: v = &o->Hcp; break; case 7: v = &o->Hll; break; case 8: v = &o->S; break;               \
    case 9: v = &o->b_schur; break; case 10: v = &o->Hll_inv; break; case 11: v = &o->chi2_vec; break; \
    case 12: v = &o->dchi2; break; case 13: v = &o->prev_diag; break;                               \
    default: return 0;                                                                              \
    }                                                                                               \
    if (out) std::copy(v->begin(), v->end(), out);                                                  \
    return v->size();                                                                               \
  }                                                                                                 \
  void gro_bal_schur_structure_##SFX(void *h, int64_t *colptr, int64_t *rowidx) {                   \
    auto *o = static_cast<BalOracle<T> *>(h);                                                       \
    std::copy(o->S_colptr.begin(), o->S_colptr.end(), colptr);                                      \
    std::copy(o->S_row.begin(), o->S_row.end(), rowidx);                                            \
  }                                                                                                 \
  void gro_bal_schur_matvec_##SFX(void *h, const T *x, T *y) { static_cast<BalOracle<T> *>(h)->schur_matvec(x, y); } \
  void gro_bal_landmark_update_##SFX(void *h, const T *xp, T *xl) { static_cast<BalOracle<T> *>(h)->landmark_update(xp, xl); } \
  /* reference-layout export: call with null outputs to size, then again to fill */                 \
  size_t gro_bal_export_hessian_##SFX(void *h, T *values, int64_t *colptr, int64_t *rowidx,         \
                                      int64_t *offsets, size_t *nblocks) {                          \
    auto *o = static_cast<BalOracle<T> *>(h);                                                       \
    std::vector<T> v; std::vector<int64_t> cp, ri, of;                                              \
    o->export_hessian(v, cp, ri, of);                                                               \
    if (nblocks) *nblocks = ri.size();                                                              \
    if (values) std::copy(v.begin(), v.end(), values);                                              \
    if (colptr) std::copy(cp.begin(), cp.end(), colptr);                                            \
    if (rowidx) std::copy(ri.begin(), ri.end(), rowidx);                                            \
    if (offsets) std::copy(of.begin(), of.end(), offsets);                                          \
    return v.size();                                                                                \
  }                                                                                                 \
  /* scalar upper CSC of H (which=0) or S (which=1) */                                              \
  size_t gro_bal_export_csc_##SFX(void *h, int which, int64_t *p, int64_t *i, T *x) {               \
    auto *o = static_cast<BalOracle<T> *>(h);                                                       \
    std::vector<int64_t> pp, ii; std::vector<T> xx;                                                 \
    if (which == 0) o->export_hessian_csc(pp, ii, xx); else o->export_schur_csc(pp, ii, xx);        \
    if (p) std::copy(pp.begin(), pp.end(), p);                                                      \
    if (i) std::copy(ii.begin(), ii.end(), i);                                                      \
    if (x) std::copy(xx.begin(), xx.end(), x);                                                      \
    return xx.size();                                                                               \
  }                                                                                                 \
  void gro_bal_solver_update_values_##SFX(void *h, int kind) { static_cast<BalOracle<T> *>(h)->solver_update_values(kind); } \
  void gro_bal_solver_set_damping_##SFX(void *h, int kind, T mu, int use_identity) {                \
    static_cast<BalOracle<T> *>(h)->solver_set_damping(kind, mu, use_identity != 0);                \
  }                                                                                                 \
  int gro_bal_solver_solve_##SFX(void *h, int kind, int max_iter, double tol, double rej, T *x) {   \
    LMOptions opt; opt.solver = kind; opt.pcg_max_iter = max_iter; opt.pcg_tol = tol;               \
    opt.pcg_rejection_ratio = rej;                                                                  \
    auto *o = static_cast<BalOracle<T> *>(h);                                                       \
    const bool ok = o->solver_solve(opt, x);                                                        \
    return ok ? o->last_pcg_iters : -1;                                                             \
  }                                                                                                 \
  void gro_bal_apply_update_##SFX(void *h, const T *dx) { static_cast<BalOracle<T> *>(h)->apply_update(dx); } \
  void gro_bal_backup_##SFX(void *h) { static_cast<BalOracle<T> *>(h)->backup(); }                  \
  void gro_bal_revert_##SFX(void *h) { static_cast<BalOracle<T> *>(h)->revert(); }                  \
  /* stats: [iterations_run, accepted, pcg_iterations, solve_s, loop_s, setup_s] */                 \
  int gro_bal_lm_##SFX(void *h, int solver, int iterations, double initial_damping,                 \
                       int use_identity, int pcg_max_iter, double pcg_tol, double pcg_rej,          \
                       double *chi2_trace, double *lambda_trace, double *stats, int early_stop) {   \
    LMOptions opt; opt.solver = solver; opt.iterations = iterations;                                \
    opt.initial_damping = initial_damping; opt.use_identity = use_identity;                         \
    opt.pcg_max_iter = pcg_max_iter; opt.pcg_tol = pcg_tol; opt.pcg_rejection_ratio = pcg_rej;      \
    opt.early_stop = early_stop;                                                                    \
    std::vector<double> ct, lt; LMStats st;                                                         \
    const bool run = static_cast<BalOracle<T> *>(h)->levenberg_marquardt(opt, ct, lt, st);          \
    for (size_t k = 0; k < ct.size(); ++k) { chi2_trace[k] = ct[k]; lambda_trace[k] = lt[k]; }      \
    for (size_t k = ct.size(); k < (size_t)iterations + 1; ++k) { chi2_trace[k] = std::nan(""); lambda_trace[k] = std::nan(""); } \
    stats[0] = st.iterations_run; stats[1] = st.accepted; stats[2] = st.pcg_iterations;             \
    stats[3] = st.solve_seconds; stats[4] = st.loop_seconds; stats[5] = st.setup_seconds;           \
    return run ? 1 : 0;                                                                             \
  }
GRO_FOR_T(X)
#undef X

/* ---- timed CPU baseline (cpu_baseline.hpp): OpenMP assembly + single-thread LDL^T, or all-cores PCG ---- */
#define X(T, SFX)                                                                                   \
  void *gro_baseline_create_##SFX(size_t nc, size_t np, size_t no, const T *c, const T *p, const T *o, \
                                  const int32_t *ci, const int32_t *pi) {                           \
    auto *b = new CpuBaseline<T>();                                                                 \
    b->init(nc, np, no, c, p, o, ci, pi);                                                           \
    return b;                                                                                       \
  }                                                                                                 \
  void gro_baseline_destroy_##SFX(void *h) { delete static_cast<CpuBaseline<T> *>(h); }             \
  /* stats: as gro_bal_lm; times[14]: linearize hessian schur export_csc ldlt_analyze ldlt_factor ldlt_solve backsub pcg update_chi2 loop setup ldlt_nnz threads */ \
  int gro_baseline_lm_##SFX(void *h, int solver, int iterations, double initial_damping, int pcg_max_iter, \
                            double pcg_tol, double pcg_rej, int threads, int ordering, double *chi2_trace,  \
                            double *lambda_trace, double *stats, double *times) {                   \
    auto *b = static_cast<CpuBaseline<T> *>(h);                                                     \
    if (threads <= 0) threads = omp_get_num_procs();                                                \
    b->prepare(threads, ordering);                                                                  \
    LMOptions opt; opt.solver = solver; opt.iterations = iterations; opt.initial_damping = initial_damping; \
    opt.pcg_max_iter = pcg_max_iter; opt.pcg_tol = pcg_tol; opt.pcg_rejection_ratio = pcg_rej;      \
    std::vector<double> ct, lt; LMStats st;                                                         \
    const bool run = b->levenberg_marquardt_mt(opt, ct, lt, st);                                    \
    for (size_t k = 0; k < ct.size(); ++k) { chi2_trace[k] = ct[k]; lambda_trace[k] = lt[k]; }      \
    stats[0] = st.iterations_run; stats[1] = st.accepted; stats[2] = st.pcg_iterations;             \
    stats[3] = st.solve_seconds; stats[4] = st.loop_seconds; stats[5] = st.setup_seconds;           \
    const BaselineTimes &m = b->tm;                                                                 \
    const double tv[14] = {m.linearize, m.hessian, m.schur, m.export_csc, m.ldlt_analyze, m.ldlt_factor, m.ldlt_solve, \
                           m.backsub, m.pcg, m.update_chi2, m.loop, m.setup, (double)m.ldlt_nnz, (double)threads}; \
    for (int k = 0; k < 14; ++k) times[k] = tv[k];                                                  \
    return run ? 1 : 0;                                                                             \
  }
GRO_FOR_T(X)
#undef X

// amd.hpp: perm[new] = old of the symmetric matrix whose UPPER triangle is given in scalar CSC (tests)
void gro_amd_order(int64_t n, const int64_t *Ap, const int64_t *Ai, int64_t *perm) {
  const std::vector<int64_t> p = amd_order(n, Ap, Ai);
  for (int64_t i = 0; i < n; ++i) perm[i] = i < (int64_t)p.size() ? p[i] : -1;
}
// nnz(L) (strictly lower part) of the simplicial LDL^T under a given order (perm may be NULL: natural order)
int64_t gro_ldlt_fill(int64_t n, const int64_t *Ap, const int64_t *Ai, const int64_t *perm) {
  SparseLDLT l;
  std::vector<int64_t> pv;
  if (perm) pv.assign(perm, perm + n);
  l.analyze(n, Ap, Ai, pv);
  return (int64_t)l.Li.size();
}

// BASELINE configs[0] (circle_fit.hpp): pts [n][2] in / out; fixed, factor_on: n bytes each; solver 0 EigenLDLTSolver, 1 PCGSolver +
// IdentityPreconditioner; traces of iterations + 1 doubles; stats: accepted, inner PCG iterations.  Returns the iterations run.
int gro_circle_lm_f64(size_t n, double R, double *pts, const unsigned char *fixed, const unsigned char *factor_on, int solver, int iterations,
                      double initial_damping, int use_identity, int pcg_max_iter, double pcg_tol, double pcg_rej, double *chi2_trace,
                      double *lambda_trace, int *stats) {
  CircleOracle o(n, R, pts);
  for (size_t i = 0; i < n; ++i) { o.fixed[i] = fixed ? fixed[i] : 0; o.factor_on[i] = factor_on ? factor_on[i] : 1; }
  const int it = o.levenberg_marquardt(solver, iterations, initial_damping, use_identity != 0, pcg_max_iter, pcg_tol, pcg_rej, chi2_trace, lambda_trace, &stats[0], &stats[1]);
  for (size_t i = 0; i < 2 * n; ++i) pts[i] = o.p[i];
  return it;
}

} // extern "C"
