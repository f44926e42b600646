"""Properties at BASELINE.json's full sizes (the oracle needs minutes there, so these are size-independent
identities of the path itself) and the edge cases of the domain at small size.

  * S is applied consistently by its three implementations: explicit block product, implicit two-pass
    operator, dense Cholesky (PCG on S run to convergence equals the direct solve: the reference's own
    bar of 5e-4, tests/schur.cu:340-389);
  * S x is linear and symmetric; explicit- and implicit-Schur PCG produce the same iterates;
  * the full-system PCG run to convergence agrees with the Schur direct solve (tests/schur.cu:242-289);
  * apply_update / revert is exact; LM decreases chi2; a 4-way landmark-sharded run reproduces the
    single-GPU run.
"""
import threading

import numpy as np
import pytest

import graphite_amd as ga
from graphite_amd import dist as gdist, synth

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def ladybug1723():
    return synth.make_config("ladybug-1723")


@pytest.fixture(scope="module")
def venice1778():
    return synth.make_config("venice-1778")


def prepared(prob, dtype, solver, mu=1e-4):
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    g.solver_update_structure(solver)
    g.linearize()
    g.solver_update_values(solver)
    g.solver_set_damping(solver, mu)
    return g


def test_schur_matvec_linear_and_symmetric_ladybug1723(ladybug1723):
    g = prepared(ladybug1723, np.float64, ga.SOLVER_PCG_SCHUR)
    g.schur_update_values()
    rng = np.random.default_rng(0)
    m = 9 * g.Nc
    x, y = rng.standard_normal(m), rng.standard_normal(m)
    Sx, Sy = g.schur_matvec(x), g.schur_matvec(y)
    assert relerr(g.schur_matvec(2.5 * x - 0.75 * y), 2.5 * Sx - 0.75 * Sy) < 1e-12
    assert abs(y @ Sx - x @ Sy) / abs(y @ Sx) < 1e-11
    assert x @ Sx > 0  # damped S is positive definite
    g.close()


def test_explicit_and_implicit_schur_pcg_same_iterates_ladybug1723(ladybug1723):
    ge = prepared(ladybug1723, np.float64, ga.SOLVER_PCG_SCHUR)
    gi = prepared(ladybug1723, np.float64, ga.SOLVER_PCG_SCHUR_IMPLICIT)
    for max_iter, tol in ((5, 0.0), (40, 1e-10)):
        dxe, ite = ge.solver_solve(ga.SOLVER_PCG_SCHUR, max_iter=max_iter, tol=tol, rej=1e6)
        dxi, iti = gi.solver_solve(ga.SOLVER_PCG_SCHUR_IMPLICIT, max_iter=max_iter, tol=tol, rej=1e6)
        assert ite == iti
        assert relerr(dxi, dxe) < 1e-7
    ge.close(); gi.close()


def test_pcg_on_S_converges_to_the_dense_cholesky_solution_ladybug1723(ladybug1723):
    """tests/schur.cu:340-389 at full size: PCG-Schur (512 it, tol 1e-14, rejection 1e6) vs direct, < 5e-4."""
    gd = prepared(ladybug1723, np.float64, ga.SOLVER_DENSE_SCHUR)
    gp = prepared(ladybug1723, np.float64, ga.SOLVER_PCG_SCHUR_IMPLICIT)
    dxd, _ = gd.solver_solve(ga.SOLVER_DENSE_SCHUR)
    dxp, it = gp.solver_solve(ga.SOLVER_PCG_SCHUR_IMPLICIT, max_iter=512, tol=1e-14, rej=1e6)
    assert np.abs(dxp - dxd).max() < 5e-4
    assert relerr(dxp, dxd) < 1e-5
    # the direct solution satisfies the reduced system: S x_p = b_S
    gd.schur_update_values()
    xp = dxd[:9 * gd.Nc]
    assert relerr(gd.schur_matvec(xp), gd.get("b_schur")) < 1e-9
    gd.close(); gp.close()


def test_full_system_pcg_agrees_with_schur_direct_ladybug1723(ladybug1723):
    """tests/schur.cu:242-289 (full vs Schur solve, 1e-8 with direct solvers); here the full system is
    solved by block-Jacobi PCG run to 300 iterations, so the bar is PCG convergence, 1e-3 relative."""
    gd = prepared(ladybug1723, np.float64, ga.SOLVER_DENSE_SCHUR)
    gp = prepared(ladybug1723, np.float64, ga.SOLVER_PCG)
    dxd, _ = gd.solver_solve(ga.SOLVER_DENSE_SCHUR)
    dxp, it = gp.solver_solve(ga.SOLVER_PCG, max_iter=300, tol=1e-16, rej=1e9)
    assert relerr(dxp, dxd) < 1e-3
    gd.close(); gp.close()


def test_dense_schur_venice1778_fp32_solves_its_own_system(venice1778):
    """fully dense S (1778 cameras, 16 002 unknowns, fp32): S x = b_S to fp32 accuracy."""
    g = prepared(venice1778, np.float32, ga.SOLVER_DENSE_SCHUR)
    dx, _ = g.solver_solve(ga.SOLVER_DENSE_SCHUR)
    g.schur_update_values()
    xp = dx[:9 * g.Nc]
    r = g.schur_matvec(xp) - g.get("b_schur")
    assert np.abs(r).max() / np.abs(g.get("b_schur")).max() < 2e-3
    g.close()


def test_update_revert_exact_and_lm_monotone_ladybug1723(ladybug1723):
    g = ga.BalProblem(ladybug1723.cameras, ladybug1723.points, ladybug1723.obs, ladybug1723.cam_idx, ladybug1723.pt_idx,
                      dtype=np.float64)
    g.linearize()
    dx = np.random.default_rng(1).standard_normal(g.n) * 1e-3
    g.backup_parameters()
    g.apply_update(dx)
    c1, _ = g.get_params()
    assert not np.array_equal(c1, ladybug1723.cameras)
    g.revert_parameters()
    c, p = g.get_params()
    assert np.array_equal(c, ladybug1723.cameras) and np.array_equal(p, ladybug1723.points)
    chi0 = g.chi2()
    ct, lt, st = g.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=12)
    assert abs(ct[0] - chi0) / chi0 < 1e-12
    assert np.all(np.diff(ct) <= 1e-9 * ct[0])          # rejected steps repeat the value, accepted ones decrease it
    assert ct[-1] < 0.05 * ct[0]
    assert abs(g.chi2() - ct[-1]) / ct[-1] < 1e-12       # the kept vertices are the ones chi2 was quoted for
    g.close()


def test_sharded_4way_reproduces_single_gpu_ladybug1723(ladybug1723):
    single = ga.BalProblem(ladybug1723.cameras, ladybug1723.points, ladybug1723.obs, ladybug1723.cam_idx,
                           ladybug1723.pt_idx, dtype=np.float64)
    ct, lt, st = single.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=5)
    c1, p1 = single.get_params()
    single.close()
    world = 4
    shards = [gdist.partition_by_landmark(ladybug1723, r, world) for r in range(world)]
    eng = [ga.BalProblem(s.cameras, s.points, s.obs, s.cam_idx, s.pt_idx, dtype=np.float64, shard=True) for s in shards]
    gdist.init_local_group(eng)
    out, err = [None] * world, []

    def work(r):
        try:
            out[r] = eng[r].levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=5)
        except Exception as e:  # pragma: no cover
            err.append(e)

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not err and all(o is not None for o in out)
    for r in range(world):
        assert np.allclose(out[r][0], ct, rtol=1e-9)
        assert out[r][2]["pcg_iterations"] == st["pcg_iterations"]
    cams = [e.get_params()[0] for e in eng]
    pts = gdist.assemble_points(shards, [e.get_params()[1] for e in eng])
    assert all(np.array_equal(c, cams[0]) for c in cams)
    assert np.allclose(cams[0], c1, rtol=1e-7, atol=1e-10) and np.allclose(pts, p1, rtol=1e-7, atol=1e-10)
    [e.close() for e in eng]


@pytest.mark.parametrize("solver", [ga.SOLVER_PCG, ga.SOLVER_PCG_SCHUR_IMPLICIT])
def test_sharded_2way_venice1778_fp32(venice1778, solver):
    """BASELINE configs[3] at its full size, two landmark shards in one process: every rank sees the chi2 of the
    unsharded run (fp32: the sums are re-associated, so to fp32 accuracy) and the replicated cameras stay bit-identical."""
    single = ga.BalProblem(venice1778.cameras, venice1778.points, venice1778.obs, venice1778.cam_idx, venice1778.pt_idx, dtype=np.float32)
    ct, _, _ = single.levenberg_marquardt(solver=solver, iterations=3)
    single.close()
    world = 2
    shards = [gdist.partition_by_landmark(venice1778, r, world) for r in range(world)]
    assert sum(len(s.obs) for s in shards) == len(venice1778.obs)
    eng = [ga.BalProblem(s.cameras, s.points, s.obs, s.cam_idx, s.pt_idx, dtype=np.float32, shard=True) for s in shards]
    gdist.init_local_group(eng)
    out, err = [None] * world, []

    def work(r):
        try:
            out[r] = eng[r].levenberg_marquardt(solver=solver, iterations=3)
        except Exception as e:  # pragma: no cover
            err.append(e)

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not err and all(o is not None for o in out)
    m = min(len(ct), len(out[0][0]))
    assert m >= 3
    for r in range(world):
        assert np.allclose(out[r][0][:m], ct[:m], rtol=1e-4)
        assert np.array_equal(out[r][0], out[0][0])
    cams = [e.get_params()[0] for e in eng]
    assert all(np.array_equal(c, cams[0]) for c in cams)
    [e.close() for e in eng]


# ---- edge cases of the domain (small) -------------------------------------------------------------
def edge_problem(seed=7):
    """ragged degrees: one point seen by every camera, many seen by exactly two; one camera with a single
    observation; camera 0 with an exactly zero rotation vector (the reference's theta == 0 branch,
    projection_jacobians.cuh:175-212: zero rotation derivative)."""
    base = synth.make_problem(12, 60, 400, seed=seed, window=12, name="edge")
    cam_idx, pt_idx, obs = list(base.cam_idx), list(base.pt_idx), list(map(tuple, base.obs))
    have = set(zip(cam_idx, pt_idx))
    cams = base.cameras.copy()
    cams[0, 0:3] = 0.0
    pts = base.points.copy()
    for c in range(12):  # point 0 seen by every camera
        if (c, 0) not in have:
            cam_idx.append(c); pt_idx.append(0); obs.append((0.0, 0.0)); have.add((c, 0))
    # camera 11 keeps a single observation
    keep = [i for i, (c, l) in enumerate(zip(cam_idx, pt_idx)) if c != 11 or l == 0]
    cam_idx = np.array([cam_idx[i] for i in keep], np.int32)
    pt_idx = np.array([pt_idx[i] for i in keep], np.int32)
    obs = np.array([obs[i] for i in keep], np.float64)
    obs = synth.project(cams, pts, cam_idx, pt_idx) + np.random.default_rng(seed).normal(0, 0.5, obs.shape)
    # points that lost all observations are dropped (the library refuses unobserved vertices)
    used = np.unique(pt_idx)
    remap = -np.ones(60, np.int64); remap[used] = np.arange(len(used))
    return cams, pts[used], obs, cam_idx, remap[pt_idx].astype(np.int32)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 5e-3)])
def test_edge_cases_against_oracle(oracle_mod, dtype, tol):
    cams, pts, obs, ci, pi = edge_problem()
    gpu = ga.BalProblem(cams, pts, obs, ci, pi, dtype=dtype)
    ref = oracle_mod.BalOracle(cams, pts, obs, ci, pi, dtype=dtype)
    gpu.solver_update_structure(ga.SOLVER_PCG_SCHUR)
    gpu.linearize(); ref.linearize(); ref.hessian_update()
    Hcc = gpu.get("Hcc").reshape(-1, 9, 9)
    assert np.all(Hcc[0][:3, :] == 0) and np.all(Hcc[0][:, :3] == 0)  # theta == 0: zero rotation block, as the reference
    for k in ("b", "Hcc", "Hll", "Hcp"):
        # scales of the zero columns are 1/eps; compare the unscaled-insensitive quantities entry-wise
        a, b = gpu.get(k), ref.get(k)
        assert np.allclose(a, b, rtol=tol, atol=tol * np.abs(b).max()), k
    for gs, os_ in ((ga.SOLVER_PCG, oracle_mod.SOLVER_PCG), (ga.SOLVER_PCG_SCHUR, oracle_mod.SOLVER_PCG_SCHUR),
                    (ga.SOLVER_PCG_SCHUR_IMPLICIT, oracle_mod.SOLVER_PCG_SCHUR)):
        gpu.set_params(cams, pts); ref.set_params(cams, pts)
        ct_g, _, _ = gpu.levenberg_marquardt(solver=gs, iterations=6)
        ct_r, _, _ = ref.levenberg_marquardt(solver=os_, iterations=6)
        assert len(ct_g) == len(ct_r)
        assert np.abs(ct_g - ct_r).max() / ct_r.max() < max(tol, 1e-8) * 10
    gpu.close()


def test_limits_are_refused_not_overflowed():
    prob = synth.make_config("mini-6")
    lib = ga._lib.lib()
    import ctypes as C
    h = C.c_void_p()
    # more observations than 32-bit block indexing allows: refused from the count alone, nothing is read
    st = lib.gr_bal_create(C.byref(h), C.c_int(1), C.c_int64(6), C.c_int64(40), C.c_int64(2 ** 31 // 27 + 1),
                           prob.cameras.ctypes.data_as(C.c_void_p), prob.points.ctypes.data_as(C.c_void_p),
                           prob.obs.ctypes.data_as(C.c_void_p), prob.cam_idx.ctypes.data_as(C.c_void_p),
                           prob.pt_idx.ctypes.data_as(C.c_void_p), C.c_int(0), None)
    assert st == 1  # GR_ERR_INVALID
    # an index outside [0, Nc) is refused
    ci = prob.cam_idx.copy(); ci[3] = 6
    with pytest.raises(ga._lib.GraphiteError):
        ga.BalProblem(prob.cameras, prob.points, prob.obs, ci, prob.pt_idx)
    # a point without observations is refused
    pts = np.vstack([prob.points, [[0.1, 0.2, 0.3]]])
    with pytest.raises(ga._lib.GraphiteError):
        ga.BalProblem(prob.cameras, pts, prob.obs, prob.cam_idx, prob.pt_idx)


def test_final13682_pcg_and_mixed_precision():
    """BASELINE configs[4] shape (13 682 cameras, 4.46 M points, 29 M observations), fp64 PCG and the
    fp32-Jacobian mode: chi2 decreases, both modes agree to fp32-Jacobian rounding, implicit Schur works."""
    prob = synth.make_config("final-13682")
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    chi0 = g.chi2()
    ct, _, st = g.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=3)
    assert abs(ct[0] - chi0) / chi0 < 1e-12 and ct[-1] < 0.5 * ct[0] and np.all(np.diff(ct) <= 0)
    g.set_params(prob.cameras, prob.points)
    g.set_jacobian_precision(np.float32)
    ctm, _, stm = g.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=3)
    assert np.abs(ctm - ct).max() / ct.max() < 1e-6 and stm["pcg_iterations"] == st["pcg_iterations"]
    g.set_jacobian_precision(np.float64)
    g.set_params(prob.cameras, prob.points)
    cti, _, _ = g.levenberg_marquardt(solver=ga.SOLVER_PCG_SCHUR_IMPLICIT, iterations=3)
    assert cti[-1] < 0.5 * cti[0]
    g.close()
