"""Randomised shape sweep: small BAL graphs whose sizes sit on the boundaries of the kernels' tilings
(64-lane waves, 256-observation tiles, (wave, camera) segments, 252-scalar camera tiles, points seen by
more cameras than a wave has lanes, a single camera ...), every stage against the oracle.  fp64 bars as in
test_gpu_parity.py."""
import numpy as np
import pytest

import graphite_amd as ga
from graphite_amd import synth

pytestmark = pytest.mark.gpu

# (Nc, Np, No, window, seed)
SHAPES = [
    (2, 4, 8, 2, 1),          # the smallest graph the generator can make
    (3, 30, 63, 3, 2),        # one short wave
    (3, 30, 64, 3, 3),        # exactly one wave
    (4, 25, 65, 4, 4),        # one lane into the second wave
    (5, 60, 255, 5, 5),
    (5, 60, 256, 5, 6),       # exactly one tile
    (5, 60, 257, 5, 7),
    (28, 300, 1023, 8, 8),    # 28 cameras = one 252-scalar camera tile
    (29, 300, 1025, 8, 9),    # one camera into the second camera tile
    (70, 40, 2000, 70, 10),   # points seen by up to 70 cameras (> 64 lanes), ~29 observations per camera
    (100, 30, 2900, 100, 11), # nearly complete camera x point graph
    (6, 900, 2400, 6, 12),    # 400 observations per camera: several tiles per camera
    (64, 5000, 12000, 4, 13), # narrow window: each camera sees few points, degree ~2.4
    (130, 700, 9000, 130, 14),
]


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s[:3])))
def test_every_stage_matches_the_oracle(oracle_mod, shape):
    Nc, Np, No, window, seed = shape
    prob = synth.make_problem(Nc, Np, No, seed=seed, window=window)
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    mu = 1e-4
    gpu.solver_update_structure(ga.SOLVER_PCG_SCHUR)
    gpu.linearize(); ref.linearize(); ref.hessian_update()
    assert abs(gpu.chi2() - ref.chi2()) / ref.chi2() < 1e-12
    for k in ("residuals", "scales", "b", "Hcc", "Hll", "Hcp"):
        assert relerr(gpu.get(k), ref.get("res" if k == "residuals" else k)) < 1e-9, k
    gpu.solver_update_values(ga.SOLVER_PCG_SCHUR); gpu.solver_set_damping(ga.SOLVER_PCG_SCHUR, mu); gpu.schur_update_values()
    ref.apply_damping(mu); ref.schur_update()
    cp_g, ri_g = gpu.schur_structure(); cp_r, ri_r = ref.schur_structure()
    assert np.array_equal(cp_g, cp_r) and np.array_equal(ri_g, ri_r)
    assert relerr(gpu.get("S"), ref.get("S")) < 1e-9 and relerr(gpu.get("b_schur"), ref.get("b_schur")) < 1e-9
    xp = np.linspace(-1, 1, 9 * Nc)
    assert relerr(gpu.schur_matvec(xp), ref.schur_matvec(xp)) < 1e-9
    assert relerr(gpu.landmark_update(xp), ref.landmark_update(xp)) < 1e-9
    for gs, os_ in ((ga.SOLVER_PCG, oracle_mod.SOLVER_PCG), (ga.SOLVER_PCG_IDENTITY, oracle_mod.SOLVER_PCG_IDENTITY),
                    (ga.SOLVER_PCG_SCHUR, oracle_mod.SOLVER_PCG_SCHUR), (ga.SOLVER_PCG_SCHUR_IMPLICIT, oracle_mod.SOLVER_PCG_SCHUR),
                    (ga.SOLVER_DENSE_SCHUR, oracle_mod.SOLVER_LDLT_SCHUR)):
        gpu.set_params(prob.cameras, prob.points); ref.set_params(prob.cameras, prob.points)
        gpu.solver_update_structure(gs); gpu.linearize(); gpu.solver_update_values(gs); gpu.solver_set_damping(gs, mu)
        ref.linearize(); ref.solver_update_values(os_); ref.solver_set_damping(os_, mu)
        dx_g, it_g = gpu.solver_solve(gs, max_iter=6, tol=0.0, rej=1e6)
        dx_r, it_r = ref.solver_solve(os_, max_iter=6, tol=0.0, rej=1e6)
        assert relerr(dx_g, dx_r) < 1e-6, (gs, relerr(dx_g, dx_r))
        if gs != ga.SOLVER_DENSE_SCHUR:
            assert it_g == it_r
        ct_g, lt_g, st_g = gpu.levenberg_marquardt(solver=gs, iterations=4)
        ct_r, lt_r, st_r = ref.levenberg_marquardt(solver=os_, iterations=4)
        assert len(ct_g) == len(ct_r) and np.allclose(ct_g, ct_r, rtol=1e-7, atol=1e-10 * ct_r[0]), gs
        assert np.allclose(lt_g, lt_r, rtol=1e-3)
    gpu.close()


@pytest.mark.parametrize("shape", [SHAPES[3], SHAPES[9], SHAPES[11]], ids=lambda s: "x".join(map(str, s[:3])))
def test_fp32_stages(oracle_mod, shape):
    Nc, Np, No, window, seed = shape
    prob = synth.make_problem(Nc, Np, No, seed=seed, window=window)
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
    gpu.solver_update_structure(ga.SOLVER_PCG_SCHUR)
    gpu.linearize(); ref.linearize(); ref.hessian_update()
    for k in ("b", "Hcc", "Hll", "Hcp"):
        assert relerr(gpu.get(k), ref.get(k)) < 5e-3, k
    ct_g, _, _ = gpu.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=3)
    ct_r, _, _ = ref.levenberg_marquardt(solver=oracle_mod.SOLVER_PCG, iterations=3)
    m = min(len(ct_g), len(ct_r))
    assert np.allclose(ct_g[:m], ct_r[:m], rtol=5e-3)
    gpu.close()


@pytest.mark.parametrize("shape", [SHAPES[7], SHAPES[9]], ids=lambda s: "x".join(map(str, s[:3])))
def test_huber_loss_all_solvers(oracle_mod, shape):
    """HuberLoss (loss.hpp:32-51) through every solver: the robust weights enter b, H, S and the operators."""
    Nc, Np, No, window, seed = shape
    prob = synth.make_problem(Nc, Np, No, seed=seed, window=window, noise_px=2.0)
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    gpu.set_loss(ga.LOSS_HUBER, 1.5)
    ref.set_loss(oracle_mod.LOSS_HUBER, 1.5)
    for gs, os_ in ((ga.SOLVER_PCG, oracle_mod.SOLVER_PCG), (ga.SOLVER_PCG_SCHUR, oracle_mod.SOLVER_PCG_SCHUR),
                    (ga.SOLVER_PCG_SCHUR_IMPLICIT, oracle_mod.SOLVER_PCG_SCHUR), (ga.SOLVER_DENSE_SCHUR, oracle_mod.SOLVER_LDLT_SCHUR)):
        gpu.set_params(prob.cameras, prob.points); ref.set_params(prob.cameras, prob.points)
        ct_g, lt_g, _ = gpu.levenberg_marquardt(solver=gs, iterations=5)
        ct_r, lt_r, _ = ref.levenberg_marquardt(solver=os_, iterations=5)
        assert len(ct_g) == len(ct_r) and np.allclose(ct_g, ct_r, rtol=1e-7), gs
        assert np.allclose(lt_g, lt_r, rtol=1e-3)
    gpu.close()
