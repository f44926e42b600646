"""BASELINE.json's headline configurations, at FULL size, against the CPU oracle.

The oracle's LM on Ladybug-1723 fp64 costs 0.6 s per block-Jacobi-PCG iteration, 2.3 s per PCG-Schur
iteration and 6 s per LDL^T-Schur iteration of host time; Venice-1778 fp32 PCG 5 s per iteration — cheap
enough to pin the chi2 / lambda traces, the inner iteration counts and the final parameters of the
configurations bench.py measures, not only mini-50 / Ladybug-49.

Bars: north star 1e-6 relative on the residual (chi2) trace; fp64 holds 1e-8 and the tests assert that.
fp32: 1e-4 relative on chi2 (SURVEY 8(d)'s starting tolerance), stated per test.
Follows optimizer/levenberg_marquardt.hpp:166-240, solver/pcg.hpp:61-232, solver/pcg_schur.hpp:79-168,
solver/eigen_schur.hpp:71-108.
"""
import numpy as np
import pytest

import graphite_amd as ga
from graphite_amd import synth

pytestmark = pytest.mark.gpu


def rel_trace(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b) / np.abs(b)))


@pytest.fixture(scope="module")
def ladybug1723():
    return synth.make_config("ladybug-1723")


@pytest.fixture(scope="module")
def venice1778():
    return synth.make_config("venice-1778")


def run_pair(oracle_mod, prob, dtype, gsolver, osolver, iterations, jac32=False, **kw):
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    if jac32:
        gpu.set_jacobian_precision(np.float32)
    g = gpu.levenberg_marquardt(solver=gsolver, iterations=iterations, **kw)
    r = ref.levenberg_marquardt(solver=osolver, iterations=iterations, **kw)
    cg, pg = gpu.get_params()
    cr, pr = ref.get_params()
    gpu.close()
    return g, r, (cg, pg), (cr, pr)


def test_ladybug1723_fp64_block_jacobi_pcg_lm_trace(oracle_mod, ladybug1723):
    """bench.py's default line: 8 LM iterations, bal.cu defaults (10 inner iterations, tol 1, rejection 5)."""
    (ct, lt, st), (ct_r, lt_r, st_r), (cg, pg), (cr, pr) = run_pair(
        oracle_mod, ladybug1723, np.float64, ga.SOLVER_PCG, oracle_mod.SOLVER_PCG, 8)
    assert st["iterations_run"] == st_r["iterations_run"] == 8
    assert st["accepted"] == st_r["accepted"]
    assert st["pcg_iterations"] == st_r["pcg_iterations"]
    assert rel_trace(ct, ct_r) < 1e-8      # north star: 1e-6
    assert rel_trace(lt, lt_r) < 1e-6      # lambda follows rho^3: amplifies the chi2 rounding
    assert np.abs(cg - cr).max() / np.abs(cr).max() < 1e-8
    # weakly observed points travel thousands of units in 8 iterations and carry the PCG rounding with them
    assert np.abs(pg - pr).max() / np.abs(pr).max() < 1e-6


def test_ladybug1723_fp64_pcg_fixed_inner_iterations(oracle_mod, ladybug1723):
    """the bench's second line: tol 0, so every solve runs its 10 inner iterations (unless rz hits the
    rejection ratio): pins the PCG recurrence itself at full size."""
    kw = dict(pcg_tol=0.0, pcg_max_iter=10)
    (ct, lt, st), (ct_r, lt_r, st_r), (cg, _), (cr, _) = run_pair(
        oracle_mod, ladybug1723, np.float64, ga.SOLVER_PCG, oracle_mod.SOLVER_PCG, 4, **kw)
    assert st["pcg_iterations"] == st_r["pcg_iterations"]
    assert st["accepted"] == st_r["accepted"]
    assert rel_trace(ct, ct_r) < 1e-8
    assert np.abs(cg - cr).max() / np.abs(cr).max() < 1e-7


@pytest.mark.parametrize("solver", ["pcg_schur", "pcg_schur_implicit"])
def test_ladybug1723_fp64_schur_pcg_lm_trace(oracle_mod, ladybug1723, solver):
    gs = dict(pcg_schur=ga.SOLVER_PCG_SCHUR, pcg_schur_implicit=ga.SOLVER_PCG_SCHUR_IMPLICIT)[solver]
    (ct, lt, st), (ct_r, lt_r, st_r), (cg, pg), (cr, pr) = run_pair(
        oracle_mod, ladybug1723, np.float64, gs, oracle_mod.SOLVER_PCG_SCHUR, 3)
    assert st["pcg_iterations"] == st_r["pcg_iterations"]
    assert st["accepted"] == st_r["accepted"]
    assert rel_trace(ct, ct_r) < 1e-8
    assert rel_trace(lt, lt_r) < 1e-6
    assert np.abs(pg - pr).max() / np.abs(pr).max() < 1e-8


def test_ladybug1723_fp64_direct_schur_lm_trace(oracle_mod, ladybug1723):
    """dense MFMA Cholesky of S against the oracle's simplicial LDL^T of S (eigen_schur.hpp:71-108)."""
    (ct, lt, st), (ct_r, lt_r, st_r), (cg, pg), (cr, pr) = run_pair(
        oracle_mod, ladybug1723, np.float64, ga.SOLVER_DENSE_SCHUR, oracle_mod.SOLVER_LDLT_SCHUR, 2)
    assert st["accepted"] == st_r["accepted"]
    assert rel_trace(ct, ct_r) < 1e-8
    assert np.abs(cg - cr).max() / np.abs(cr).max() < 1e-7
    assert np.abs(pg - pr).max() / np.abs(pr).max() < 1e-7


def test_venice1778_fp32_block_jacobi_pcg_lm_trace(oracle_mod, venice1778):
    """configs[3] on one GPU, fp32 throughout.  chi2 is a sum of 5 M terms of ~1e1 magnitude taken in a different
    order: 1e-4 relative; the inner iteration count is a discrete function of fp32 dots and is
    allowed to differ by one per solve."""
    (ct, lt, st), (ct_r, lt_r, st_r), (cg, pg), (cr, pr) = run_pair(
        oracle_mod, venice1778, np.float32, ga.SOLVER_PCG, oracle_mod.SOLVER_PCG, 3)
    assert st["accepted"] == st_r["accepted"]
    assert abs(st["pcg_iterations"] - st_r["pcg_iterations"]) <= 1
    assert rel_trace(ct, ct_r) < 2e-5  # round 5: was 1e-4
    assert np.abs(pg - pr).max() / np.abs(pr).max() < 1e-3


def test_ladybug1723_mixed_precision_lm_trace(oracle_mod, ladybug1723):
    """configs[4]'s arithmetic (fp32 Jacobians, fp64 residuals / sums / PCG) at Ladybug-1723 size against the
    all-fp64 oracle: the Jacobian rounding (1e-7 relative) moves the step, not the residual evaluation, so the
    chi2 trace agrees to ~1e-5 after the first iterations."""
    (ct, lt, st), (ct_r, lt_r, st_r), _, _ = run_pair(
        oracle_mod, ladybug1723, np.float64, ga.SOLVER_PCG, oracle_mod.SOLVER_PCG, 4, jac32=True)
    assert st["accepted"] == st_r["accepted"]
    assert ct[0] == pytest.approx(ct_r[0], rel=1e-12)  # residuals are evaluated in fp64
    assert rel_trace(ct, ct_r) < 1e-4


def test_final13682_fp64_block_jacobi_pcg_against_oracle(oracle_mod):
    """configs[4]'s graph at FULL size (13 682 cameras, 4.7 M points, 29 M observations) on one GPU, fp64: the largest
    configuration runs the point-tiled order (run-length-capped tile count), observation-ordered operator output and the
    direction-kernel PCG form — four LM iterations against the oracle (≈2 min of host time), then the fp32-Jacobian variant
    of the config on the same engine."""
    prob = synth.make_config("final-13682")
    (ct, lt, st), (ct_r, lt_r, st_r), (cg, pg), (cr, pr) = run_pair(
        oracle_mod, prob, np.float64, ga.SOLVER_PCG, oracle_mod.SOLVER_PCG, 4)
    assert st["accepted"] == st_r["accepted"]
    assert st["pcg_iterations"] == st_r["pcg_iterations"]
    assert rel_trace(ct, ct_r) < 1e-8
    assert np.abs(cg - cr).max() / np.abs(cr).max() < 1e-7
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    gpu.set_jacobian_precision(np.float32)
    ct32, _, st32 = gpu.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=4)
    gpu.close()
    assert ct32[0] == pytest.approx(ct_r[0], rel=1e-12)      # residuals in fp64
    assert rel_trace(ct32, ct_r) < 1e-4                      # fp32 Jacobians move the step at 1e-7 relative
