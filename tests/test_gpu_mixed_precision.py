"""BASELINE.json configs[4] mode (fp32 Jacobian evaluation + fp64 PCG, the reference's Graph<double, float>,
examples/bal.cu --precision FP64-FP32) at parity-test size.  The reference pins no values for this mode
("parity unpinned"); the bar here: the mixed run stays within fp32-Jacobian rounding of the fp64 oracle
(chi2 trace 1e-5 relative, assembled blocks 1e-5, delta_x 1e-3) and is measurably not the fp64 run."""
import numpy as np
import pytest

import graphite_amd as ga
from graphite_amd import synth

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", ["mini-50", "ladybug-49"])
def test_mixed_assembly_close_to_fp64(oracle_mod, name):
    prob = synth.make_config(name)
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    gpu.set_jacobian_precision(np.float32)
    gpu.solver_update_structure(ga.SOLVER_PCG_SCHUR)
    gpu.linearize()
    ref.linearize()
    ref.hessian_update()
    # residuals / chi2 stay fp64
    assert abs(gpu.chi2() - ref.chi2()) / ref.chi2() < 1e-12
    assert relerr(gpu.get("residuals"), ref.get("res")) < 1e-12
    for key_g, key_r in (("b", "b"), ("Hcc", "Hcc"), ("Hll", "Hll"), ("Hcp", "Hcp"), ("scales", "scales")):
        e = relerr(gpu.get(key_g), ref.get(key_r))
        assert 1e-12 < e < 2e-5, (key_g, e)  # fp32-rounded Jacobian entries: not the fp64 bits, but close
    gpu.close()


@pytest.mark.parametrize("solver", ["pcg", "pcg_schur", "pcg_schur_implicit", "dense_schur"])
def test_mixed_lm_trace_close_to_fp64(oracle_mod, solver):
    prob = synth.make_config("ladybug-49")
    gs = dict(pcg=ga.SOLVER_PCG, pcg_schur=ga.SOLVER_PCG_SCHUR, pcg_schur_implicit=ga.SOLVER_PCG_SCHUR_IMPLICIT,
              dense_schur=ga.SOLVER_DENSE_SCHUR)[solver]
    os_ = dict(pcg=oracle_mod.SOLVER_PCG, pcg_schur=oracle_mod.SOLVER_PCG_SCHUR, pcg_schur_implicit=oracle_mod.SOLVER_PCG_SCHUR,
               dense_schur=oracle_mod.SOLVER_LDLT_SCHUR)[solver]
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    gpu.set_jacobian_precision(np.float32)
    ct_g, lt_g, st = gpu.levenberg_marquardt(solver=gs, iterations=6)
    ct_r, lt_r, _ = ref.levenberg_marquardt(solver=os_, iterations=6)
    assert len(ct_g) == len(ct_r)
    assert np.abs(ct_g - ct_r).max() / ct_r.max() < 1e-5
    assert ct_g[-1] < 0.1 * ct_g[0]
    gpu.close()


def test_fp32_problem_rejects_fp64_jacobians():
    prob = synth.make_config("mini-50")
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
    gpu.set_jacobian_precision(np.float32)  # no-op
    with pytest.raises(ga._lib.GraphiteError):
        gpu.set_jacobian_precision(np.float64)
    gpu.close()
