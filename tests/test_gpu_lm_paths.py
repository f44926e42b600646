"""The LM host loop's shortcuts do not change the optimisation: speculative trial linearisation, the fused iteration head
(k_finalize_bj: finalisation + block-Jacobi + PCG start + device-side accept decision) with the trial step applied by the
last direction launch, trial linearisation enqueued ahead of the PCG exit flag — each switched off through its knob,
the chi2 / lambda traces and the vertices must agree with the default path (same arithmetic; only the order of
the dot-product partials may differ)."""
import numpy as np
import pytest

import graphite_amd as ga
from graphite_amd import synth

pytestmark = pytest.mark.gpu


def run(prob, dtype, solver, iterations, **kw):
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    ct, lt, st = g.levenberg_marquardt(solver=solver, iterations=iterations, **kw)
    c, p = g.get_params()
    g.close()
    return ct, lt, st, c, p


@pytest.mark.parametrize("name,iterations", [("mini-50", 12), ("ladybug-49", 15)])
@pytest.mark.parametrize("solver", [ga.SOLVER_PCG, ga.SOLVER_PCG_IDENTITY])
@pytest.mark.parametrize("knob", ["GR_LM_AHEAD", "GR_LM_SPECULATE", "GR_LM_FUSED"])
def test_shortcuts_leave_the_trace_alone(monkeypatch, name, iterations, solver, knob):
    prob = synth.make_config(name)
    base = run(prob, np.float64, solver, iterations)
    monkeypatch.setenv(knob, "0")
    off = run(prob, np.float64, solver, iterations)
    assert len(base[0]) == len(off[0])
    assert np.allclose(base[0], off[0], rtol=1e-9) and np.allclose(base[1], off[1], rtol=1e-7)
    assert base[2]["pcg_iterations"] == off[2]["pcg_iterations"] and base[2]["accepted"] == off[2]["accepted"]
    assert np.allclose(base[3], off[3], rtol=1e-8, atol=1e-11) and np.allclose(base[4], off[4], rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("knob", ["GR_LM_AHEAD", "GR_LM_SPECULATE", "GR_LM_FUSED"])
def test_rejected_steps_go_through_every_path(oracle_mod, monkeypatch, knob):
    """Noisy observations, almost no damping, PCG run to convergence: about half of the steps are rejected (speculative
    linearisations thrown away, streaks restarting).  Default path, knob-off path and the oracle agree."""
    prob = synth.make_problem(8, 200, 1600, seed=1, noise_px=30.0)
    kw = dict(initial_damping=1e-12, pcg_max_iter=30, pcg_tol=1e-30, pcg_rej=1e30)
    base = run(prob, np.float64, ga.SOLVER_PCG, 25, **kw)
    monkeypatch.setenv(knob, "0")
    off = run(prob, np.float64, ga.SOLVER_PCG, 25, **kw)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ct_r, lt_r, st_r = ref.levenberg_marquardt(solver=oracle_mod.SOLVER_PCG, iterations=25, **kw)
    assert len(base[0]) == len(off[0]) == len(ct_r)
    assert np.allclose(base[0], off[0], rtol=1e-8) and np.allclose(base[0], ct_r, rtol=1e-6)
    assert base[2]["accepted"] == off[2]["accepted"] == st_r["accepted"]
    assert base[2]["accepted"] < base[2]["iterations_run"], "the scenario is meant to contain rejected steps"
