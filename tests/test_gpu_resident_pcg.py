"""The resident PCG launch (graphite_amd/csrc/kernels_rp.hpp, gr_bal_tuning.pcg_resident = 1): PCGSolver::solve
(/root/reference/include/graphite/solver/pcg.hpp:61-232) with every inner iteration and the trial step
(graph.hpp:292-309) inside ONE launch of 512-thread workgroups, one per CU, two grid barriers per inner iteration.

Held to the oracle's LM restatement and to the operator / update / direction launches it replaces: equal chi2 / damping
traces, equal inner-iteration and accept counts, equal final vertices — plain and Huber loss, the identity preconditioner,
identity damping, fixed vertices, rejected LM steps, rejected PCG iterations, fp32 — and its two refusals (a communicator,
a user-traits problem) fall back to the launches."""
import numpy as np
import pytest

import graphite_amd as ga
from graphite_amd import synth

pytestmark = pytest.mark.gpu


def run(prob, dtype, solver, iterations, resident, loss=None, fixed=None, **kw):
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    g.set_tuning(pcg_resident=1 if resident else 0)
    if loss is not None:
        g.set_loss(ga.LOSS_HUBER, loss)
    if fixed is not None:
        g.set_fixed(*fixed)
    ct, lt, st = g.levenberg_marquardt(solver=solver, iterations=iterations, **kw)
    c, p = g.get_params()
    g.close()
    return ct, lt, st, c, p


def same(a, b, rtol=1e-9):
    assert len(a[0]) == len(b[0])
    assert np.allclose(a[0], b[0], rtol=rtol) and np.allclose(a[1], b[1], rtol=1e-7)
    assert a[2]["pcg_iterations"] == b[2]["pcg_iterations"] and a[2]["accepted"] == b[2]["accepted"]
    assert np.allclose(a[3], b[3], rtol=1e-7, atol=1e-10) and np.allclose(a[4], b[4], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("name,iterations", [("mini-50", 12), ("ladybug-49", 15)])
@pytest.mark.parametrize("solver", [ga.SOLVER_PCG, ga.SOLVER_PCG_IDENTITY])
@pytest.mark.parametrize("use_identity", [False, True])
def test_resident_launch_equals_the_three_launches(oracle_mod, name, iterations, solver, use_identity):
    prob = synth.make_config(name)
    kw = dict(use_identity=use_identity)
    res = run(prob, np.float64, solver, iterations, True, **kw)
    base = run(prob, np.float64, solver, iterations, False, **kw)
    same(res, base)
    # one launch per solve: finalize_bj + resident + linearize per LM iteration (+ the first linearisation and the last finalisation)
    assert res[2]["kernel_launches"] <= 3 * iterations + 4 < base[2]["kernel_launches"]
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    osolver = oracle_mod.SOLVER_PCG if solver == ga.SOLVER_PCG else oracle_mod.SOLVER_PCG_IDENTITY
    ct_r, lt_r, st_r = ref.levenberg_marquardt(solver=osolver, iterations=iterations, **kw)
    assert len(ct_r) == len(res[0]) and np.allclose(res[0], ct_r, rtol=1e-8)
    assert st_r["pcg_iterations"] == res[2]["pcg_iterations"] and st_r["accepted"] == res[2]["accepted"]


def test_resident_launch_with_huber_loss_and_fixed_vertices(oracle_mod):
    prob = synth.make_config("mini-50")
    cf = np.zeros(len(prob.cameras), np.uint8); cf[[0, 7]] = 1
    pf = np.zeros(len(prob.points), np.uint8); pf[::17] = 1
    res = run(prob, np.float64, ga.SOLVER_PCG, 10, True, loss=1.5, fixed=(cf, pf))
    base = run(prob, np.float64, ga.SOLVER_PCG, 10, False, loss=1.5, fixed=(cf, pf))
    same(res, base)
    assert np.array_equal(res[3][[0, 7]], prob.cameras[[0, 7]]) and np.array_equal(res[4][::17], prob.points[::17])
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ref.set_loss(oracle_mod.LOSS_HUBER, 1.5)
    ref.set_fixed(cf, pf)
    ct_r, _, st_r = ref.levenberg_marquardt(solver=oracle_mod.SOLVER_PCG, iterations=10)
    assert np.allclose(res[0], ct_r, rtol=1e-8) and st_r["pcg_iterations"] == res[2]["pcg_iterations"]


def test_resident_launch_through_rejected_lm_steps_and_long_solves(oracle_mod):
    """Noisy observations, almost no damping, PCG run to convergence (30 inner iterations per solve): about half of the LM steps
    are rejected (the launch behind a rejected decision returns at once; the host reverts)."""
    prob = synth.make_problem(8, 200, 1600, seed=1, noise_px=30.0)
    kw = dict(initial_damping=1e-12, pcg_max_iter=30, pcg_tol=1e-30, pcg_rej=1e30)
    res = run(prob, np.float64, ga.SOLVER_PCG, 25, True, **kw)
    base = run(prob, np.float64, ga.SOLVER_PCG, 25, False, **kw)
    same(res, base, rtol=1e-8)
    assert res[2]["accepted"] < res[2]["iterations_run"], "the scenario is meant to contain rejected steps"
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ct_r, _, st_r = ref.levenberg_marquardt(solver=oracle_mod.SOLVER_PCG, iterations=25, **kw)
    assert np.allclose(res[0], ct_r, rtol=1e-6) and st_r["accepted"] == res[2]["accepted"]


def test_resident_launch_takes_a_rejected_pcg_iteration_back(oracle_mod):
    """rejection ratio 0.5: an inner iteration whose r.z does not halve is rejected (pcg.hpp:197-205) — x goes back to the
    iterate before it (here: the update of x that trails by one iteration is never applied) and the loop ends."""
    prob = synth.make_config("mini-50")
    kw = dict(pcg_max_iter=20, pcg_tol=1e-12, pcg_rej=0.5)
    res = run(prob, np.float64, ga.SOLVER_PCG, 8, True, **kw)
    base = run(prob, np.float64, ga.SOLVER_PCG, 8, False, **kw)
    same(res, base)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ct_r, _, st_r = ref.levenberg_marquardt(solver=oracle_mod.SOLVER_PCG, iterations=8, **kw)
    assert np.allclose(res[0], ct_r, rtol=1e-8) and st_r["pcg_iterations"] == res[2]["pcg_iterations"]
    assert res[2]["pcg_iterations"] < 8 * 20, "the scenario is meant to end solves on a rejected iteration"


def test_resident_launch_fp32(oracle_mod):
    prob = synth.make_config("ladybug-49")
    res = run(prob, np.float32, ga.SOLVER_PCG, 15, True)
    base = run(prob, np.float32, ga.SOLVER_PCG, 15, False)
    assert len(res[0]) == len(base[0]) and np.allclose(res[0], base[0], rtol=1e-5)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
    ct_r, _, _ = ref.levenberg_marquardt(solver=oracle_mod.SOLVER_PCG, iterations=15)
    assert np.allclose(res[0], ct_r, rtol=1e-5)  # the suite's fp32 trace bar (tests/test_gpu_parity.py)


def test_resident_launch_at_the_headline_size():
    """Ladybug-1723 fp64 (BASELINE configs[2]): 5.2 observation blocks per wave, 3.8 owned tiles per workgroup — the shape the
    per-lane capacities (6 blocks, 4 tiles) were chosen for."""
    prob = synth.make_config("ladybug-1723")
    res = run(prob, np.float64, ga.SOLVER_PCG, 6, True)
    base = run(prob, np.float64, ga.SOLVER_PCG, 6, False)
    assert np.allclose(res[0], base[0], rtol=1e-10) and res[2]["pcg_iterations"] == base[2]["pcg_iterations"]
    scale = np.maximum(np.abs(base[4]), 1.0)
    assert np.max(np.abs(res[4] - base[4]) / scale) < 1e-7 and np.allclose(res[3], base[3], rtol=1e-7, atol=1e-9)
    assert res[2]["kernel_launches"] < base[2]["kernel_launches"]


def test_a_problem_that_does_not_fit_keeps_the_launches():
    """more than 6 x 512 observations per CU: the tuning field is accepted, the solver runs its launches"""
    prob = synth.make_problem(60, 30000, 900000, seed=9, window=16)
    res = run(prob, np.float64, ga.SOLVER_PCG, 2, True)
    base = run(prob, np.float64, ga.SOLVER_PCG, 2, False)
    assert res[2]["kernel_launches"] == base[2]["kernel_launches"] and np.allclose(res[0], base[0], rtol=1e-10)
