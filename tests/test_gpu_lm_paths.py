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


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_schur_head_that_goes_on_after_a_rejected_step(oracle_mod, monkeypatch, dtype):
    """PCGSchurSolver on a small reduced system (device-decided head, kernels_sf.hpp): by default a rejected step does not stop the head —
    the finalisation launch takes the vertices back, keeps its sums, raises the damping and the head runs on (second buffer of camera-point
    blocks, per-point sums kept in double).  Against GR_SCHUR_FUSED=1 (the head stops, the host reverts and re-linearises) and
    GR_SCHUR_FUSED=0 (host-driven loop), and against the oracle.
    Scenario 1 — noisy observations, almost no damping, PCG run to convergence: the first seven steps are rejected in a row, about half of all.
    fp64: the same chi2 / damping traces and final vertices to the last bit as the stopping form.  fp32: the reduced system is close to
    singular here and turns one ulp of an input into 1e-4 of the step — the three forms differ from each other by that much (the stopping
    form re-linearises with camera packs rebuilt by another kernel, the form that goes on keeps the original linearisation): held at 2e-3,
    with equal accept counts.
    Scenario 2 — Ladybug-49 with the bench line's options (9 of 20 steps rejected): fp32 traces equal at 1e-6."""
    prob = synth.make_problem(8, 200, 1600, seed=1, noise_px=30.0)
    kw = dict(initial_damping=1e-12, pcg_max_iter=30, pcg_tol=1e-30, pcg_rej=1e30)
    res = {}
    for mode in ("2", "1", "0"):
        monkeypatch.setenv("GR_SCHUR_FUSED", mode)
        res[mode] = run(prob, dtype, ga.SOLVER_PCG_SCHUR, 25, **kw)
    a, b, c = res["2"], res["1"], res["0"]
    assert a[2]["accepted"] < a[2]["iterations_run"], "the scenario is meant to contain rejected steps"
    assert a[2]["accepted"] == b[2]["accepted"] == c[2]["accepted"]
    if dtype == np.float64:
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])  # chi2 and damping traces: bit for bit
        assert a[2]["pcg_iterations"] == b[2]["pcg_iterations"]
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])  # final cameras and points
        assert np.allclose(a[0], c[0], rtol=1e-8)
        ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
        ct_r, lt_r, st_r = ref.levenberg_marquardt(solver=oracle_mod.SOLVER_PCG_SCHUR, iterations=25, **kw)
        assert len(a[0]) == len(ct_r) and np.allclose(a[0], ct_r, rtol=1e-6) and a[2]["accepted"] == st_r["accepted"]
    else:
        assert np.allclose(a[0], b[0], rtol=2e-3) and np.allclose(a[0], c[0], rtol=2e-3) and np.allclose(a[1], b[1], rtol=2e-2)
        l49 = synth.make_config("ladybug-49")
        kw49 = dict(initial_damping=1e-4, pcg_max_iter=10, pcg_tol=1.0, pcg_rej=5.0)
        r = {}
        for mode in ("2", "1"):
            monkeypatch.setenv("GR_SCHUR_FUSED", mode)
            r[mode] = run(l49, np.float32, ga.SOLVER_PCG_SCHUR, 20, **kw49)
        assert r["2"][2]["accepted"] < r["2"][2]["iterations_run"]
        assert r["2"][2]["accepted"] == r["1"][2]["accepted"] and np.allclose(r["2"][0], r["1"][0], rtol=1e-6)


def test_a_cooperative_launch_that_times_out_is_not_sticky(monkeypatch):
    """the cooperative PCG on S (one launch, software grid barrier) is armed only when the occupancy query says its
    workgroups are all resident, and a barrier that times out all the same (CUs held by someone else) no longer breaks the handle:
    the flag is cleared, the stale step is taken back, the iteration runs again on the host-driven loop and so do all later ones.
    GR_TEST_COOP_TIMEOUT=1 makes the first cooperative launch of a handle wait for a workgroup that does not exist (2 s bound)."""
    prob = synth.make_config("mini-50")
    base = run(prob, np.float64, ga.SOLVER_PCG_SCHUR, 8)
    monkeypatch.setenv("GR_TEST_COOP_TIMEOUT", "1")
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ct, lt, st = g.levenberg_marquardt(solver=ga.SOLVER_PCG_SCHUR, iterations=8)
    assert len(ct) == len(base[0]) and np.allclose(ct, base[0], rtol=1e-8)
    assert st["accepted"] == base[2]["accepted"]
    import time
    g.set_params(prob.cameras, prob.points)
    t0 = time.perf_counter()
    ct2, _, st2 = g.levenberg_marquardt(solver=ga.SOLVER_PCG_SCHUR, iterations=8)  # same handle: host-driven loop, no new time-out
    assert time.perf_counter() - t0 < 1.0
    assert np.allclose(ct2, base[0], rtol=1e-8)
    g.close()
