"""Oracle: the two Levenberg-Marquardt drivers (optimizer/levenberg_marquardt.hpp:110-242 and :255-418).
levenberg_marquardt2 is the same iteration with the ORB-SLAM-style termination of :404-414: leave after
three consecutive ACCEPTED steps that each lowered chi2 by less than 0.1 %."""
import numpy as np
import pytest

from graphite_amd import synth


@pytest.mark.parametrize("solver", ["SOLVER_PCG", "SOLVER_PCG_SCHUR"])
def test_early_stop_is_a_prefix_ending_on_three_small_gains(oracle_mod, solver):
    prob = synth.make_config("mini-50")
    kind = getattr(oracle_mod, solver)
    o = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ct, lt, st = o.levenberg_marquardt(solver=kind, iterations=40)
    o.set_params(prob.cameras, prob.points)
    ct2, lt2, st2 = o.levenberg_marquardt(solver=kind, iterations=40, early_stop=True)
    k = len(ct2)
    assert 4 <= k < len(ct)                                   # it did leave early
    assert np.array_equal(ct2, ct[:k]) and np.array_equal(lt2, lt[:k])
    # replay the rule on the full trace: an accepted step lowers lambda (mu *= alpha <= 2/3), a rejected one raises it
    num_bad, stop = 0, None
    for i in range(len(ct) - 1):
        if lt[i + 1] < lt[i]:
            num_bad = num_bad + 1 if (ct[i] - ct[i + 1]) * 1e3 < ct[i] else 0
            if num_bad >= 3:
                stop = i + 1
                break
    assert stop is not None and stop + 1 == k
    assert st2["iterations_run"] == stop


def test_circle_oracle_configs0(oracle_mod):
    """oracle/circle_fit.hpp (BASELINE configs[0], examples/circle.cu:87-160 at 100 vertices): both inner solvers of the
    restatement — sparse LDL^T of the damped Hessian and PCG with the identity preconditioner — walk the same LM trace on this
    block-diagonal system, every free vertex lands on the circle, the fixed vertex and the vertex whose factor is off keep their
    bits, and what is left of chi2 is the fixed vertex's own residual."""
    import numpy as np
    from graphite_amd import synth
    n, R = 100, 4.0
    pts = synth.make_circle(n, R)
    fixed = np.zeros(n); fixed[n - 1] = 1
    on = np.ones(n); on[2] = 0
    a = oracle_mod.circle_lm(pts, R, fixed, on, solver="eigen", iterations=100)
    b = oracle_mod.circle_lm(pts, R, fixed, on, solver="pcg", iterations=100)
    assert len(a[0]) == len(b[0]) and np.allclose(a[0], b[0], rtol=1e-10) and a[3]["accepted"] == b[3]["accepted"]
    assert b[3]["pcg_iterations"] > 0 and a[3]["pcg_iterations"] == 0
    for ct, lt, p, st in (a, b):
        free = np.delete(np.arange(n), [2, n - 1])
        assert np.allclose(np.hypot(p[free, 0], p[free, 1]), R, atol=1e-12)
        assert np.array_equal(p[2], pts[2]) and np.array_equal(p[n - 1], pts[n - 1])
        r_fixed = pts[n - 1] @ pts[n - 1] - R * R
        assert abs(ct[-1] - r_fixed ** 2) < 1e-12 and ct[0] > 100 * ct[-1]
    # the first step is the damped Gauss-Newton step of each vertex in its scaled coordinates (graph.hpp:254-287): with J = [2x, 2y]
    # scaled to signs, p moves by -r / (2 + mu) * [1 / (2x), 1 / (2y)]
    one = oracle_mod.circle_lm(pts, R, fixed, on, solver="eigen", iterations=1)
    i = 0
    x, y = pts[i]
    r0 = x * x + y * y - R * R
    mu = 1e-6
    step = -r0 / (2.0 + mu * 1.0) * np.array([1 / (2 * x), 1 / (2 * y)])
    assert np.allclose(one[2][i] - pts[i], step, rtol=1e-9)
