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
