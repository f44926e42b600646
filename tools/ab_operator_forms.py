# the operator launch of the three PCG forms, stand-alone (gr_bal_diag_time): direction-kernel form, lazy direction, single reduction
import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga
from graphite_amd import synth
for name, dt in (("ladybug-1723", np.float64), ("venice-1778", np.float32)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]: continue
    prob = synth.make_config(name)
    for form, kw in (("direction", dict(pcg_lazy=0, pcg_single_reduction=0)), ("lazy", dict(pcg_lazy=1, pcg_single_reduction=0)), ("single-reduction", dict(pcg_lazy=0, pcg_single_reduction=1))):
        g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
        g.set_tuning(**kw)
        g.solver_update_structure(ga.SOLVER_PCG)
        f = g.lib.gr_bal_diag_time; f.restype = C.c_double
        print(name, form, "operator %.2f us  update %.2f us" % (f(g.h, 0, 0, 50), f(g.h, 3, 0, 50)), flush=True)
        g.close()
