# A/B of the operator's point gather: plain pts / ps arrays against the [X Y Z | s.p] record (gr_bal_diag_time, back to back)
import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga
from graphite_amd import synth
for name, dt in (("ladybug-1723", np.float64), ("venice-1778", np.float32), ("final-13682", np.float64)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]: continue
    prob = synth.make_config(name)
    for rec in (0, 1):
        g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
        g.set_tuning(pcg_lazy=0, point_records=rec)
        g.solver_update_structure(ga.SOLVER_PCG)  # observation order and record layout are decided here
        f = g.lib.gr_bal_diag_time; f.restype = C.c_double
        print(name, "records", rec, "operator %.2f us" % f(g.h, 0, 0, 50), flush=True)
        g.close()
