import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga
from graphite_amd import synth
prob = synth.make_config("ladybug-1723")
for gm in (0, 3, 4):
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    g.set_tuning(grid_mult=gm, pcg_lazy=0, point_tiles=0, point_records=0)
    f = g.lib.gr_bal_diag_time; f.restype = C.c_double
    print("grid_mult", gm, "operator %.2f us  linearize %.2f us" % (f(g.h, 0, 0, 50), f(g.h, 1, 0, 50)))
    g.close()
