import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga
from graphite_amd import synth
prob = synth.make_config("ladybug-1723")
g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
f = g.lib.gr_bal_diag_time; f.restype = C.c_double
for i in range(2):
    print("update regular %.2f us, first-iteration form %.2f us, direction %.2f" % (f(g.h, 3, 0, 50), f(g.h, 3, 100, 50), f(g.h, 4, 0, 50)))
