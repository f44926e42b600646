import sys, ctypes as C, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga
from graphite_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else 'ladybug-1723'
dt = np.float64 if (len(sys.argv) < 3 or sys.argv[2] == 'f64') else np.float32
prob = synth.make_config(name)
g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
f = g.lib.gr_bal_diag_time; f.restype = C.c_double
g.solver_update_structure(ga.SOLVER_PCG)  # the observation order (plain / point-tiled) is decided here
def t(which, var=0, reps=100): return f(g.h, C.c_int(which), C.c_int(var), C.c_int(reps))
print("linearize variants (1 no point-record write, 2 no camera reduction, 4 no point gather, 8 no J math):")
for v in (0, 1, 2, 3, 4, 7, 8, 15): print("  var %2d  %.2f us" % (v, t(1, v)))
