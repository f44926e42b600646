import sys, ctypes as C, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga
from graphite_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else 'ladybug-1723'
dt = np.float64 if (len(sys.argv) < 3 or sys.argv[2] == 'f64') else np.float32
prob = synth.make_config(name)
g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
f = g.lib.gr_bal_diag_time; f.restype = C.c_double
def t(which, var=0, reps=100): return f(g.h, C.c_int(which), C.c_int(var), C.c_int(reps))
print("block_jacobi: all %.2f us, cameras only %.2f, points only %.2f, with fused PCG start %.2f" % (t(6, 0), t(6, 1), t(6, 2), t(6, 3)))
