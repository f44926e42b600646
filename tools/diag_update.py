"""A/B of the PCG update kernel in isolation: whole, camera part, point part, grid sizes."""
import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga
from graphite_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else 'ladybug-1723'
dt = np.float64 if (len(sys.argv) < 3 or sys.argv[2] == 'f64') else np.float32
prob = synth.make_config(name)
g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
f = g.lib.gr_bal_diag_time; f.restype = C.c_double
def t(which, var=0, reps=100): return f(g.h, C.c_int(which), C.c_int(var), C.c_int(reps))
print('update whole', round(t(3), 2), 'cam only', round(t(3, 1), 2), 'points only', round(t(3, 2), 2))
for nb in (128, 256, 512, 1024, 2048): print('  blocks', nb, round(t(3, nb), 2))
print('block_jacobi', round(t(6), 2), 'cameras only', round(t(6, 1), 2), 'points only', round(t(6, 2), 2), 'with PCG start', round(t(6, 3), 2))
print('finalize', round(t(5), 2), 'camera part', round(t(5, 1), 2), 'point part', round(t(5, 2), 2))
print('direction', round(t(4), 2), 'finalize', round(t(5), 2), 'operator', round(t(0), 2), 'linearize', round(t(1), 2))
