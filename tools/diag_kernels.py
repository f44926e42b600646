import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga
from graphite_amd import synth
name=sys.argv[1] if len(sys.argv)>1 else 'ladybug-1723'
dt=np.float64 if (len(sys.argv)<3 or sys.argv[2]=='f64') else np.float32
prob=synth.make_config(name)
g=ga.BalProblem(prob.cameras,prob.points,prob.obs,prob.cam_idx,prob.pt_idx,dtype=dt)
f=g.lib.gr_bal_diag_time; f.restype=C.c_double
g.solver_update_structure(ga.SOLVER_PCG)  # the observation order (plain / point-tiled) is decided here
def t(which,var=0,reps=50): return f(g.h,C.c_int(which),C.c_int(var),C.c_int(reps))
print('operator variants:')
for v in (0,1,2,3,8,16,31,32,64,128,224,95,255): print('  var',v, round(t(0,v),2),'us')
for w,nm in ((1,'linearize'),(2,'chi2'),(3,'pcg_update'),(4,'pcg_direction'),(5,'lin_finalize')): print(nm, round(t(w),2),'us')
print('chi2+rho', round(t(2,1),2))
