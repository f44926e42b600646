"""Phase stamps of the resident PCG launch (GR_RP_DEBUG=1): python tools/rp_phases.py [config] [f64|f32] [pcg_max_iter]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GR_RP_DEBUG"] = "1"
os.environ["GR_PCG_RESIDENT"] = "1"
import graphite_amd as ga  # noqa: E402
from graphite_amd import synth  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "ladybug-1723"
dtype = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prob = synth.make_config(name)
g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
for it in (3, 3):
    g.set_params(prob.cameras, prob.points)
    ct, lt, st = g.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=it, pcg_max_iter=mi)
    print(st["pcg_iterations"], ct)
g.close()
