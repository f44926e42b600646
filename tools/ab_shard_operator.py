# One landmark shard of a workload treated as a problem of its own: the operator / linearise / update launches stand-alone
# (gr_bal_diag_time) under the observation orders and record layouts the tuner chooses between
import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga
from graphite_amd import synth, dist as gdist
name = sys.argv[1] if len(sys.argv) > 1 else "final-13682"
dt = np.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else np.float32
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
prob = synth.make_config(name)
part = gdist.partition_by_landmark(prob, 0, world)
print(name, "shard 0 of", world, ":", part.shape, flush=True)
for tiles, g3g in ((0, 0), (0, 1), (8, 0), (8, 1), (16, 0)):
    for rec in (0, 1):
        g = ga.BalProblem(part.cameras, part.points, part.obs, part.cam_idx, part.pt_idx, dtype=dt)
        g.set_tuning(pcg_lazy=0, point_records=rec, point_tiles=tiles, g3_gather=g3g)
        g.solver_update_structure(ga.SOLVER_PCG)
        f = g.lib.gr_bal_diag_time; f.restype = C.c_double
        print("point tiles %2d g3 in observation order %d records %d: operator %.1f us  linearise %.1f us  update %.1f us" % (tiles, g3g, rec, f(g.h, 0, 0, 20), f(g.h, 1, 0, 20), f(g.h, 3, 0, 20)), flush=True)
        g.close()
