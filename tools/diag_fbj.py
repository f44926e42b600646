# k_finalize_bj in isolation (back to back, caches warm): whole, cameras only, points only, with the decision prologue, one tile per workgroup;
# with a GR_DIAG build of the library (make EXTRA=-DGR_DIAG) also the ablations of its point part
import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga
from graphite_amd import synth
w = sys.argv[1] if len(sys.argv) > 1 else "ladybug-1723"
p = synth.make_config(w)
g = ga.BalProblem(p.cameras, p.points, p.obs, p.cam_idx, p.pt_idx, dtype=np.float64)
g.set_tuning(pcg_lazy=0)
f = g.lib.gr_bal_diag_time; f.restype = C.c_double
rows = [("finalize (old)", 5, 0), ("finalize points", 5, 2), ("block_jacobi fused start (old)", 6, 3), ("bj points", 6, 2),
        ("fbj", 9, 0), ("fbj cameras", 9, 1), ("fbj points", 9, 2), ("fbj + decision", 9, 4)]
if len(sys.argv) > 2:
    rows += [("fbj points, few stores", 9, 2 + 16 * 1), ("fbj points, no inverse math", 9, 2 + 16 * 2), ("fbj points, no record loads", 9, 2 + 16 * 4), ("fbj points, no atomics", 9, 2 + 16 * 8),
             ("fbj points, few stores + no math", 9, 2 + 16 * 3), ("fbj points, stores+math+loads off", 9, 2 + 16 * 7), ("fbj points, all off", 9, 2 + 16 * 15)]
for name, which, var in rows:
    print("%-36s %7.2f us" % (name, f(g.h, which, var, 20)))
