"""tests/cpp/test_pose_graph.hip (generic layer / pose engine) against oracle/pose_graph.py, and its time per LM iteration:
python tools/pose_graph_check.py [n_poses] [iterations] [pcg iterations] [pcg tolerance] [mode]"""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphite_amd import synth
from oracle.pose_graph import PoseGraphOracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
its = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pit = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ptol = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
mode = sys.argv[5] if len(sys.argv) > 5 else "manual"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p0, fx, e, m, info, tr = synth.make_pose_graph(n)
d = tempfile.mkdtemp()
f = os.path.join(d, "g.txt"); out = os.path.join(d, "out.txt")
delta = 3.0 if mode.endswith("huber") else 0.0
synth.write_pose_graph(f, p0, fx, e, m, info, huber_delta=delta)
rows = lambda s: np.array([[float(x) for x in ln.split()[1:4]] for ln in s.splitlines() if len(ln.split()) == 6 and ln.split()[0].isdigit()])
for env_extra, tag in (({"GRAPHITE_GENERIC_ONLY": "1"}, "generic kernels"), ({}, "default")):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([os.path.join(root, "build", "test_pose_graph"), f, "pcg", str(its), mode, str(pit), str(ptol), out], capture_output=True, text=True, env=env)
    if r.returncode != 0:
        print(r.stdout[-1500:], r.stderr[-1500:]); sys.exit(1)
    tr_gpu = rows(r.stdout)
    o = PoseGraphOracle(p0, fx, e, m, info, huber_delta=delta)
    t0 = time.time()
    ct, lt, st = o.levenberg_marquardt(iterations=its, pcg_max_iter=pit, pcg_tol=ptol)
    k = min(len(tr_gpu), len(ct) - 1)
    rel = np.max(np.abs(tr_gpu[:k, 1] - ct[1:k + 1]) / ct[1:k + 1])
    got = np.loadtxt(out)
    print(f"{tag}: {len(e)} factors, {n} poses; chi2 {ct[0]:.6g} -> {ct[-1]:.6g}; trace rel diff {rel:.2e} over {k} iterations; lambda rel diff "
          f"{np.max(np.abs(tr_gpu[:k, 2] - lt[1:k + 1]) / lt[1:k + 1]):.2e}; max |pose - oracle| {np.abs(got - o.x).max():.2e}; oracle pcg iterations {st['pcg_iterations']}")
    print("   ", [ln for ln in r.stdout.splitlines() if ln.startswith(("ENGINE", "LM_SECONDS"))])
    if os.environ.get("POSE_TABLE"): print(r.stdout)
