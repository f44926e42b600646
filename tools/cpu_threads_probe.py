"""Host-side probe: how the OpenMP CPU baseline scales with the thread count on this box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from graphite_amd import synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
prob = synth.make_config("ladybug-1723")
b = oracle.CpuBaseline(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx)
for thr in (1, 4, 8, 16, 32, 64, 128):
    b.reset()
    ct, lt, st, tm = b.levenberg_marquardt(oracle.SOLVER_PCG, 4, threads=thr)
    print(thr, "threads: PCG LM it/s", round(st["iterations_run"] / st["loop_seconds"], 3), flush=True)
