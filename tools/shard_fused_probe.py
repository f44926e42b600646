#!/usr/bin/env python3
"""One landmark shard of Venice-1778 fp32 run alone: kernel times of the sharded inner iteration in its unfused form, fused with
one rank, and fused with V virtual ranks (gr_bal_tuning.shard_virtual_ranks) — where the fused form's time goes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch.distributed as dist
dist.init_process_group("gloo")
import graphite_amd as ga
from graphite_amd import synth, dist as gdist
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prob = synth.make_config(sys.argv[2] if len(sys.argv) > 2 else "venice-1778")
part = gdist.partition_by_landmark(prob, 0, world)
dt = np.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else np.float32
kw = dict(solver=ga.SOLVER_PCG, initial_damping=1e-4, pcg_max_iter=10, pcg_tol=0.0, pcg_rej=1e30)
forms = (("unfused", dict(shard_fused=0)), ("unfused, single-reduction form", dict(shard_fused=0, pcg_single_reduction=1)), ("fused x1", dict(shard_fused=1, shard_virtual_ranks=0, pcg_single_reduction=1)),
                   (f"fused x{world} virtual", dict(shard_fused=1, shard_virtual_ranks=world, pcg_single_reduction=1)))
if len(sys.argv) > 4: forms = tuple(f for f in forms if f[0].startswith(sys.argv[4]))
for name, tune in (("WHOLE problem, one GPU, no communicator", None),) + forms:
    if tune is None:
        g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
        tune = {}; src = prob
    else:
        src = part
        g = ga.BalProblem(part.cameras, part.points, part.obs, part.cam_idx, part.pt_idx, dtype=dt, shard=True)
        gdist.init_comm_ipc(g, 0, 1, slot_bytes=max(4 << 20, world * (90 * prob.shape[0] * 8 + 4096)), rccl_fallback=False)  # the virtual ranks cut a slot into `world` pieces
    if tune:
        if tune.get("shard_virtual_ranks"): gdist.set_contributors(g, gdist.contributor_masks(prob, world), 0)
        g.set_tuning(**tune)
    g.levenberg_marquardt(iterations=3, **kw)
    g.set_params(src.cameras, src.points)
    ct, lt, st = g.levenberg_marquardt(iterations=10, **kw)
    g.set_params(src.cameras, src.points)
    _, _, stp = g.levenberg_marquardt(iterations=10, profile=True, **kw)
    ks = g.kernel_stats()
    print(f"== {name}: {st['loop_seconds'] / st['iterations_run'] * 1e3:.3f} ms per LM iteration, {st['kernel_launches'] / st['iterations_run']:.1f} launches, {st['collectives'] / st['iterations_run']:.1f} collectives")
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"])[:9]:
        print(f"   {k:24s} {v['launches']:5d} launches  {v['total_ms'] * 1e3 / max(v['launches'], 1):8.2f} us avg")
    g.close()
dist.destroy_process_group()
