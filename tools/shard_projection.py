#!/usr/bin/env python3
"""PROJECTION (not a measurement) of the 8-rank landmark-sharded run from ONE GPU.

The 8-way landmark partition of a workload (default Venice-1778 fp32, BASELINE.json configs[3]) is built exactly as
bench.py builds it; every shard is then run ALONE on this GPU through the sharded code path (communicator of one rank:
every all-reduce of the real run is issued, as a 1-rank operation), with the same fixed number of inner iterations as
the unsharded run, so that all nine runs do the same algorithmic work per LM iteration.  Printed:

    T1                  seconds per LM iteration, unsharded, no communicator
    T_r, r = 0..7       seconds per LM iteration of shard r alone (includes its 1-rank collectives)
    c                   collectives per LM iteration
    projected T8      = max_r T_r + c * (L8 - L1)     L1 = measured latency of the 1-rank all-reduce of a camera vector,
                                                      L8 = assumed latency of the 8-rank one-shot mailbox all-reduce
    projected speed-up = T1 / projected T8

L8 cannot be measured on a 1-GPU box: the figure used is the measured two-process latency on one GPU (tools/ipc_latency.py,
profiles/r02_v3_ipc_allreduce_latency_2_processes_one_gpu.txt: 11.9 us for 121 KB) scaled by message size, i.e. the
mechanism's floor WITHOUT an xGMI hop — the projection is an upper bound on the scaling.  What it does show is the part that
cannot shrink: the replicated camera-space work, launch floors and collectives that max_r T_r still contains.
"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="venice-1778")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--pcg-iterations", type=int, default=10)
    ap.add_argument("--l8-us", type=float, default=None, help="assumed 8-rank all-reduce latency in us (default: scaled from the 2-process figure)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    import torch, torch.distributed as dist
    dist.init_process_group("gloo")
    import graphite_amd as ga
    from graphite_amd import synth, dist as gdist
    dtype = np.float32 if args.dtype == "f32" else np.float64
    prob = synth.make_config(args.workload)
    Nc, Np, No = prob.shape
    kw = dict(solver=ga.SOLVER_PCG, initial_damping=1e-4, pcg_max_iter=args.pcg_iterations, pcg_tol=0.0, pcg_rej=1e30)

    def run(g, part):
        g.levenberg_marquardt(iterations=3, **kw)
        best = None
        for _ in range(3):
            g.set_params(part.cameras, part.points)
            torch.cuda.synchronize()
            ct, lt, st = g.levenberg_marquardt(iterations=args.steps, **kw)
            per = st["loop_seconds"] / max(st["iterations_run"], 1)
            if best is None or per < best[0]:
                best = (per, st)
        return best

    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    t1, st1 = run(g, prob)
    g.close()
    shards = []
    lat1 = None
    for r in range(args.world):
        part = gdist.partition_by_landmark(prob, r, args.world)
        g = ga.BalProblem(part.cameras, part.points, part.obs, part.cam_idx, part.pt_idx, dtype=dtype, shard=True)
        gdist.init_comm_ipc(g, 0, 1, slot_bytes=max(4 << 20, 2 * 90 * Nc * 8), rccl_fallback=False)  # one slot holds the largest grouped message (Hcc + bc + chi2)
        tr, st = run(g, part)
        if lat1 is None:
            f = g.lib.gr_bal_diag_time; f.restype = C.c_double
            lat1 = f(g.h, C.c_int(8), C.c_int(0), C.c_int(200))  # us per 1-rank all-reduce of a camera-space vector
        shards.append({"rank": r, "points": int(part.shape[1]), "observations": int(part.shape[2]), "seconds_per_lm_iteration": tr,
                       "collectives_per_lm_iteration": st["collectives"] / max(st["iterations_run"], 1), "pcg_iterations": st["pcg_iterations"]})
        g.close()
    w = np.dtype(dtype).itemsize
    msg_kb = 9 * Nc * w / 1024.0
    l8 = args.l8_us if args.l8_us is not None else 4.3 + (11.9 - 4.3) * min(1.0, msg_kb / 121.0)
    c = max(s["collectives_per_lm_iteration"] for s in shards)
    tmax = max(s["seconds_per_lm_iteration"] for s in shards)
    t8 = tmax + c * (l8 - lat1) * 1e-6
    res = {"kind": "PROJECTION from one GPU (tools/shard_projection.py), not a multi-GPU measurement",
           "workload": f"{args.workload} {args.dtype}, block-Jacobi PCG, {args.pcg_iterations} fixed inner iterations, {args.world} landmark shards",
           "T1_seconds_per_lm_iteration": t1, "T1_lm_iterations_per_sec": 1.0 / t1, "shards": shards,
           "max_shard_seconds_per_lm_iteration": tmax, "collectives_per_lm_iteration": c,
           "L1_us_one_rank_allreduce_camera_vector": lat1, "L8_us_assumed": l8, "camera_vector_kb": msg_kb,
           "projected_T8_seconds_per_lm_iteration": t8, "projected_lm_iterations_per_sec": 1.0 / t8, "projected_speedup": t1 / t8,
           "speedup_if_collectives_were_free": t1 / (tmax - c * lat1 * 1e-6),
           "ideal": args.world}
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
