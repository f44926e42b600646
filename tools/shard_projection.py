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

Round 4: (i) the inner iteration's message is FUSED into the operator / update launches (gr_bal_tuning.shard_fused) and each
shard run plays all `world` ranks of that message on its own mailbox (gr_bal_tuning.shard_virtual_ranks: `world` stores per
pushed value, `world` flags, `world` slots summed per consumed value), so T_r CONTAINS the N-rank cost of the fused messages;
(ii) the collectives that still have a kernel of their own (the linearisation group, the closing scalars) are priced with
L8 MEASURED between `world` processes sharing this GPU (--l8-us from tools/ipc_latency.py <world>; the assumed 1-rank-like
figure of round 3 is gone).  What remains outside the measurement is the xGMI hop itself (~2 us per message on MI300-class
parts): --hop-us adds it per message, fused ones included.  Still a projection — no run here crosses an xGMI link.
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
    ap.add_argument("--l8-us", type=float, default=None, help="MEASURED latency (us) of the mailbox all-reduce of a camera vector between `world` processes on this GPU (tools/ipc_latency.py <world>)")
    ap.add_argument("--l8-large-us", type=float, default=None, help="the same for the linearisation group (Hcc + bc + chi2: 90 Nc scalars), one per LM iteration")
    ap.add_argument("--l8-small-us", type=float, default=None, help="the same for a message of a few scalars (the closing dots of a solve that ran into its cap)")
    ap.add_argument("--hop-us", type=float, default=2.0, help="xGMI hop added to every message (fused ones included): not measurable on one GPU")
    ap.add_argument("--link-gbs", type=float, default=153.0, help="xGMI link bandwidth (GB/s): every message also pays bytes / link (the peers are reached over their own links in parallel)")
    ap.add_argument("--unfused", action="store_true", help="round-3 form: a kernel of its own for every all-reduce")
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
    has = gdist.contributor_masks(prob, args.world, point_weight=gdist.point_weight_for(dtype))
    print("contributors per camera: mean %.2f of %d ranks" % (has.sum(0).mean(), args.world), file=sys.stderr)
    for r in range(args.world):
        part = gdist.partition_by_landmark(prob, r, args.world, point_weight=gdist.point_weight_for(dtype))
        g = ga.BalProblem(part.cameras, part.points, part.obs, part.cam_idx, part.pt_idx, dtype=dtype, shard=True)
        gdist.init_comm_ipc(g, 0, 1, slot_bytes=max(4 << 20, args.world * (90 * Nc * 8 + 4096)), rccl_fallback=False)  # one slot holds the largest grouped message (Hcc + bc + chi2)
        gdist.set_contributors(g, has, r)  # the masks the ranks of a real run agree on; here one rank plays them all
        if args.unfused:
            g.set_tuning(shard_fused=0)
        else:
            g.set_tuning(shard_fused=1, shard_virtual_ranks=args.world, pcg_single_reduction=1)
        tr, st = run(g, part)
        # per-phase device time of this shard (one more, profiled, pass): a sharded iteration costs the slowest rank of EACH phase
        g.set_params(part.cameras, part.points)
        _, _, stp = g.levenberg_marquardt(iterations=args.steps, profile=True, **kw)
        ksr = g.kernel_stats()
        lin_names = ("linearize", "linearize_hcp", "linearize_finalize", "finalize_bj", "chi2")
        lin_s = sum(v["total_ms"] for k, v in ksr.items() if k in lin_names) * 1e-3 / max(stp["iterations_run"], 1)
        if lat1 is None:
            f = g.lib.gr_bal_diag_time; f.restype = C.c_double
            lat1 = f(g.h, C.c_int(8), C.c_int(0), C.c_int(200))  # us per 1-rank all-reduce of a camera-space vector
            lat1_large = f(g.h, C.c_int(8), C.c_int(90 * Nc), C.c_int(200))
            lat1_small = f(g.h, C.c_int(8), C.c_int(8), C.c_int(200))
        shards.append({"rank": r, "points": int(part.shape[1]), "observations": int(part.shape[2]), "seconds_per_lm_iteration": tr,
                       "linearise_seconds_per_lm_iteration": lin_s, "other_seconds_per_lm_iteration": max(0.0, tr - lin_s),
                       "collectives_per_lm_iteration": st["collectives"] / max(st["iterations_run"], 1), "pcg_iterations": st["pcg_iterations"],
                       "kernel_launches_per_lm_iteration": st["kernel_launches"] / max(st["iterations_run"], 1),
                       "fused_messages_per_lm_iteration": st["fused_messages"] / max(st["iterations_run"], 1)})
        g.close()
    w = np.dtype(dtype).itemsize
    msg_kb = 9 * Nc * w / 1024.0
    if args.l8_us is None:
        raise SystemExit("--l8-us: pass the latency measured by `python tools/ipc_latency.py %d` for a camera-space vector" % args.world)
    l8 = args.l8_us
    c = max(s["collectives_per_lm_iteration"] for s in shards)
    inner = max(s["fused_messages_per_lm_iteration"] for s in shards)      # messages that travel inside launches (inner iterations + the linearisation)
    c_kernel = c if args.unfused else max(0.0, c - inner)                   # collectives that still have a kernel of their own
    tmax = max(s["seconds_per_lm_iteration"] for s in shards)
    # bytes on the wire: the inner-iteration messages carry a camera vector, the linearisation group (one per LM iteration) 90 Nc scalars
    # a rank pushes only the rows of the cameras it holds (contributor masks): the busiest rank's count sets the wire time
    held = float(Nc) if args.unfused else float(has.sum(1).max())
    wire = (max(0.0, c - 1.0) * 9 * held * w + 90 * held * w) / (args.link_gbs * 1e9)
    if args.unfused or args.l8_large_us is None or args.l8_small_us is None:
        t8 = tmax + c_kernel * (l8 - lat1) * 1e-6 + c * args.hop_us * 1e-6 + wire
        priced = "every kernel collective at the camera-vector latency"
    else:
        # fused form: what keeps a kernel of its own are messages of a few scalars (the closing dots of a solve that ran into its cap)
        t8 = tmax + c_kernel * (args.l8_small_us - lat1_small) * 1e-6 + c * args.hop_us * 1e-6 + wire
        priced = "%.2f small messages at L8_small per LM iteration (inner iterations and the linearisation group are fused)" % c_kernel
    # phase-wise pricing: the ranks synchronise at every fused message, so the iteration costs the slowest rank of
    # each phase — max_r(linearise_r) + max_r(everything else_r) — not the slowest rank's total
    tphase = max(s["linearise_seconds_per_lm_iteration"] for s in shards) + max(s["other_seconds_per_lm_iteration"] for s in shards)
    t8_phase = t8 - tmax + tphase
    res = {"kind": "PROJECTION from one GPU (tools/shard_projection.py), not a multi-GPU measurement",
           "workload": f"{args.workload} {args.dtype}, block-Jacobi PCG, {args.pcg_iterations} fixed inner iterations, {args.world} landmark shards",
           "T1_seconds_per_lm_iteration": t1, "T1_lm_iterations_per_sec": 1.0 / t1, "shards": shards,
           "max_shard_seconds_per_lm_iteration": tmax, "collectives_per_lm_iteration": c,
           "form": "unfused (kernel per all-reduce)" if args.unfused else "inner-iteration message fused into operator / update, %d virtual ranks inside every shard run" % args.world,
           "collectives_with_a_kernel_of_their_own_per_lm_iteration": c_kernel, "fused_messages_per_lm_iteration": 0.0 if args.unfused else inner,
           "L1_us_one_rank_allreduce_camera_vector": lat1, "L8_us_measured_between_%d_processes_on_one_gpu" % args.world: l8,
           "xgmi_hop_us_assumed_per_message": args.hop_us, "xgmi_link_gbs_assumed": args.link_gbs, "wire_seconds_per_lm_iteration": wire, "cameras_held_by_the_busiest_rank": held, "contributors_per_camera_mean": float(has.sum(0).mean()), "camera_vector_kb": msg_kb, "kernel_collectives_priced_as": priced,
           "L1_us_large_small": [lat1_large, lat1_small], "L8_us_large_small_measured": [args.l8_large_us, args.l8_small_us],
           "projected_T8_seconds_per_lm_iteration": t8, "projected_lm_iterations_per_sec": 1.0 / t8, "projected_speedup": t1 / t8,
           "slowest_rank_per_phase_seconds_per_lm_iteration": tphase, "projected_T8_phasewise_seconds_per_lm_iteration": t8_phase,
           "projected_speedup_phasewise": t1 / t8_phase,
           "phasewise_note": "T8 with max_r(linearise_r) + max_r(rest_r) in place of max_r(total_r): the figure to compare a SCALE record with",
           "speedup_if_collectives_were_free": t1 / (tmax - c_kernel * lat1 * 1e-6),
           "ideal": args.world}
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
