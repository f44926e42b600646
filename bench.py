#!/usr/bin/env python3
"""bench.py — LM iterations/s of the MI355X hot path on a synthetic BAL-shaped problem.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--solver pcg|pcg-schur] [--dtype f64|f32]

A "step" is one Levenberg-Marquardt iteration (solve + trial update + chi2 +
accept/relinearise or reject) of optimizer::levenberg_marquardt
(/root/reference/include/graphite/optimizer/levenberg_marquardt.hpp:166-240) over the whole
problem.  Default workload: BASELINE.json configs[2], BAL Ladybug-1723 shape
(1723 cameras, 156502 points, 678718 observations), fp64, block-Jacobi PCG
(PCGSolver + BlockJacobiPreconditioner, 10 inner iterations, tol 1.0, rejection 5.0 =
examples/bal.cu:296-309 defaults), lambda 1e-4 — the configuration the north star's
1-GPU target is quoted on.  Inputs are resident in HBM before the timed region.

N > 1: one process per GPU (torch.distributed.run); the factor graph is sharded by
landmark range, cameras replicated, camera-space sums all-reduced with RCCL inside
libgraphite_mi355x.so.  The SAME problem is split over the ranks, so scaling = "strong".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ladybug-1723")
    ap.add_argument("--solver", default=None, choices=[None, "pcg", "pcg-schur", "pcg-schur-implicit", "dense-schur"])
    ap.add_argument("--dtype", default=None, choices=[None, "f32", "f64"])
    ap.add_argument("--pcg-iterations", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-iters", type=int, default=6)  # ~12 s of host work on Ladybug-1723
    ap.add_argument("--dump-kernels", default=None, help="write per-kernel HIP-event table to this JSON file")
    return ap.parse_args()


DEFAULTS = {  # workload -> (solver, dtype) as BASELINE.json configs name them
    "ladybug-49": ("pcg-schur", "f32"),
    "ladybug-1723": ("pcg", "f64"),
    "venice-1778": ("pcg-schur-implicit", "f32"),  # same iterates as pcg-schur, S never formed (3.5x faster here)
    "final-13682": ("pcg", "f64"),
}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # GR_BENCH_FORCE_COMM=1 (testing): run the sharded code path (process group, RCCL communicator, all-reduces)
    # even with a single rank, e.g. under `python -m torch.distributed.run --nproc-per-node 1`
    sharded = world > 1 or os.environ.get("GR_BENCH_FORCE_COMM") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import graphite_amd as ga
    from graphite_amd import synth

    dsolver, ddtype = DEFAULTS.get(args.workload, ("pcg", "f64"))
    solver_name = args.solver or dsolver
    dtype_name = args.dtype or ddtype
    dtype = np.float64 if dtype_name == "f64" else np.float32
    solver = {"pcg": ga.SOLVER_PCG, "pcg-schur": ga.SOLVER_PCG_SCHUR, "pcg-schur-implicit": ga.SOLVER_PCG_SCHUR_IMPLICIT, "dense-schur": ga.SOLVER_DENSE_SCHUR}[solver_name]

    prob = synth.make_config(args.workload)
    Nc, Np, No = prob.shape
    if sharded:
        from graphite_amd import dist as gdist
        part = gdist.partition_by_landmark(prob, rank, world)
        gpu = ga.BalProblem(part.cameras, part.points, part.obs, part.cam_idx, part.pt_idx, dtype=dtype,
                            device=local_rank, shard=True)
        gdist.init_comm(gpu, rank, world)
    else:
        part = prob
        gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype,
                            device=local_rank)

    lm_kw = dict(solver=solver, initial_damping=1e-4, pcg_max_iter=args.pcg_iterations, pcg_tol=1.0, pcg_rej=5.0)

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup: W untimed LM iterations, then restart from the initial guess
    if args.warmup > 0:
        gpu.levenberg_marquardt(iterations=args.warmup, **lm_kw)
    gpu.set_params(part.cameras, part.points)

    barrier()
    t0 = time.perf_counter()
    ct, lt, st = gpu.levenberg_marquardt(iterations=args.steps, profile=False, **lm_kw)
    barrier()
    dt = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # second, identical pass with HIP events around every hot kernel (kept out of `value`)
    gpu.set_params(part.cameras, part.points)
    barrier()
    ct2, _, st2 = gpu.levenberg_marquardt(iterations=args.steps, profile=True, **lm_kw)
    barrier()
    ks = gpu.kernel_stats()

    if rank != 0:
        if sharded:
            dist.destroy_process_group()
        return

    steps_run = st["iterations_run"]
    w = np.dtype(dtype).itemsize
    dominant = max(ks.items(), key=lambda kv: kv[1]["total_ms"]) if ks else None
    roofline = None
    if dominant:
        name, k = dominant
        # look-ahead launches that found the PCG loop finished return at once and move nothing: their
        # (small) time is charged to the active launches, their bytes are not counted
        active = max(k.get("active_launches", k["launches"]), 1)
        avg_s = k["total_ms"] * 1e-3 / active
        achieved = k["bytes_per_launch"] / avg_s / 1e9
        bound, peak, unit = "hbm", HBM_PEAK_GBS, "GB/s"
        if name in ("chol_syrk", "chol_trsm", "chol_syrk_col"):
            # the dense reduced-camera Cholesky is the one MFMA-bound stage: v_mfma_f64_16x16x4_f64 /
            # v_mfma_f32_16x16x4_f32 run at the vector rate (MI355X_MICROARCH.md: 157.3 TF fp32, 78.6 TF fp64)
            bound, unit = "mfma", "TFLOP/s"
            peak = 78.6 if w == 8 else 157.3
            achieved = k["flops_per_launch"] / avg_s / 1e12
        roofline = {"bound": bound, "kernel": name, "achieved": round(achieved, 2), "peak": peak,
                    "unit": unit, "frac": round(achieved / peak, 5), "traffic": None,
                    "avg_launch_us": round(avg_s * 1e6, 3), "launches": k["launches"], "active_launches": active,
                    "algorithmic_bytes_per_launch": k["bytes_per_launch"],
                    "flops_per_launch": k["flops_per_launch"],
                    "achieved_gflops": round(k["flops_per_launch"] / avg_s / 1e9, 1)}
    # PCG GFLOP/s with the reference's flop count for one matrix-free iteration
    n = 9 * Nc + 3 * Np
    flops_pcg_iter = (104.0 * No + 2 * (81 * Nc + 9 * Np) + 12 * n) if solver_name == "pcg" else None
    pcg_gflops = None
    # inner-solve time is measured in the profiled pass (the timed pass replays whole LM iterations as hipGraphs,
    # which have no per-solve events)
    if flops_pcg_iter and st2["solve_seconds"] > 0:
        pcg_gflops = flops_pcg_iter * st2["pcg_iterations"] / st2["solve_seconds"] / 1e9

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        import oracle
        ref = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
        it = args.cpu_baseline_iters
        _, _, cst = ref.levenberg_marquardt(solver=oracle.SOLVER_LDLT_SCHUR, iterations=it, initial_damping=1e-4)
        cpu = {"value": round(cst["iterations_run"] / cst["loop_seconds"], 5), "unit": "LM iterations/s",
               "cores": 1, "kind": "port",
               "sample": f"{it} LM iterations of the same workload ({args.workload} {dtype_name}), CPU restatement "
                         "of the reference's eigen-schur path (all stages on one host core: linearise, Schur, "
                         "simplicial LDL^T; the reference runs only the LDL^T on the CPU)",
               "seconds": round(cst["loop_seconds"], 3)}
        if solver_name == "pcg":
            # like for like: the same block-Jacobi PCG algorithm (10 inner iterations, tol 1.0) on the same core
            _, _, pst = ref.levenberg_marquardt(solver=oracle.SOLVER_PCG, iterations=2 * it, initial_damping=1e-4)
            cpu["same_algorithm"] = {"value": round(pst["iterations_run"] / pst["loop_seconds"], 5), "unit": "LM iterations/s",
                                     "cores": 1, "kind": "port", "sample": f"{2 * it} LM iterations, matrix-free block-Jacobi PCG",
                                     "seconds": round(pst["loop_seconds"], 3)}

    # PMC traffic of the dominant kernel, measured offline with the same command under
    # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) and committed under profiles/
    if roofline:
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = f"{args.workload} {dtype_name} {solver_name}"
            kk = [k for k in tr.get(key, {}) if k.startswith("k_" + roofline["kernel"])]
            if kk:
                roofline["traffic"] = tr[key][kk[0]]["hbm_bytes"]
                roofline["traffic_note"] = tr.get("note")
        except Exception:
            pass
        # the reference ALGORITHM streams stored Jacobians: SURVEY §8(d) bytes per matrix-free PCG iteration
        if solver_name == "pcg":
            ref_bytes = No * (24 * w + 8) + 14 * n * w + (81 * Nc + 9 * Np) * w
            roofline["reference_algorithm_bytes_per_pcg_iteration"] = ref_bytes
            roofline["reference_algorithm_time_at_peak_us"] = round(ref_bytes / HBM_PEAK_GBS / 1e3, 2)

    also = None
    if world == 1 and args.workload == "ladybug-1723" and args.solver is None and args.dtype is None:
        # BASELINE.json configs[1] next to the default configs[2]: Ladybug-49 fp32, Schur + PCG
        p49 = synth.make_config("ladybug-49")
        g49 = ga.BalProblem(p49.cameras, p49.points, p49.obs, p49.cam_idx, p49.pt_idx, dtype=np.float32, device=local_rank)
        kw49 = dict(solver=ga.SOLVER_PCG_SCHUR, initial_damping=1e-4, pcg_max_iter=10, pcg_tol=1.0, pcg_rej=5.0)
        g49.levenberg_marquardt(iterations=3, **kw49)
        g49.set_params(p49.cameras, p49.points)
        torch.cuda.synchronize()
        t49 = time.perf_counter()
        c49, _, s49 = g49.levenberg_marquardt(iterations=args.steps, **kw49)
        torch.cuda.synchronize()
        t49 = time.perf_counter() - t49
        also = {"workload": "BAL ladybug-49 shape (49 cameras, 7776 points, 31843 observations), pcg-schur, f32",
                "value": round(s49["iterations_run"] / t49, 2), "unit": "LM iterations/s", "steps_run": s49["iterations_run"],
                "ms_per_step": round(t49 / max(s49["iterations_run"], 1) * 1e3, 4), "chi2_initial": float(c49[0]),
                "chi2_final": float(c49[-1])}
        g49.close()

    line = {
        "metric": "lm_iterations_per_sec", "value": round(steps_run / dt, 4), "unit": "LM iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / max(steps_run, 1) * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": f"BAL {args.workload} shape ({Nc} cameras, {Np} points, {No} observations), "
                               f"{solver_name}, {args.pcg_iterations} inner iterations, lambda 1e-4",
                   "solver": solver_name, "parallelism": f"landmark-sharded x{world}" if world > 1 else "single GPU"},
        "steps_run": steps_run, "accepted_steps": st["accepted"], "pcg_iterations": st["pcg_iterations"],
        "pcg_gflops": None if pcg_gflops is None else round(pcg_gflops, 2),
        "chi2_initial": float(ct[0]), "chi2_final": float(ct[-1]), "mse_final": float(ct[-1]) / No,
        "solve_seconds": round(st["solve_seconds"], 6), "loop_seconds": round(st["loop_seconds"], 6),
        "roofline": roofline, "cpu_baseline": cpu, "also": also,
    }
    if args.dump_kernels:
        with open(args.dump_kernels, "w") as f:
            json.dump({"kernels": ks, "line": line}, f, indent=1)
    print(json.dumps(line))
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
