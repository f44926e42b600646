#!/usr/bin/env python3
"""bench.py — LM iterations/s of the MI355X hot path on a synthetic BAL-shaped problem.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--solver ...] [--dtype f64|f32] [--repeats R]

A "step" is one Levenberg-Marquardt iteration (solve + trial update + chi2 +
accept/relinearise or reject) of optimizer::levenberg_marquardt
(/root/reference/include/graphite/optimizer/levenberg_marquardt.hpp:166-240) over the whole
problem.  Default workload: BASELINE.json configs[2], BAL Ladybug-1723 shape
(1723 cameras, 156502 points, 678718 observations), fp64, block-Jacobi PCG
(PCGSolver + BlockJacobiPreconditioner, 10 inner iterations, tol 1.0, rejection 5.0 =
examples/bal.cu:296-309 defaults), lambda 1e-4 — the configuration the north star's
1-GPU target is quoted on.  Inputs are resident in HBM before the timed region.

The timed region (exactly K steps from the reset initial guess, barrier + synchronize on both sides, max over
ranks) is repeated R times (default 7); `value` is the MEDIAN repeat, min/max are printed beside it.  Every
number of the line (pcg_iterations, solve_seconds, pcg_gflops, chi2 trace) comes from that same median repeat.
`parity_rel` compares its chi2 trace with the CPU oracle's trace of the same algorithm (the cpu_baseline leg);
above 1e-6 (fp64) the bench FAILS (exit code 3).

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
PARITY_BAR = {"f64": 1e-6, "f32": 1e-4}  # north star: residuals 1e-6 relative (fp64); fp32: SURVEY 8(d)'s starting tolerance
MFMA_KERNELS = ("chol_syrk", "chol_trsm", "chol_syrk_col", "chol_potrf", "spchol_update", "spchol_trsm", "spchol_potrf")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=7)
    ap.add_argument("--workload", default="ladybug-1723")
    ap.add_argument("--solver", default=None, choices=[None, "pcg", "pcg-schur", "pcg-schur-implicit", "dense-schur"])
    ap.add_argument("--dtype", default=None, choices=[None, "f32", "f64"])
    ap.add_argument("--pcg-iterations", type=int, default=10)
    ap.add_argument("--pcg-tol", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--parity-only", action="store_true", help="cpu_baseline: only the sequential same-algorithm oracle leg (the parity gate), no further CPU timing legs")
    ap.add_argument("--cpu-baseline-iters", type=int, default=2)  # direct-solve legs: ~6 s (Schur) / ~12 s (full H) per iteration
    ap.add_argument("--dump-kernels", default=None, help="write per-kernel HIP-event table to this JSON file")
    ap.add_argument("--pmc-traffic", default="auto", choices=["auto", "off"],
                    help="auto (N = 1): measure roofline.traffic live — two child passes of this workload under `rocprofv3 --pmc` "
                         "(FETCH_SIZE, WRITE_SIZE) after the timed region; off: the committed profiles/pmc_traffic.json is looked up")
    return ap.parse_args()


DEFAULTS = {  # workload -> (solver, dtype) as BASELINE.json configs name them
    "ladybug-49": ("pcg-schur", "f32"),
    "ladybug-1723": ("pcg", "f64"),
    "venice-1778": ("pcg", "f32"),
    "final-13682": ("pcg", "f64"),
}


def effective_cores():
    """host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes
    expose 256 logical CPUs behind a 16-CPU quota; 256 OpenMP threads there run 30x slower than 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def median_index(xs):
    order = sorted(range(len(xs)), key=lambda i: xs[i])
    return order[len(order) // 2]


def self_launch(n):
    """`python3 bench.py --gpus N` started plainly (no WORLD_SIZE): start the N ranks as a CHILD torch.distributed.run
    (never an exec: nothing here has touched the GPU yet, and it must stay that way in this parent), relay its output and
    return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd).returncode


def measure_pmc_traffic(args, solver_name, dtype_name, kernel):
    """roofline.traffic measured BY THIS RUN: two short child passes of the same workload under rocprofv3 (one counter per pass, as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE counts 64 B per 128-B request on gfx950, WRITE_SIZE is KB of 64-B writes), median over
    the dominant kernel's launches.  Children, never an exec; any failure returns None and the committed table is looked up instead."""
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None  # this process is itself being profiled: no profiler inside a profiler
    med = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="gr_pmc_", dir="/tmp")
        try:
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--workload", args.workload, "--solver", solver_name, "--dtype", dtype_name, "--pcg-iterations", str(args.pcg_iterations),
                   "--pcg-tol", str(args.pcg_tol), "--no-cpu-baseline", "--no-also", "--repeats", "1", "--steps", "10", "--warmup", "2", "--pmc-traffic", "off"]
            r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return None
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and ("k_" + kernel) in row["Kernel_Name"].split("(")[0]:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None
            med[counter] = statistics.median(vals)
        except Exception:
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return {"hbm_bytes": 2 * med["FETCH_SIZE"] * 1024 + med["WRITE_SIZE"] * 1024, "FETCH_SIZE_KB": med["FETCH_SIZE"], "WRITE_SIZE_KB": med["WRITE_SIZE"]}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # GR_BENCH_SHARE_GPU=1 (testing only): every rank uses GPU 0, the process group is gloo and all all-reduces go through
    # the IPC mailboxes (RCCL cannot put two ranks on one device) — exercises the N > 1 code path of this file on a
    # 1-GPU box; the numbers it prints are not a scaling measurement
    share_gpu = os.environ.get("GR_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # GR_BENCH_FORCE_COMM=1 (testing): run the sharded code path (process group, RCCL communicator, all-reduces)
    # even with a single rank, e.g. under `python -m torch.distributed.run --nproc-per-node 1`
    sharded = world > 1 or os.environ.get("GR_BENCH_FORCE_COMM") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import graphite_amd as ga
    from graphite_amd import synth

    SOLVERS = {"pcg": ga.SOLVER_PCG, "pcg-schur": ga.SOLVER_PCG_SCHUR, "pcg-schur-implicit": ga.SOLVER_PCG_SCHUR_IMPLICIT,
               "dense-schur": ga.SOLVER_DENSE_SCHUR}

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not sharded:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    transport = {"kind": None}

    def make_engine(workload, dtype):
        prob = synth.make_config(workload)
        t0 = time.perf_counter()
        if sharded:
            from graphite_amd import dist as gdist
            part = gdist.partition_by_landmark(prob, rank, world, point_weight=gdist.point_weight_for(dtype))  # points by first camera, cut by observations + weighted points
            gpu = ga.BalProblem(part.cameras, part.points, part.obs, part.cam_idx, part.pt_idx, dtype=dtype,
                                device=local_rank, shard=True)
            # small all-reduces go peer to peer through IPC-mapped mailboxes (one hop over xGMI), RCCL carries what does
            # not fit a slot; the mailboxes are verified at start-up and every rank drops to RCCL together if that fails.
            # GR_COMM=rccl: RCCL for everything.
            if share_gpu:
                used = gdist.init_comm_ipc(gpu, rank, world, slot_bytes=16 << 20, rccl_fallback=False)
                transport["kind"] = "ipc-mailbox (shared GPU, test mode)"
            elif os.environ.get("GR_COMM", "ipc") == "rccl":
                gdist.init_comm(gpu, rank, world)
                transport["kind"] = "rccl"
            else:
                used = gdist.init_comm_ipc(gpu, rank, world, slot_bytes=4 << 20, rccl_fallback=True)
                transport["kind"] = "ipc-mailbox+rccl" if used else "rccl (ipc verification failed)"
        else:
            part = prob
            gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype, device=local_rank)
        torch.cuda.synchronize()
        return prob, part, gpu, time.perf_counter() - t0

    def timed_runs(gpu, part, steps, warmup, repeats, lm_kw):
        """W untimed LM iterations, then `repeats` x (reset, barrier, time exactly `steps` iterations, barrier)."""
        if warmup > 0:
            gpu.levenberg_marquardt(iterations=warmup, **lm_kw)
        runs = []
        for _ in range(max(1, repeats)):
            gpu.set_params(part.cameras, part.points)
            barrier()
            t0 = time.perf_counter()
            ct, lt, st = gpu.levenberg_marquardt(iterations=steps, profile=False, **lm_kw)
            barrier()
            dt = max_over_ranks(time.perf_counter() - t0)
            runs.append((dt, ct, lt, st))
        return runs

    def summarise(runs):
        dts = [r[0] for r in runs]
        m = median_index(dts)
        dt, ct, lt, st = runs[m]
        steps_run = st["iterations_run"]
        return dict(dt=dt, ct=ct, lt=lt, st=st, steps_run=steps_run, value=steps_run / dt,
                    value_min=steps_run / max(dts), value_max=steps_run / min(dts))

    dsolver, ddtype = DEFAULTS.get(args.workload, ("pcg", "f64"))
    solver_name = args.solver or dsolver
    dtype_name = args.dtype or ddtype
    dtype = np.float64 if dtype_name == "f64" else np.float32
    solver = SOLVERS[solver_name]

    prob, part, gpu, create_seconds = make_engine(args.workload, dtype)
    Nc, Np, No = prob.shape
    lm_kw = dict(solver=solver, initial_damping=1e-4, pcg_max_iter=args.pcg_iterations, pcg_tol=args.pcg_tol, pcg_rej=5.0)

    runs = timed_runs(gpu, part, args.steps, args.warmup, args.repeats, lm_kw)
    main_run = summarise(runs)
    st, ct = main_run["st"], main_run["ct"]

    # one more identical pass with HIP events around every hot kernel (kept out of `value`): the roofline leg
    gpu.set_params(part.cameras, part.points)
    barrier()
    _, _, st_prof = gpu.levenberg_marquardt(iterations=args.steps, profile=True, **lm_kw)
    barrier()
    ks = gpu.kernel_stats()
    if main_run["st"]["solve_seconds"] <= 0:
        # host-driven LM forms (Schur solvers, landmark shards) time the solve with HIP events only when profiling is on (an
        # event between two launches is a ~6 us bubble): solve_seconds then comes from this profiled pass, not from `value`'s
        main_run["st"]["solve_seconds"] = st_prof["solve_seconds"]
        main_run["st"]["solve_seconds_from"] = "profiled pass"

    n = 9 * Nc + 3 * Np
    w = np.dtype(dtype).itemsize
    # PCG GFLOP/s with the reference's flop count for one matrix-free iteration
    flops_pcg_iter = 104.0 * No + 2 * (81 * Nc + 9 * Np) + 12 * n
    PCG_CEILING_GFLOPS = {8: 3100.0, 4: 6200.0}[w]  # 0.39 flop/B (fp64) at 8 TB/s; fp32 moves half the bytes

    def pcg_gflops_of(s):
        if solver_name != "pcg" or s["solve_seconds"] <= 0:
            return None
        return flops_pcg_iter * s["pcg_iterations"] / s["solve_seconds"] / 1e9

    # ---- second line: the same workload with pcg_tol = 0, i.e. every solve runs its 10 inner iterations --------
    fixed = None
    if solver_name == "pcg" and args.pcg_tol != 0.0 and not args.no_also:
        kw0 = dict(lm_kw, pcg_tol=0.0)
        r0 = summarise(timed_runs(gpu, part, args.steps, 1, min(args.repeats, 5), kw0))
        g0 = flops_pcg_iter * r0["st"]["pcg_iterations"] / r0["st"]["solve_seconds"] / 1e9 if r0["st"]["solve_seconds"] > 0 else None
        fixed = {"workload": f"same, pcg_tol = 0 ({args.pcg_iterations} fixed inner iterations per solve unless the rejection test fires)",
                 "value": round(r0["value"], 2), "value_min": round(r0["value_min"], 2), "value_max": round(r0["value_max"], 2),
                 "unit": "LM iterations/s", "steps_run": r0["steps_run"], "pcg_iterations": r0["st"]["pcg_iterations"],
                 "solve_seconds": round(r0["st"]["solve_seconds"], 6),
                 "pcg_gflops": None if g0 is None else round(g0, 1), "pcg_gflops_hbm_ceiling": PCG_CEILING_GFLOPS,
                 "pcg_frac_of_ceiling": None if g0 is None else round(g0 / PCG_CEILING_GFLOPS, 4),
                 "us_per_pcg_iteration": round(r0["st"]["solve_seconds"] / max(r0["st"]["pcg_iterations"], 1) * 1e6, 2),
                 "chi2_final": float(r0["ct"][-1])}

    venice, final_mixed = None, None
    default_line = args.workload == "ladybug-1723" and args.solver is None and args.dtype is None and not args.no_also
    if (sharded and args.workload != "venice-1778" and not args.no_also) or (not sharded and default_line):
        # the configuration the north star's 8-GPU target is quoted on (BASELINE.json configs[3]): Venice-1778 fp32 — on every N,
        # the N = 1 line included, so that its strong scaling can be read off the lines of one SCALE run
        gpu.close()
        vprob, vpart, vgpu, _ = make_engine("venice-1778", np.float32)
        vkw = dict(solver=ga.SOLVER_PCG, initial_damping=1e-4, pcg_max_iter=10, pcg_tol=1.0, pcg_rej=5.0)
        rv = summarise(timed_runs(vgpu, vpart, args.steps, 3, min(args.repeats, 5), vkw))
        vNc, vNp, vNo = vprob.shape
        par = f"landmark-sharded x{world}" if sharded else "single GPU"
        venice = {"workload": f"BAL venice-1778 shape ({vNc} cameras, {vNp} points, {vNo} observations), pcg, f32, {par}",
                  "value": round(rv["value"], 2), "value_min": round(rv["value_min"], 2), "value_max": round(rv["value_max"], 2),
                  "unit": "LM iterations/s", "steps_run": rv["steps_run"], "accepted_steps": rv["st"]["accepted"],
                  "pcg_iterations": rv["st"]["pcg_iterations"], "ms_per_step": round(rv["dt"] / max(rv["steps_run"], 1) * 1e3, 4),
                  "collectives_per_lm_iteration": round(rv["st"]["collectives"] / max(rv["steps_run"], 1), 2),
                  "chi2_initial": float(rv["ct"][0]), "chi2_final": float(rv["ct"][-1])}
        vgpu.close()
        del vprob, vpart
        if os.environ.get("GR_BENCH_FINAL") == "1":
            # BASELINE.json configs[4]: Final-13682, fp32 Jacobian entries + fp64 PCG (the reference's FP64-FP32 mode), 3 LM iterations.
            # Opt-in (GR_BENCH_FINAL=1, any N): every rank synthesises the 29 M observations (~1 minute), which the default line,
            # bound to finish within minutes on every N, does not spend
            fprob, fpart, fgpu, _ = make_engine("final-13682", np.float64)
            fgpu.set_jacobian_precision(np.float32)
            rf = summarise(timed_runs(fgpu, fpart, 3, 1, 3, vkw))
            fNc, fNp, fNo = fprob.shape
            final_mixed = {"workload": f"BAL final-13682 shape ({fNc} cameras, {fNp} points, {fNo} observations), pcg, fp32 Jacobians + fp64 PCG, {par}",
                           "value": round(rf["value"], 2), "value_min": round(rf["value_min"], 2), "value_max": round(rf["value_max"], 2),
                           "unit": "LM iterations/s", "steps_run": rf["steps_run"], "accepted_steps": rf["st"]["accepted"],
                           "pcg_iterations": rf["st"]["pcg_iterations"], "ms_per_step": round(rf["dt"] / max(rf["steps_run"], 1) * 1e3, 4),
                           "collectives_per_lm_iteration": round(rf["st"]["collectives"] / max(rf["steps_run"], 1), 2),
                           "chi2_initial": float(rf["ct"][0]), "chi2_final": float(rf["ct"][-1])}
            fgpu.close()
            del fprob, fpart

    if rank != 0:
        if sharded:
            dist.destroy_process_group()
        return

    steps_run = main_run["steps_run"]
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
        if name in MFMA_KERNELS:
            # the reduced-camera Cholesky (dense tiles or the nested-dissection sparse form) is the one MFMA-bound stage: v_mfma_f64_16x16x4_f64 /
            # v_mfma_f32_16x16x4_f32 run at the vector rate (MI355X_MICROARCH.md: 157.3 TF fp32, 78.6 TF fp64)
            bound, unit = "mfma", "TFLOP/s"
            peak = 78.6 if w == 8 else 157.3
            achieved = k["flops_per_launch"] / avg_s / 1e12
        roofline = {"bound": bound, "kernel": name, "achieved": round(achieved, 2), "peak": peak,
                    "unit": unit, "frac": round(achieved / peak, 5), "traffic": None,
                    "avg_launch_us": round(avg_s * 1e6, 3), "launches": k["launches"], "active_launches": active,
                    "algorithmic_bytes_per_launch": k["bytes_per_launch"],
                    "flops_per_launch": k["flops_per_launch"],
                    "achieved_gflops": round(k["flops_per_launch"] / avg_s / 1e9, 1),
                    "kernels": {nm: {"avg_us": round(v["total_ms"] * 1e3 / max(v.get("active_launches", v["launches"]), 1), 2),
                                     "active_launches": v.get("active_launches", v["launches"]),
                                     "frac_of_hbm_peak": round(v["bytes_per_launch"] / (v["total_ms"] * 1e-3 / max(v.get("active_launches", v["launches"]), 1)) / 1e9 / HBM_PEAK_GBS, 4) if v["total_ms"] > 0 else None}
                                for nm, v in ks.items()}}
    pcg_gflops = pcg_gflops_of(st)

    # ---- CPU baseline + parity against the oracle ---------------------------------------------------------------
    cpu, parity_rel, parity_steps = None, None, 0
    if not args.no_cpu_baseline and world > 1 and rank == 0:
        # N > 1: no CPU timing legs (they belong to the N = 1 line), but the parity check of the sharded run's trace
        # against the oracle of the FULL problem is kept: it is what shows that the shards solve the same problem
        import oracle
        osolver = {"pcg": oracle.SOLVER_PCG, "pcg-schur": oracle.SOLVER_PCG_SCHUR, "pcg-schur-implicit": oracle.SOLVER_PCG_SCHUR,
                   "dense-schur": oracle.SOLVER_LDLT_SCHUR}[solver_name]
        per_it = {"pcg": 0.7, "pcg-schur": 2.5, "pcg-schur-implicit": 2.5, "dense-schur": 6.0}[solver_name] * (No / 678718.0)
        parity_steps = int(max(2, min(steps_run, 12, 10.0 / per_it)))
        ref = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
        ct_r, _, _ = ref.levenberg_marquardt(solver=osolver, iterations=parity_steps, initial_damping=1e-4,
                                             pcg_max_iter=args.pcg_iterations, pcg_tol=args.pcg_tol, pcg_rej=5.0)
        m = min(len(ct_r), len(ct))
        parity_rel = float(np.max(np.abs(np.asarray(ct[:m]) - ct_r[:m]) / np.abs(ct_r[:m])))
        del ref
    if not args.no_cpu_baseline and world == 1:
        import oracle
        nproc = effective_cores()
        sample_tag = f"{args.workload} {dtype_name}"
        # (a) like for like: the same algorithm as the GPU line (matrix-free block-Jacobi PCG / PCG on S / LDL^T of S),
        #     sequential oracle: doubles as the parity reference of the timed GPU trace
        osolver = {"pcg": oracle.SOLVER_PCG, "pcg-schur": oracle.SOLVER_PCG_SCHUR, "pcg-schur-implicit": oracle.SOLVER_PCG_SCHUR,
                   "dense-schur": oracle.SOLVER_LDLT_SCHUR}[solver_name]
        per_it = {"pcg": 0.7, "pcg-schur": 2.5, "pcg-schur-implicit": 2.5, "dense-schur": 6.0}[solver_name] * (No / 678718.0)
        parity_steps = int(max(2, min(steps_run, 12, 10.0 / per_it)))
        ref = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
        ct_r, _, pst = ref.levenberg_marquardt(solver=osolver, iterations=parity_steps, initial_damping=1e-4,
                                               pcg_max_iter=args.pcg_iterations, pcg_tol=args.pcg_tol, pcg_rej=5.0)
        m = min(len(ct_r), len(ct))
        parity_rel = float(np.max(np.abs(np.asarray(ct[:m]) - ct_r[:m]) / np.abs(ct_r[:m])))
        same_1 = {"value": round(pst["iterations_run"] / pst["loop_seconds"], 5), "unit": "LM iterations/s", "cores": 1, "kind": "port",
                  "sample": f"{parity_steps} LM iterations of {sample_tag}, oracle restatement of the same solver ({solver_name}), sequential",
                  "seconds": round(pst["loop_seconds"], 3)}
        del ref
        base = oracle.CpuBaseline(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
        it = args.cpu_baseline_iters

        def leg(solver_kind, iters, threads, ordering):
            base.reset()
            c, _, s, tm = base.levenberg_marquardt(solver_kind, iters, threads=threads, ordering=ordering, initial_damping=1e-4,
                                                   pcg_max_iter=args.pcg_iterations, pcg_tol=args.pcg_tol)
            return c, s, tm

        small_enough = Nc <= 4000 and not args.parity_only  # the simplicial LDL^T of a 16 k x 16 k banded S takes seconds; a dense 123 k one does not finish
        cpu = {}
        if small_enough:
            # (b) the reference's "eigen_solver CPU path": eigen-schur = assembly + Schur reduction (GPU in the reference; here all
            #     host cores) + ONE-thread simplicial LDL^T (src/eigen_solver.cpp:10-29), minimum-degree ordering (Eigen: AMD)
            c_s, s_s, t_s = leg(oracle.SOLVER_LDLT_SCHUR, it, nproc, 1)
            ldlt_s = (t_s["ldlt_factor"] + t_s["ldlt_solve"]) / max(s_s["iterations_run"], 1)
            cpu = {"value": round(s_s["iterations_run"] / s_s["loop_seconds"], 5), "unit": "LM iterations/s", "cores": int(t_s["threads"]), "kind": "port",
                   "sample": f"{it} LM iterations of {sample_tag}: restatement of the reference's eigen-schur path "
                             f"(EigenSchurLDLTSolver): linearise + Hessian + Schur reduction OpenMP on {int(t_s['threads'])} host cores, "
                             "simplicial LDL^T of S on ONE thread as src/eigen_solver.cpp:21-29, minimum-degree camera order (Eigen: AMD)",
                   "seconds": round(s_s["loop_seconds"], 3),
                   "ldlt_only_seconds_per_iteration": round(ldlt_s, 4),
                   "ldlt_only_lm_iterations_per_sec": round(1.0 / ldlt_s, 4),
                   "ldlt_only_note": "factorise + solve of S alone, one thread: the part the reference runs on the CPU; an upper bound on its "
                                     "LM rate on this host however fast its GPU stages are",
                   "stage_seconds": {k: round(v, 4) for k, v in t_s.items() if k not in ("threads", "ldlt_nnz")},
                   "ldlt_nnz": int(t_s["ldlt_nnz"]), "chi2_final": float(c_s[-1])}
            # (c) eigen = the FULL system H (EigenLDLTSolver, solver/eigen.hpp:71-98), points first then cameras by minimum degree
            c_f, s_f, t_f = leg(oracle.SOLVER_LDLT, max(1, it // 2), nproc, 1)
            ldlt_f = (t_f["ldlt_factor"] + t_f["ldlt_solve"]) / max(s_f["iterations_run"], 1)
            cpu["full_h_ldlt"] = {"value": round(s_f["iterations_run"] / s_f["loop_seconds"], 5), "unit": "LM iterations/s", "cores": int(t_f["threads"]),
                                  "kind": "port", "sample": f"{max(1, it // 2)} LM iterations, EigenLDLTSolver restatement (full H, upper CSC, "
                                  "points then minimum-degree cameras), assembly on all cores, LDL^T on one thread",
                                  "seconds": round(s_f["loop_seconds"], 3), "ldlt_only_seconds_per_iteration": round(ldlt_f, 4),
                                  "ldlt_nnz": int(t_f["ldlt_nnz"])}
            # (d) the same eigen-schur leg with the oracle's own RCM camera order, for the ordering's effect
            _, s_r, t_r = leg(oracle.SOLVER_LDLT_SCHUR, 1, nproc, 0)
            cpu["eigen_schur_rcm_order"] = {"ldlt_only_seconds_per_iteration": round(t_r["ldlt_factor"] + t_r["ldlt_solve"], 4), "ldlt_nnz": int(t_r["ldlt_nnz"])}
        # (e) all host cores on the GPU line's own algorithm (matrix-free block-Jacobi PCG)
        if solver_name == "pcg" and not args.parity_only:
            pit = int(max(4, min(16, 12.0 / (0.35 * No / 678718.0))))
            c_p, s_p, t_p = leg(oracle.SOLVER_PCG, pit, nproc, 1)
            allc = {"value": round(s_p["iterations_run"] / s_p["loop_seconds"], 5), "unit": "LM iterations/s", "cores": int(t_p["threads"]), "kind": "port",
                    "sample": f"{pit} LM iterations of {sample_tag}, matrix-free block-Jacobi PCG, every stage OpenMP on {int(t_p['threads'])} cores",
                    "seconds": round(s_p["loop_seconds"], 3), "chi2_final": float(c_p[-1])}
            if not cpu:
                cpu = dict(allc)
            cpu["same_algorithm_all_cores"] = allc
        if not cpu:
            cpu = dict(same_1)
        cpu["same_algorithm"] = same_1
        cpu["host_cores"] = nproc
        cpu["host_cores_note"] = f"usable cores = affinity mask capped by the cgroup CPU quota ({os.cpu_count()} logical CPUs visible)"

    # PMC traffic of the dominant kernel, measured offline with the same command under
    # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) and committed under profiles/
    live = None
    if roofline and world == 1 and args.pmc_traffic == "auto" and roofline["bound"] == "hbm":
        live = measure_pmc_traffic(args, solver_name, dtype_name, roofline["kernel"])
    if roofline and live:
        roofline["traffic"] = live["hbm_bytes"]
        roofline["traffic_source"] = ("measured by this run: two child passes of the same workload under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE; "
                                      "10 LM iterations each), median over the kernel's launches")
        roofline["traffic_detail"] = live
        roofline["traffic_over_algorithmic"] = round(live["hbm_bytes"] / roofline["algorithmic_bytes_per_launch"], 3)
        roofline["traffic_note"] = "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)"
    elif roofline:
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = f"{args.workload} {dtype_name} {solver_name}"
            kk = [k for k in tr.get(key, {}) if k.startswith("k_" + roofline["kernel"])]
            if kk:
                roofline["traffic"] = tr[key][kk[0]]["hbm_bytes"]
                roofline["traffic_source"] = "profiles/pmc_traffic.json (offline rocprofv3 --pmc passes of this command; not measured by this run)"
                roofline["traffic_note"] = tr.get("note")
        except Exception:
            pass
        # the reference ALGORITHM streams stored Jacobians: SURVEY §8(d) bytes per matrix-free PCG iteration
        if solver_name == "pcg":
            ref_bytes = No * (24 * w + 8) + 14 * n * w + (81 * Nc + 9 * Np) * w
            roofline["reference_algorithm_bytes_per_pcg_iteration"] = ref_bytes
            roofline["reference_algorithm_time_at_peak_us"] = round(ref_bytes / HBM_PEAK_GBS / 1e3, 2)
            # second roofline entry, in the survey's own unit: ONE WHOLE matrix-free PCG iteration (operator + update + direction)
            # against SURVEY 8(d)'s bytes for it (the reference algorithm with stored Jacobians read once), timed on the
            # fixed-iteration `also` run where every solve runs all its inner iterations
            if fixed and fixed.get("us_per_pcg_iteration"):
                ach = ref_bytes / (fixed["us_per_pcg_iteration"] * 1e-6) / 1e9
                roofline["reference_equivalent_pcg_iteration"] = {
                    "reference_equivalent_GBs": round(ach, 2), "reference_equivalent_frac_of_hbm_peak": round(ach / HBM_PEAK_GBS, 5),
                    "us_per_iteration": fixed["us_per_pcg_iteration"], "reference_algorithm_bytes_per_iteration": ref_bytes,
                    "note": "NOT a measured bandwidth: SURVEY 8(d) bytes of the reference algorithm's PCG iteration (stored Jacobians) / "
                            "measured device time per iteration of this implementation, which recomputes J and moves fewer bytes"}

    also = []
    if fixed:
        also.append(fixed)
    if venice:
        also.append(venice)
    if final_mixed:
        also.append(final_mixed)
    if world == 1 and args.workload == "ladybug-1723" and args.solver is None and args.dtype is None and not args.no_also:
        # BASELINE.json configs[1] next to the default configs[2]: Ladybug-49 fp32, Schur + PCG
        p49 = synth.make_config("ladybug-49")
        g49 = ga.BalProblem(p49.cameras, p49.points, p49.obs, p49.cam_idx, p49.pt_idx, dtype=np.float32, device=local_rank)
        kw49 = dict(solver=ga.SOLVER_PCG_SCHUR, initial_damping=1e-4, pcg_max_iter=10, pcg_tol=1.0, pcg_rej=5.0)
        r49 = summarise(timed_runs(g49, p49, args.steps, 3, min(args.repeats, 5), kw49))
        also.append({"workload": "BAL ladybug-49 shape (49 cameras, 7776 points, 31843 observations), pcg-schur, f32",
                     "value": round(r49["value"], 2), "value_min": round(r49["value_min"], 2), "value_max": round(r49["value_max"], 2),
                     "unit": "LM iterations/s", "steps_run": r49["steps_run"],
                     "ms_per_step": round(r49["dt"] / max(r49["steps_run"], 1) * 1e3, 4), "chi2_initial": float(r49["ct"][0]),
                     "chi2_final": float(r49["ct"][-1])})
        g49.close()

    line = {
        "metric": "lm_iterations_per_sec", "value": round(main_run["value"], 4), "unit": "LM iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(main_run["dt"] / max(steps_run, 1) * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": f"BAL {args.workload} shape ({Nc} cameras, {Np} points, {No} observations), "
                               f"{solver_name}, {args.pcg_iterations} inner iterations, tol {args.pcg_tol:g}, lambda 1e-4",
                   "solver": solver_name, "parallelism": f"landmark-sharded x{world}" if world > 1 else "single GPU"},
        "repeats": len(runs), "value_min": round(main_run["value_min"], 2), "value_max": round(main_run["value_max"], 2),
        "value_note": "median of `repeats` timed regions of exactly `steps` LM iterations each, every one from the reset initial guess",
        "steps_run": steps_run, "accepted_steps": st["accepted"], "pcg_iterations": st["pcg_iterations"],
        "pcg_gflops": None if pcg_gflops is None else round(pcg_gflops, 2),
        "pcg_gflops_note": "reference flop count per matrix-free iteration (SURVEY 8d) x inner iterations / device time inside solve, same timed region as `value`",
        "chi2_initial": float(ct[0]), "chi2_final": float(ct[-1]), "mse_final": float(ct[-1]) / No,
        "solve_seconds": round(st["solve_seconds"], 6), "loop_seconds": round(st["loop_seconds"], 6),
        "setup_seconds": round(st["setup_seconds"], 6), "create_seconds": round(create_seconds, 4),
        "setup_note": "create_seconds: gr_bal_create (orderings, upload); setup_seconds: solver structure + first linearisation inside levenberg_marquardt; both outside `value`",
        "transport": transport["kind"],
        "collectives_per_lm_iteration": round(st.get("collectives", 0) / max(steps_run, 1), 2) if sharded else 0,
        "parity_rel": parity_rel, "parity_steps": parity_steps,
        "parity_note": "max relative difference of the timed run's chi2 trace against the CPU oracle's trace of the same solver",
        "roofline": roofline, "cpu_baseline": cpu, "also": also or None,
    }
    if args.dump_kernels:
        with open(args.dump_kernels, "w") as f:
            json.dump({"kernels": ks, "line": line}, f, indent=1)
    print(json.dumps(line))
    if sharded:
        dist.destroy_process_group()
    if parity_rel is not None and not (parity_rel < PARITY_BAR[dtype_name]):
        sys.stderr.write(f"bench.py: PARITY FAILURE: chi2 trace differs from the oracle by {parity_rel:.3e} (bar {PARITY_BAR[dtype_name]:g})\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
