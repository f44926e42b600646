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


def pmc_pass(args, solver_name, dtype_name, kernel, counters):
    """one short child pass of the same workload under `rocprofv3 --pmc <counters>` (a child, never an exec): median of every counter
    over the dominant kernel's launches, or None"""
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
    out = tempfile.mkdtemp(prefix="gr_pmc_", dir="/tmp")
    try:
        cmd = [prof, "--pmc", *counters, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
               "--workload", args.workload, "--solver", solver_name, "--dtype", dtype_name, "--pcg-iterations", str(args.pcg_iterations),
               "--pcg-tol", str(args.pcg_tol), "--no-cpu-baseline", "--no-also", "--repeats", "1", "--steps", "10", "--warmup", "2", "--pmc-traffic", "off"]
        r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", capture_output=True, text=True, timeout=240)
        if r.returncode != 0:
            return None
        vals = {}
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if ("k_" + kernel) in row["Kernel_Name"].split("(")[0]:
                    vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        if not vals:
            return None
        return {k: statistics.median(v) for k, v in vals.items()}
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def measure_pmc_traffic(args, solver_name, dtype_name, kernel):
    """roofline.traffic measured BY THIS RUN: two short child passes of the same workload under rocprofv3 (one counter per pass, as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE counts 64 B per 128-B request on gfx950, WRITE_SIZE is KB of 64-B writes), median over
    the dominant kernel's launches.  Any failure returns None and the committed table is looked up instead."""
    med = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        m = pmc_pass(args, solver_name, dtype_name, kernel, [counter])
        if not m or counter not in m:
            return None
        med[counter] = m[counter]
    return {"hbm_bytes": 2 * med["FETCH_SIZE"] * 1024 + med["WRITE_SIZE"] * 1024, "FETCH_SIZE_KB": med["FETCH_SIZE"], "WRITE_SIZE_KB": med["WRITE_SIZE"]}


def measure_sq(args, solver_name, dtype_name, kernel):
    """what the dominant kernel's waves do with their time: one child pass with the SQ activity counters
    (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, disjoint) and one each for rocprofv3's derived
    VALUBusy / OccupancyPercent (gfx94x formulas on gfx950)"""
    raw = pmc_pass(args, solver_name, dtype_name, kernel, ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"])
    if not raw or not raw.get("SQ_WAVE_CYCLES"):
        return None
    wc = raw["SQ_WAVE_CYCLES"]
    out = {"mem_wait": round(raw.get("SQ_WAIT_ANY", 0.0) / wc, 4), "issue_stall": round(raw.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
           "issuing": round(raw.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
           "valu_share_of_issued": round(raw.get("SQ_ACTIVE_INST_VALU", 0.0) / max(raw.get("SQ_ACTIVE_INST_ANY", 0.0), 1.0), 4),
           "waves_per_launch": raw.get("SQ_WAVES"),
           "definitions": "fractions of the kernel's wave-cycles (SQ_WAVE_CYCLES): mem_wait = SQ_WAIT_ANY (waves parked on s_waitcnt / barrier), "
                          "issue_stall = SQ_WAIT_INST_ANY, issuing = SQ_ACTIVE_INST_ANY; valu_share_of_issued = SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_ANY",
           "raw": {k: raw[k] for k in sorted(raw)}}
    for name, key in (("VALUBusy", "valu_busy"), ("OccupancyPercent", "occupancy_percent")):
        d = pmc_pass(args, solver_name, dtype_name, kernel, [name])
        out[key] = None if not d or name not in d else round(d[name], 3)
    if out.get("occupancy_percent") is not None:
        out["waves_per_simd"] = round(out["occupancy_percent"] / 100.0 * 8.0, 2)  # 8 wave slots per SIMD
    return out


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
                # (gr_bal_tuning.comm_transport / GR_COMM_TRANSPORT: 0 = the same hand-shake but RCCL only, 2 = the peer mapping treated
                # as refused — both end with every rank on RCCL, the path a node without peer access takes)
                used = gdist.init_comm_ipc(gpu, rank, world, slot_bytes=4 << 20, rccl_fallback=True)
                transport["kind"] = "ipc-mailbox+rccl" if used else "rccl (mailboxes not used: refused, failed verification, or comm_transport = 0)"
        else:
            part = prob
            gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype, device=local_rank)
        torch.cuda.synchronize()
        return prob, part, gpu, time.perf_counter() - t0

    def timed_runs(gpu, part, steps, warmup, repeats, lm_kw):
        """W untimed LM iterations, then `repeats` x (reset, barrier, time exactly `steps` iterations, barrier)."""
        if warmup > 0:
            gpu.levenberg_marquardt(iterations=warmup, **lm_kw)
        if os.environ.get("GR_TEST_KILL_RANK") == str(rank):
            # fault injection (tests/test_gpu_ipc.py): this rank dies after the warm-up, its peers are inside collectives that wait for
            # it — bounded waits (gr_bal_tuning.ipc_timeout_ms) end them with GR_ERR_COMM, torch.distributed.run ends the job: non-zero exit
            os._exit(17)
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

    def config_cached(workload):
        """synth.make_config through a /tmp cache keyed by the workload name (= shape + seed): the 29 M observations of
        final-13682 take about a minute to synthesise"""
        path = f"/tmp/graphite_synth_{workload}_v1.npz"
        if workload in ("venice-1778", "final-13682") and os.path.exists(path):
            try:
                z = np.load(path)
                return synth.BalProblem(z["cameras"], z["points"], z["obs"], z["cam_idx"], z["pt_idx"], workload)
            except Exception:
                pass
        prob = synth.make_config(workload)
        if workload in ("venice-1778", "final-13682") and rank == 0:
            try:
                np.savez(path + ".tmp.npz", cameras=prob.cameras, points=prob.points, obs=prob.obs, cam_idx=prob.cam_idx, pt_idx=prob.pt_idx)
                os.replace(path + ".tmp.npz", path)
            except Exception:
                pass
        return prob

    def roofline_of(ks, itemsize):
        """dominant kernel of a profiled pass: algorithmic bytes (or flops) per launch / its mean ACTIVE launch"""
        if not ks:
            return None
        name, k = max(ks.items(), key=lambda kv: kv[1]["total_ms"])
        active = max(k.get("active_launches", k["launches"]), 1)
        avg_s = k["total_ms"] * 1e-3 / active
        achieved = k["bytes_per_launch"] / avg_s / 1e9
        bound, peak, unit = "hbm", HBM_PEAK_GBS, "GB/s"
        if name in MFMA_KERNELS:
            bound, unit = "mfma", "TFLOP/s"
            peak = 78.6 if itemsize == 8 else 157.3
            achieved = k["flops_per_launch"] / avg_s / 1e12
        return {"bound": bound, "kernel": name, "achieved": round(achieved, 2), "peak": peak, "unit": unit, "frac": round(achieved / peak, 5),
                "traffic": None, "avg_launch_us": round(avg_s * 1e6, 3), "launches": k["launches"], "active_launches": active,
                "algorithmic_bytes_per_launch": k["bytes_per_launch"], "flops_per_launch": k["flops_per_launch"]}

    def cache_residency(working_set_bytes):
        """where the working set of one LM iteration lives — 32 MiB of L2 (8 x 4 MiB, not coherent across XCDs) and
        the 256 MiB Infinity Cache, whose hits rocprofv3's FETCH_SIZE counts like HBM fetches (MI355X_MICROARCH.md)"""
        l2, ic = 32 * 2 ** 20, 256 * 2 ** 20
        where = "L2" if working_set_bytes <= l2 else ("Infinity Cache" if working_set_bytes <= ic else "HBM")
        return {"working_set_bytes": int(working_set_bytes), "l2_bytes": l2, "infinity_cache_bytes": ic,
                "working_set_over_l2": round(working_set_bytes / l2, 3), "working_set_over_infinity_cache": round(working_set_bytes / ic, 3),
                "served_from": where,
                "note": "arrays one LM iteration touches (observations, index streams, per-observation records, vectors, blocks); "
                        + ("they fit the Infinity Cache: `traffic` is then mostly cache-served, and `frac` of the 8 TB/s HBM peak is a "
                           "lower bound on how close the kernel is to what actually feeds it" if where != "HBM" else
                           "beyond the Infinity Cache: `traffic` is HBM traffic")}

    def latency_roofline(ks, ra, residency, hbm_record):
        """A workload whose LM iteration touches less than the L2s hold and whose kernels are all a few launch floors long is bound by
        the NUMBER of dependent launches and in-launch grid barriers, not by bytes: an HBM fraction of 0.005 says nothing.
        The record then carries the launch-floor model instead: floor = launches per LM iteration x 4.6 us (a launch that finds nothing
        to do: profiles/HISTORY.md, kernel trace) + grid barriers per LM iteration x 2.3 us (the cooperative PCG's flagged-record rendezvous; 3.0 us for the counter barrier of the resident matrix-free PCG; measured),
        frac = floor / measured time per LM iteration; `hbm` keeps the byte view of the dominant kernel for reference."""
        if residency["served_from"] != "L2" or not ks:
            return None
        steps_run = max(ra["steps_run"], 1)
        per_iter_us = ra["dt"] / steps_run * 1e6
        launches = sum(v["launches"] for v in ks.values())
        longest = max(v["total_ms"] * 1e3 / max(v.get("active_launches", v["launches"]), 1) for v in ks.values())
        if longest > 40.0:
            return None
        LAUNCH_FLOOR_US, BARRIER_US = 4.6, (2.3 if "schur_pcg_coop" in ks else 3.0)  # (round 6: the cooperative PCG on S meets through flagged records, 2.3 us measured)
        inner = ra["st"]["pcg_iterations"] / steps_run
        coop = ks.get("schur_pcg_coop") or ks.get("pcg_resident")
        # the cooperative PCG on S: one barrier after its start, then three per inner iteration; the resident matrix-free PCG: two per inner iteration
        barriers = 0.0
        if "schur_pcg_coop" in ks:
            barriers = (1.0 + 3.0 * inner) * ks["schur_pcg_coop"]["launches"] / steps_run
        elif "pcg_resident" in ks:
            barriers = 2.0 * inner * ks["pcg_resident"]["launches"] / steps_run
        per_iter_launches = launches / steps_run
        floor_us = per_iter_launches * LAUNCH_FLOOR_US + barriers * BARRIER_US
        kernel_us = sum(v["total_ms"] for v in ks.values()) * 1e3 / steps_run
        return {"bound": "latency", "kernel": hbm_record["kernel"], "unit": "us per LM iteration", "achieved": round(per_iter_us, 2), "peak": round(floor_us, 2),
                "frac": round(floor_us / per_iter_us, 4), "traffic": None,
                "launches_per_lm_iteration": round(per_iter_launches, 2), "launch_floor_us": LAUNCH_FLOOR_US,
                "grid_barriers_per_lm_iteration": round(barriers, 2), "grid_barrier_us": BARRIER_US,
                "kernel_time_us_per_lm_iteration": round(kernel_us, 2), "longest_kernel_us": round(longest, 2),
                "note": "latency-bound: the working set sits in the L2s and every kernel is a few launch floors long; peak = launches x floor + "
                        "barriers x barrier cost (the time a perfect implementation of THIS launch structure would take), frac = peak / measured",
                "hbm": {k: hbm_record[k] for k in ("achieved", "peak", "frac", "avg_launch_us", "algorithmic_bytes_per_launch") if k in hbm_record}}

    def working_set(nc, npts, nobs, itemsize, nseg_est=None):
        nn = 9 * nc + 3 * npts
        nseg_est = nseg_est or (nobs / 64 + nc)
        return nobs * (2 * itemsize + 12) + nobs * 11 * itemsize + 63 * nseg_est * itemsize + 12 * nn * itemsize + (24 * nc + 2 * (81 * nc + 9 * npts)) * itemsize

    def also_entry(workload, dt, solver_key, label, steps, warmup, repeats, jac32=False, parity_iters=0):
        """one more BASELINE config on this box: value + parity_rel of the timed trace against the oracle + its dominant kernel's roofline"""
        aprob = config_cached(workload)
        t0 = time.perf_counter()
        if sharded:
            from graphite_amd import dist as gdist
            apart = gdist.partition_by_landmark(aprob, rank, world, point_weight=gdist.point_weight_for(dt))
            agpu = ga.BalProblem(apart.cameras, apart.points, apart.obs, apart.cam_idx, apart.pt_idx, dtype=dt, device=local_rank, shard=True)
            if share_gpu:
                gdist.init_comm_ipc(agpu, rank, world, slot_bytes=16 << 20, rccl_fallback=False)
            elif os.environ.get("GR_COMM", "ipc") == "rccl":
                gdist.init_comm(agpu, rank, world)
            else:
                gdist.init_comm_ipc(agpu, rank, world, slot_bytes=4 << 20, rccl_fallback=True)
        else:
            apart = aprob
            agpu = ga.BalProblem(aprob.cameras, aprob.points, aprob.obs, aprob.cam_idx, aprob.pt_idx, dtype=dt, device=local_rank)
        if jac32:
            agpu.set_jacobian_precision(np.float32)
        akw = dict(solver=SOLVERS[solver_key], initial_damping=1e-4, pcg_max_iter=10, pcg_tol=1.0, pcg_rej=5.0)
        ra = summarise(timed_runs(agpu, apart, steps, warmup, repeats, akw))
        agpu.set_params(apart.cameras, apart.points)
        barrier()
        agpu.levenberg_marquardt(iterations=steps, profile=True, **akw)
        barrier()
        aks = agpu.kernel_stats()
        agpu.close()
        aNc, aNp, aNo = aprob.shape
        isz = np.dtype(dt).itemsize
        par = f"landmark-sharded x{world}" if sharded else "single GPU"
        e = {"workload": f"BAL {workload} shape ({aNc} cameras, {aNp} points, {aNo} observations), {label}, {par}",
             "value": round(ra["value"], 2), "value_min": round(ra["value_min"], 2), "value_max": round(ra["value_max"], 2),
             "unit": "LM iterations/s", "steps_run": ra["steps_run"], "accepted_steps": ra["st"]["accepted"],
             "pcg_iterations": ra["st"]["pcg_iterations"], "ms_per_step": round(ra["dt"] / max(ra["steps_run"], 1) * 1e3, 4),
             "collectives_per_lm_iteration": round(ra["st"].get("collectives", 0) / max(ra["steps_run"], 1), 2),
             "chi2_initial": float(ra["ct"][0]), "chi2_final": float(ra["ct"][-1]), "parity_rel": None, "roofline": roofline_of(aks, isz)}
        if e["roofline"]:
            res = cache_residency(working_set(aNc, aNp, aNo, isz))
            lat = latency_roofline(aks, ra, res, e["roofline"])
            if lat:
                e["roofline"] = lat
            e["roofline"]["cache_residency"] = res
        if parity_iters > 0 and rank == 0 and not args.no_cpu_baseline:
            import oracle
            osolver = {"pcg": oracle.SOLVER_PCG, "pcg-schur": oracle.SOLVER_PCG_SCHUR, "pcg-schur-implicit": oracle.SOLVER_PCG_SCHUR,
                       "dense-schur": oracle.SOLVER_LDLT_SCHUR}[solver_key]
            m = min(parity_iters, ra["steps_run"])
            # the oracle runs in the graph precision; the fp32-Jacobian mode is compared with the fp64 oracle (its bar is fp32's)
            ref = oracle.BalOracle(aprob.cameras, aprob.points, aprob.obs, aprob.cam_idx, aprob.pt_idx, dtype=dt)
            ct_r, _, _ = ref.levenberg_marquardt(solver=osolver, iterations=m, initial_damping=1e-4, pcg_max_iter=10, pcg_tol=1.0, pcg_rej=5.0)
            k = min(len(ct_r), len(ra["ct"]))
            e["parity_rel"] = float(np.max(np.abs(np.asarray(ra["ct"][:k]) - ct_r[:k]) / np.abs(ct_r[:k])))
            e["parity_steps"] = k - 1
            e["parity_bar"] = PARITY_BAR["f32" if (isz == 4 or jac32) else "f64"]
            del ref
        del aprob, apart
        return e

    def pose_graph_entry(steps):
        """The pose-graph engine (include/graphite/engine_pose.hpp): a 10 000-pose / 48 593-factor SE(2) graph of the generic C++ API — one vertex
        descriptor, binary between-factors with dense 3 x 3 information matrices, one fixed pose — through tests/cpp/test_pose_graph.hip
        (a hipcc-compiled client; its binary travels in build/), PCGSolver + block-Jacobi, 10 inner iterations.  value = LM iterations /
        seconds of the optimiser's own per-iteration clock (the table's Time column, second call of the process); the generic kernels on
        the same command beside it; parity of the chi2 trace against oracle/pose_graph.py."""
        import subprocess
        exe = os.path.join(ROOT, "build", "test_pose_graph")
        if not os.path.exists(exe):
            return None
        n = 10000
        p0, fx, e, m, info, _ = synth.make_pose_graph(n)
        path = "/tmp/graphite_bench_pose_graph_10k.txt"
        synth.write_pose_graph(path, p0, fx, e, m, info, huber_delta=0.0)
        cmd = [exe, path, "pcg", str(steps), "manual", "10", "1.0"]
        env = dict(os.environ, POSE_REPEAT="2")
        env.pop("GRAPHITE_GENERIC_ONLY", None); env.pop("GRAPHITE_POSE_ENGINE", None)
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            g = subprocess.run(cmd, env=dict(env, GRAPHITE_GENERIC_ONLY="1"), capture_output=True, text=True, timeout=600)
        except Exception:
            return None
        if r.returncode != 0 or g.returncode != 0 or "POSE_ENGINE_HANDOVERS 2" not in r.stdout:
            return None

        def table(out):
            rows = [ln.split() for ln in out.splitlines() if len(ln.split()) == 6 and ln.split()[0].isdigit()]
            rows = rows[-steps:]  # the second call
            return np.array([[float(x) for x in f[1:5]] for f in rows])
        te, tg = table(r.stdout), table(g.stdout)
        if len(te) < 2 or len(tg) < 2:
            return None
        wall = lambda out: float([ln for ln in out.splitlines() if ln.startswith("LM_SECONDS")][0].split()[1])
        it_e, it_g = float(np.median(te[1:, 3])), float(np.median(tg[1:, 3]))
        ent = {"workload": f"pose-graph engine: {n} SE(2) poses, {len(e)} between-factors with dense 3 x 3 information matrices (generic-API graph, one vertex descriptor; "
                           "engine_pose.hpp kernels on the user's traits), pcg + block-Jacobi, 10 inner iterations, f64, single GPU",
               "value": round(1.0 / it_e, 2), "unit": "LM iterations/s", "steps_run": len(te), "ms_per_step": round(it_e * 1e3, 4),
               "value_note": "median of the optimiser's per-iteration clock (iterations 1..) in the second call of the process; whole_call_ms is the wall clock of that call, set-up included",
               "whole_call_ms": round(wall(r.stdout) * 1e3, 3), "chi2_initial": float(te[0, 0]), "chi2_final": float(te[-1, 1]),
               "generic_kernels": {"value": round(1.0 / it_g, 2), "unit": "LM iterations/s", "ms_per_step": round(it_g * 1e3, 4), "whole_call_ms": round(wall(g.stdout) * 1e3, 3)},
               "speedup_vs_generic_kernels": round(it_g / it_e, 2), "parity_rel": None,
               "roofline": {"bound": "latency", "kernel": "k_pe_solve", "unit": "us per LM iteration", "achieved": round(it_e * 1e6, 2),
                            "peak": round(2 * 4.6 + 21 * 2.3, 2), "frac": round((2 * 4.6 + 21 * 2.3) / (it_e * 1e6), 4), "traffic": None,
                            "launches_per_lm_iteration": 2, "launch_floor_us": 4.6, "grid_rendezvous_per_lm_iteration": 21, "grid_rendezvous_us": 2.3,
                            "note": "latency-bound: 2 launches and 1 + 2 x 10 grid-wide rendezvous (one write-through store + one polling load each, measured 2.3 us) "
                                    "per LM iteration; peak = the time this launch structure would take with nothing else on the chain, frac = peak / measured"}}
        # the same graph through EigenLDLTSolver: the block-sparse Hessian on the library's nested-dissection tile Cholesky (gr_spchol)
        dsteps = 5
        try:
            d = subprocess.run([exe, path, "eigen", str(dsteps), "manual", "10", "1.0"], env=env, capture_output=True, text=True, timeout=600)
        except Exception:
            d = None
        if d is not None and d.returncode == 0 and "SPARSE_FACTORISATION 1" in d.stdout:
            td = table(d.stdout)[-dsteps:]
            ent["direct_solver"] = {"solver": "EigenLDLTSolver -> gr_spchol (tile-sparse nested-dissection Cholesky, block size 3)", "ms_per_step": round(float(np.median(td[1:, 3])) * 1e3, 4),
                                    "value": round(1.0 / float(np.median(td[1:, 3])), 2), "unit": "LM iterations/s", "steps_run": len(td), "chi2_final": float(td[-1, 1]), "parity_rel": None}
        if not args.no_cpu_baseline:
            from oracle.pose_graph import PoseGraphOracle
            if "direct_solver" in ent:
                od = PoseGraphOracle(p0, fx, e, m, info)
                cd, _, _ = od.levenberg_marquardt(iterations=dsteps, direct=True)
                kd = min(len(td), len(cd) - 1)
                ent["direct_solver"]["parity_rel"] = float(np.max(np.abs(td[:kd, 1] - cd[1:kd + 1]) / np.abs(cd[1:kd + 1])))
            o = PoseGraphOracle(p0, fx, e, m, info)
            ct, _, _ = o.levenberg_marquardt(iterations=steps, pcg_max_iter=10, pcg_tol=1.0)
            k = min(len(te), len(ct) - 1)
            ent["parity_rel"] = float(np.max(np.abs(te[:k, 1] - ct[1:k + 1]) / np.abs(ct[1:k + 1])))
            ent["parity_bar"] = 1e-8
            ent["parity_note"] = "max relative difference of the chi2 trace against oracle/pose_graph.py (numpy restatement of the generic pipeline)"
        return ent

    def user_traits_entry(steps):
        """The engine's kernels instantiated on USER traits (include/graphite/engine_model.hpp): the Ladybug-1723 shape as a graph of
        the generic C++ API whose factors carry per-factor information matrices and per-factor Huber deltas — not the built-in camera
        model — through tests/cpp/test_engine_model.hip (a hipcc-compiled client; its binary travels in build/).  value = LM iterations
        / seconds inside the engine's loop of a second call on the cached problem; roofline from a GR_PROFILE_KERNELS pass; parity
        against the oracle with the same per-factor tables."""
        import subprocess
        exe = os.path.join(ROOT, "build", "test_engine_model")
        src = os.path.join(ROOT, "tests", "cpp", "test_engine_model.hip")
        if not os.path.exists(exe):
            hipcc = "/opt/rocm/bin/hipcc"
            if not os.path.exists(hipcc):
                return None
            os.makedirs(os.path.dirname(exe), exist_ok=True)
            lib = os.path.join(ROOT, "graphite_amd")
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O2", f"-I{ROOT}/include", src, f"-L{lib}", "-lgraphite_mi355x", f"-Wl,-rpath,{lib}", "-o", exe],
                               capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                return None
        uprob = synth.make_config("ladybug-1723")
        path = "/tmp/graphite_bench_ladybug1723.txt"
        synth.write_bal(path, uprob)
        uprob = synth.read_bal(path)  # the text round trip is what the client sees
        env = dict(os.environ, GRAPHITE_ENGINE="model")
        env.pop("GRAPHITE_GENERIC_ONLY", None)
        cmd = [exe, path, "pcg", str(steps), "weighted", "stored", "fp64", "twice"]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            rp = subprocess.run(cmd[:-1], env=dict(env, GR_PROFILE_KERNELS="1"), capture_output=True, text=True, timeout=600)
        except Exception:
            return None
        if r.returncode != 0 or rp.returncode != 0 or "ENGINE_MODEL_HANDOVERS 1" not in r.stdout:
            return None
        trace, second, uks = [], None, {}
        for ln in r.stdout.splitlines():
            f = ln.split()
            if len(f) == 6 and f[0].isdigit():
                trace.append((float(f[1]), float(f[2])))
            if ln.startswith("SECOND_CALL_SECONDS"):
                second = dict(zip(f[0::2], f[1::2]))
        for ln in rp.stdout.splitlines():
            f = ln.split()
            if f and f[0] == "KERNEL":
                uks[f[1]] = dict(launches=int(f[3]), active_launches=int(f[5]), total_ms=float(f[7]), bytes_per_launch=float(f[9]), flops_per_launch=float(f[11]))
        if not trace or not second:
            return None
        loop_s, its = float(second["LOOP_SECONDS"]), int(second["ITERATIONS"])
        uNc, uNp, uNo = uprob.shape
        e = {"workload": f"user-traits engine: BAL ladybug-1723 shape ({uNc} cameras, {uNp} points, {uNo} observations) as a generic-API graph with per-factor "
                         "2 x 2 information matrices and per-factor Huber deltas (engine_model.hpp kernels on the user's traits, stored weighted Jacobian), pcg, f64, single GPU",
             "value": round(its / loop_s, 2), "unit": "LM iterations/s", "steps_run": its, "ms_per_step": round(loop_s / max(its, 1) * 1e3, 4),
             "value_note": "second optimiser call on the cached engine problem: LM iterations / seconds inside the engine's loop",
             "hand_over_seconds_second_call": float(second["SETUP_SECONDS"]), "chi2_initial": trace[0][0], "chi2_final": trace[-1][1],
             "parity_rel": None, "roofline": roofline_of(uks, 8)}
        # the same graph with set_jacobian_storage(false): blocks recomputed through the user's jacobian<> in every operator launch
        # (ops/product.hpp:103,292); which of the two is faster depends on the model's cost and the inner iteration count
        try:
            rd = subprocess.run([exe, path, "pcg", str(steps), "weighted", "dynamic", "fp64", "twice"], env=env, capture_output=True, text=True, timeout=600)
            for ln in rd.stdout.splitlines():
                if ln.startswith("SECOND_CALL_SECONDS") and rd.returncode == 0:
                    f = ln.split()
                    sd = dict(zip(f[0::2], f[1::2]))
                    e["recomputed_jacobian"] = {"value": round(int(sd["ITERATIONS"]) / float(sd["LOOP_SECONDS"]), 2), "unit": "LM iterations/s",
                                                "ms_per_step": round(float(sd["LOOP_SECONDS"]) / max(int(sd["ITERATIONS"]), 1) * 1e3, 4)}
        except Exception:
            pass
        if e["roofline"]:
            e["roofline"]["kernels"] = {nm: {"avg_us": round(v["total_ms"] * 1e3 / max(v["active_launches"], 1), 2), "active_launches": v["active_launches"]} for nm, v in uks.items()}
            e["roofline"]["cache_residency"] = cache_residency(working_set(uNc, uNp, uNo, 8) + uNo * 24 * 8)
        if not args.no_cpu_baseline:
            import oracle
            f = np.arange(uNo)
            a_ = 0.5 + (f % 7) / 4.0
            c_ = 0.75 + (f % 5) / 8.0
            b_ = 0.25 * ((f % 3) - 1.0) * np.sqrt(a_ * c_)
            ref = oracle.BalOracle(uprob.cameras, uprob.points, uprob.obs, uprob.cam_idx, uprob.pt_idx, dtype=np.float64)
            ref.set_factor_tables(pmat=np.stack([a_, b_, b_, c_], axis=1), loss_kinds=np.full(uNo, oracle.LOSS_HUBER), loss_deltas=1.0 + (f % 4).astype(np.float64))
            m = min(6, len(trace))
            ct_r, _, _ = ref.levenberg_marquardt(solver=oracle.SOLVER_PCG, iterations=m, initial_damping=1e-4, pcg_max_iter=10, pcg_tol=1.0, pcg_rej=5.0)
            got = np.array([trace[0][0]] + [t[1] for t in trace[:m]])
            k = min(len(ct_r), len(got))
            e["parity_rel"] = float(np.max(np.abs(got[:k] - ct_r[:k]) / np.abs(ct_r[:k])))
            e["parity_steps"] = k - 1
            e["parity_bar"] = PARITY_BAR["f64"]
            del ref
        return e

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
    # ---- audit record of a sharded run: what every rank's communicator really is after the start-up self-test,
    # and per-phase times as the MAX over ranks per phase (a sharded iteration costs the slowest rank of each phase, not of the sum)
    shard_audit = None
    if sharded:
        ci = gpu.comm_info()
        lin_names = ("linearize", "linearize_hcp", "linearize_finalize", "finalize_bj", "chi2")
        phase = {"linearise_ms": sum(v["total_ms"] for k, v in ks.items() if k in lin_names),
                 "inner_iterations_ms": sum(v["total_ms"] for k, v in ks.items() if k not in lin_names)}
        infos = [None] * world
        # hipDeviceCanAccessPeer of this rank's device to every visible device (what the mailboxes need; counting devices does not
        # initialise them).  Ranks that share a device (test mode) see one device: [[True]]
        ndev = torch.cuda.device_count()
        peers = [bool(d == local_rank or torch.cuda.can_device_access_peer(local_rank, d)) for d in range(ndev)]
        dist.all_gather_object(infos, {"rank": rank, "device": ci["device"], "peer_access": peers, "transport": ci["transport_name"], "rccl_ranks": ci["rccl_ranks"],
                                       "mailboxes_opened": ci["mailboxes_opened"], "fused_agreed": ci["fused_agreed"],
                                       "oneshot_messages": ci["oneshot_messages"], "fallback_messages": ci["fallback_messages"], **phase,
                                       "observations": int(len(part.cam_idx)), "points": int(len(part.points))})
        shard_audit = {"ranks_seen": {"process_group": world, "rccl_comm_count": [i["rccl_ranks"] for i in infos],
                                      "mailboxes_opened_per_rank": [i["mailboxes_opened"] for i in infos]},
                       "devices": [i["device"] for i in infos], "transport_per_rank": [i["transport"] for i in infos],
                       "peer_access": {"matrix": [i["peer_access"] for i in infos], "note": "row r: hipDeviceCanAccessPeer(device of rank r, device d) for every visible device d"},
                       "fused_inner_iteration_message": [i["fused_agreed"] for i in infos],
                       "oneshot_messages_per_rank": [i["oneshot_messages"] for i in infos], "fallback_messages_per_rank": [i["fallback_messages"] for i in infos],
                       "per_phase_max_over_ranks_ms": {"linearise": max(i["linearise_ms"] for i in infos), "inner_iterations": max(i["inner_iterations_ms"] for i in infos),
                                                        "note": f"device time of the phase's kernels over the profiled pass of {args.steps} LM iterations, slowest rank PER PHASE"},
                       "per_rank_ms": [{"rank": i["rank"], "linearise": round(i["linearise_ms"], 3), "inner_iterations": round(i["inner_iterations_ms"], 3),
                                        "observations": i["observations"], "points": i["points"]} for i in infos]}
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
        venice = also_entry("venice-1778", np.float32, "pcg", "pcg, f32", args.steps, 3, min(args.repeats, 5), parity_iters=3 if world == 1 else 0)
        # BASELINE.json configs[4]: Final-13682, fp32 Jacobian entries + fp64 PCG (the reference's FP64-FP32 mode), 3 LM iterations.
        # Default at N = 1 when the box has the memory for it (the oracle's parity leg keeps ~12 GB of host arrays) and
        # GR_BENCH_FINAL != 0; N > 1: opt-in (GR_BENCH_FINAL=1) — every rank would synthesise / load the 29 M observations
        want_final = os.environ.get("GR_BENCH_FINAL")
        host_gb = 0.0
        try:
            host_gb = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2 ** 30
        except Exception:
            pass
        if want_final == "1" or (want_final is None and world == 1 and host_gb >= 48 and torch.cuda.mem_get_info()[0] >= 40 * 2 ** 30):
            final_mixed = also_entry("final-13682", np.float64, "pcg", "pcg, fp32 Jacobians + fp64 PCG", 3, 1, 3, jac32=True, parity_iters=2 if world == 1 else 0)

    if rank != 0:
        if sharded:
            dist.destroy_process_group()
        return

    steps_run = main_run["steps_run"]
    roofline = roofline_of(ks, w)
    if roofline:
        roofline["achieved_gflops"] = round(roofline["flops_per_launch"] / (roofline["avg_launch_us"] * 1e-6) / 1e9, 1)
        roofline["kernels"] = {nm: {"avg_us": round(v["total_ms"] * 1e3 / max(v.get("active_launches", v["launches"]), 1), 2),
                                    "active_launches": v.get("active_launches", v["launches"]),
                                    "frac_of_hbm_peak": round(v["bytes_per_launch"] / (v["total_ms"] * 1e-3 / max(v.get("active_launches", v["launches"]), 1)) / 1e9 / HBM_PEAK_GBS, 4) if v["total_ms"] > 0 else None}
                               for nm, v in ks.items()}
        roofline["cache_residency"] = cache_residency(working_set(Nc, Np, No, w))
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
            # ordering 2 = AMD of the factorised matrix (oracle/amd.hpp): Eigen::SimplicialLDLT's default, i.e. the reference's own choice
            c_s, s_s, t_s = leg(oracle.SOLVER_LDLT_SCHUR, it, nproc, 2)
            ldlt_s = (t_s["ldlt_factor"] + t_s["ldlt_solve"]) / max(s_s["iterations_run"], 1)
            cpu = {"value": round(s_s["iterations_run"] / s_s["loop_seconds"], 5), "unit": "LM iterations/s", "cores": int(t_s["threads"]), "kind": "port",
                   "sample": f"{it} LM iterations of {sample_tag}: restatement of the reference's eigen-schur path "
                             f"(EigenSchurLDLTSolver): linearise + Hessian + Schur reduction OpenMP on {int(t_s['threads'])} host cores, "
                             "simplicial LDL^T of S on ONE thread as src/eigen_solver.cpp:21-29, AMD ordering of S as Eigen::SimplicialLDLT's default does (oracle/amd.hpp)",
                   "seconds": round(s_s["loop_seconds"], 3),
                   "ldlt_only_seconds_per_iteration": round(ldlt_s, 4),
                   "ldlt_only_lm_iterations_per_sec": round(1.0 / ldlt_s, 4),
                   "ldlt_only_note": "factorise + solve of S alone, one thread: the part the reference runs on the CPU; an upper bound on its "
                                     "LM rate on this host however fast its GPU stages are",
                   "stage_seconds": {k: round(v, 4) for k, v in t_s.items() if k not in ("threads", "ldlt_nnz")},
                   "ldlt_nnz": int(t_s["ldlt_nnz"]), "chi2_final": float(c_s[-1])}
            # (c) eigen = the FULL system H (EigenLDLTSolver, solver/eigen.hpp:71-98), AMD ordering of H
            c_f, s_f, t_f = leg(oracle.SOLVER_LDLT, max(1, it // 2), nproc, 2)
            ldlt_f = (t_f["ldlt_factor"] + t_f["ldlt_solve"]) / max(s_f["iterations_run"], 1)
            cpu["full_h_ldlt"] = {"value": round(s_f["iterations_run"] / s_f["loop_seconds"], 5), "unit": "LM iterations/s", "cores": int(t_f["threads"]),
                                  "kind": "port", "sample": f"{max(1, it // 2)} LM iterations, EigenLDLTSolver restatement (full H, upper CSC, "
                                  "AMD ordering), assembly on all cores, LDL^T on one thread",
                                  "seconds": round(s_f["loop_seconds"], 3), "ldlt_only_seconds_per_iteration": round(ldlt_f, 4),
                                  "ldlt_nnz": int(t_f["ldlt_nnz"])}
            # (d) the same eigen-schur leg under the two other orderings, for the ordering's effect: exact minimum degree on the camera
            #     graph (what rounds 2-5 quoted) and the oracle's own reverse Cuthill-McKee
            _, s_m, t_m = leg(oracle.SOLVER_LDLT_SCHUR, 1, nproc, 1)
            cpu["eigen_schur_minimum_degree_order"] = {"ldlt_only_seconds_per_iteration": round(t_m["ldlt_factor"] + t_m["ldlt_solve"], 4), "ldlt_nnz": int(t_m["ldlt_nnz"])}
            _, s_r, t_r = leg(oracle.SOLVER_LDLT_SCHUR, 1, nproc, 0)
            cpu["eigen_schur_rcm_order"] = {"ldlt_only_seconds_per_iteration": round(t_r["ldlt_factor"] + t_r["ldlt_solve"], 4), "ldlt_nnz": int(t_r["ldlt_nnz"])}
            cpu["ldlt_nnz_by_ordering"] = {"amd": int(t_s["ldlt_nnz"]), "minimum_degree": int(t_m["ldlt_nnz"]), "rcm": int(t_r["ldlt_nnz"])}
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
        # the x2 is calibrated for wide coalesced streams only; profiles/r05_fetch_size_calibration.json (tools/fetch_calib.hip: 8 M records
        # read once each from a 4 GiB table) shows the counter tallies 64 B per REQUEST: a 64-byte gather is counted in full, a 192-byte
        # record as 128, a 16 B-per-lane stream at half.  Bounds, and an estimate that only doubles the coalesced share:
        F, Wr = live["FETCH_SIZE_KB"] * 1024.0, live["WRITE_SIZE_KB"] * 1024.0
        coalesced = float(No) * (2 * w + 12)  # observation + index streams of the per-observation kernels: read as whole 128-B requests
        if roofline["kernel"] in ("pcg_update", "pcg_direction"):
            coalesced = roofline["algorithmic_bytes_per_launch"] * 0.6  # vector kernels: reads are streams
        est = F + min(F, 0.5 * coalesced) + Wr
        roofline["traffic_calibrated"] = {"lower": F + Wr, "upper": 2 * F + Wr, "estimate": est,
                                          "estimate_over_algorithmic": round(est / roofline["algorithmic_bytes_per_launch"], 3),
                                          "estimate_GBs": round(est / (roofline["avg_launch_us"] * 1e-6) / 1e9, 1),
                                          "note": "lower: every request 64 B; upper: every request 128 B (the guide's x2); estimate: the kernel's coalesced "
                                                  "streams (observations + indices) doubled, its gathers counted as they are",
                                          "calibration": "profiles/r05_fetch_size_calibration.json"}
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
    if roofline and world == 1 and args.pmc_traffic == "auto":
        sq = measure_sq(args, solver_name, dtype_name, roofline["kernel"])
        if sq:
            roofline["sq"] = sq
            roofline["valu_busy"] = sq.get("valu_busy")
            roofline["mem_wait"] = sq["mem_wait"]
            roofline["waves_per_simd"] = sq.get("waves_per_simd")
            # `bound` keeps the contract's vocabulary (which roof the fraction is taken of); `limiter` says what the counters show
            if roofline["bound"] == "hbm":
                ic = roofline.get("cache_residency", {}).get("served_from")
                roofline["limiter"] = ("memory latency at low occupancy" if sq["mem_wait"] >= 0.45 else "instruction issue" if sq["issuing"] + sq["issue_stall"] >= 0.6 else "mixed") + \
                    f": waves parked on memory {sq['mem_wait']:.0%} of their cycles, issue stalls {sq['issue_stall']:.0%}, issuing {sq['issuing']:.0%} " \
                    f"({sq['valu_share_of_issued']:.0%} of it VALU); working set served from {ic}"
    # the reference ALGORITHM streams stored Jacobians: SURVEY §8(d) bytes per matrix-free PCG iteration
    if roofline and solver_name == "pcg":
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
        also.append(also_entry("ladybug-49", np.float32, "pcg-schur", "pcg-schur, f32", args.steps, 3, min(args.repeats, 5), parity_iters=6))
        ut = user_traits_entry(args.steps)
        if ut:
            also.append(ut)
        pg = pose_graph_entry(args.steps)
        if pg:
            also.append(pg)
        # the default workload under the two other inner solvers of the path, so that their numbers are the driver's too:
        # explicit Schur complement + PCG on S, and the direct solve of S (nested-dissection tile Cholesky on MFMA)
        also.append(also_entry("ladybug-1723", np.float64, "pcg-schur", "pcg-schur, f64", args.steps, 3, min(args.repeats, 3), parity_iters=2))
        also.append(also_entry("ladybug-1723", np.float64, "dense-schur", "dense-schur (direct solve of S), f64", args.steps, 3, min(args.repeats, 3), parity_iters=2))

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
        "transport": transport["kind"], "shard_audit": shard_audit,
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
    for e in also:
        if e.get("parity_rel") is not None and not (e["parity_rel"] < e.get("parity_bar", 1e-4)):
            sys.stderr.write(f"bench.py: PARITY FAILURE in `also` entry {e['workload'][:60]}...: {e['parity_rel']:.3e} (bar {e.get('parity_bar')})\n")
            sys.exit(3)
    if parity_rel is not None and not (parity_rel < PARITY_BAR[dtype_name]):
        sys.stderr.write(f"bench.py: PARITY FAILURE: chi2 trace differs from the oracle by {parity_rel:.3e} (bar {PARITY_BAR[dtype_name]:g})\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
