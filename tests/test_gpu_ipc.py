"""The one-shot peer all-reduce (csrc/comm.hpp IpcComm) between REAL processes.

Two fresh child processes share the one GPU of the box, map each other's mailbox through
hipIpcMemHandle (handles exchanged over a gloo process group) and run the landmark-sharded solve
with every all-reduce going through the mailboxes.  Checked against host sums of the same
vectors and against the unsharded solve in this process."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

import graphite_amd as ga
from graphite_amd import dist as gdist, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_workers(world, tmp_path, env_extra=None):
    port = free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    outs = [str(tmp_path / f"rank{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ipc_worker.py"), str(r), str(world), str(port), outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=240)[0])
    finally:
        for p in procs:          # exactly the children started above
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return [json.load(open(o)) for o in outs]


@pytest.mark.parametrize("world", [2, 3])
def test_ipc_allreduce_between_processes(world, tmp_path):
    slot = 1 << 16
    res = run_workers(world, tmp_path, {"GR_TEST_IPC_SLOT": str(slot)})
    # raw all-reduces: every rank holds the rank-order sum of what the ranks put in, bit for bit
    for k, n in enumerate((1, 2, 63, 64, 65, 1000, slot // 8)):
        want = np.zeros(n)
        for r in range(world):
            want = want + np.random.default_rng(1000 * n + r).standard_normal(n)
        for r in range(world):
            assert np.array_equal(np.array(res[r]["sums"][k]), want), (n, r)
    # a message beyond the slot without a fallback communicator is an error, not a silent truncation
    assert all(r["oversize_rc"] != 0 for r in res)

    prob = synth.make_config("mini-50")
    shards = [gdist.partition_by_landmark(prob, r, world) for r in range(world)]  # the cut every worker makes (tests/ipc_worker.py)
    for dtype, tag, solver, sname in ((np.float64, "f64", ga.SOLVER_PCG, "pcg"),
                                      (np.float64, "f64", ga.SOLVER_PCG, "pcg_unfused"),
                                      (np.float64, "f64", ga.SOLVER_PCG_SCHUR_IMPLICIT, "implicit"),
                                      (np.float32, "f32", ga.SOLVER_PCG, "pcg")):
        single = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
        ct, lt, st = single.levenberg_marquardt(solver=solver, iterations=8)
        c1, p1 = single.get_params()
        single.close()
        key = f"{tag}_{sname}"
        rt, at = (1e-9, 1e-7) if tag == "f64" else (1e-5, 2e-3)
        pts = gdist.assemble_points(shards, [np.array(res[r][key]["pts"]) for r in range(world)])
        for r in range(world):
            got = res[r][key]
            n = min(len(got["chi2"]), len(ct)) if tag == "f32" else len(ct)
            assert np.allclose(got["chi2"][:n], ct[:n], rtol=rt), (key, r, got["chi2"], ct)
            assert got["cams"] == res[0][key]["cams"]                    # replicas identical across PROCESSES
            assert got["collectives"] == res[0][key]["collectives"] > 0
            if tag == "f64":
                assert got["pcg_iterations"] == st["pcg_iterations"]
        if tag == "f64":
            assert np.allclose(np.array(res[0][key]["cams"]), c1, rtol=at, atol=1e-10)
            assert np.allclose(pts, p1, rtol=at, atol=1e-10)
        assert [len(res[r][key]["pts"]) for r in range(world)] == [s.shape[1] for s in shards]
    # the fused form runs the SAME iteration (equal inner iteration counts, the same number of messages:
    # a fused message counts as one collective) with TWO launches per inner iteration less — operator + update instead of
    # operator + camera-row kernel + mailbox kernel + update (every enqueued iteration, look-ahead launches included)
    for r in range(world):
        fu, un = res[r]["f64_pcg"], res[r]["f64_pcg_unfused"]
        assert fu["pcg_iterations"] == un["pcg_iterations"] > 0
        assert fu["collectives"] == un["collectives"]
        # (the linearisation's camera-space sums are fused too: finalize + one summing launch instead of finalize + mailbox kernel +
        # camera scales [+ publish])
        saved = un["kernel_launches"] - fu["kernel_launches"]
        assert saved >= 2 * fu["pcg_iterations"] + 8, (un["kernel_launches"], fu["kernel_launches"], fu["pcg_iterations"])
        assert np.allclose(fu["chi2"], un["chi2"], rtol=1e-10)
        # a set_tuning between two sharded LM calls that makes only SOME ranks re-time a choice — the timing
        # launches stay rank-local, the ranks agree on the fused message again, and the solve is the first call's
        rt_ = res[r]["f64_pcg_retuned"]
        assert rt_["pcg_iterations"] == fu["pcg_iterations"] and rt_["collectives"] == fu["collectives"] and rt_["fused_messages"] > 0
        assert np.allclose(rt_["chi2"], fu["chi2"], rtol=1e-10)
        # gr_bal_comm_info: what a SCALE record is audited with
        assert rt_["comm"]["size"] == world and rt_["comm"]["transport_name"] == "ipc-mailbox" and rt_["comm"]["mailboxes_opened"] == world - 1
        assert rt_["comm"]["fused_agreed"] == 1 and rt_["comm"]["oneshot_messages"] > 0


def test_late_rank_within_the_wait_bound(tmp_path):
    """A rank that reaches its first collective 3 s after its peers (first-touch, JIT, host set-up work): with the default
    30 s bound the all-reduce just waits for it and every rank gets the sum."""
    res = run_workers(2, tmp_path, {"GR_TEST_IPC_DELAY": "1:3"})
    for r in res:
        assert r["timeout_ms"] == 30000 and r["rc1"] == 0 and r["rc2"] == 0
        assert r["v"] == [3.0] * 5 and r["w"] == [2.0] * 3
    assert res[0]["t_first"] > 2.0  # rank 0 really waited for the late rank


def test_late_rank_beyond_the_wait_bound_is_an_error_not_a_wrong_sum(tmp_path):
    """The same with the bound set to 0.5 s (gr_bal_tuning.ipc_timeout_ms): the waiting rank gets GR_ERR_COMM, and so does
    every later collective on that communicator at once (it holds rank-local values: it must not carry on)."""
    res = run_workers(2, tmp_path, {"GR_TEST_IPC_DELAY": "1:3", "GR_IPC_TIMEOUT_MS": "500"})
    assert res[0]["timeout_ms"] == 500
    assert res[0]["rc1"] == 6 and res[0]["rc2"] == 6   # GR_ERR_COMM
    assert res[0]["t_first"] < 2.5 and res[0]["t_second"] < 0.5


def test_bench_under_torchrun_two_ranks_share_the_gpu():
    """The driver's own N > 1 command (python -m torch.distributed.run ... bench.py --gpus 2): the ranks it starts must not
    start ranks of their own."""
    env = dict(os.environ, GR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--repeats", "1", "--no-also"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_bench_two_ranks_share_the_gpu(tmp_path):
    """bench.py's N > 1 path (torch.distributed.run, one process per rank, landmark shards, barriers, max over ranks, the
    Venice `also` line) executed with two ranks on the one GPU of the box (GR_BENCH_SHARE_GPU=1: gloo process group, every
    all-reduce through the IPC mailboxes).  Guards the flow the driver launches on a multi-GPU node; not a measurement."""
    env = dict(os.environ, GR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    # started PLAINLY, as the driver starts its N = 1 line: bench.py itself launches its ranks (torch.distributed.run as a child
    # process, before anything touches the GPU) and returns their exit code
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--repeats", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["value"] > 0 and line["scaling"] == "strong"
    assert "landmark-sharded x2" in line["config"]["parallelism"]
    assert line["collectives_per_lm_iteration"] > 0
    assert line["parity_rel"] is not None and line["parity_rel"] < 1e-6     # the sharded trace against the oracle of the full problem
    venice = [a for a in line["also"] if "venice-1778" in a["workload"]]
    assert venice and venice[0]["value"] > 0


def _bench_line(out):
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(lines[0])


def test_bench_eight_ranks_on_the_one_device():
    """De-risking the first real 8-GPU run: the driver's own command line for N = 8 —
    `bench.py --gpus 8`, started plainly — with the eight ranks mapped onto the ONE device of this box (GR_BENCH_SHARE_GPU=1:
    gloo process group, every collective through the IPC mailboxes of eight processes).  Eight landmark shards of Ladybug-49,
    the fused inner-iteration message, barriers, max over ranks: all of it runs; the audit record must show eight ranks, eight
    mailboxes opened per rank, and the sharded trace must equal the FULL problem's oracle trace."""
    env = dict(os.environ, GR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "8", "--workload", "ladybug-49", "--steps", "6",
           "--warmup", "1", "--repeats", "2", "--no-also"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _bench_line(out)
    assert line["n_gpus"] == 8 and line["value"] > 0 and "landmark-sharded x8" in line["config"]["parallelism"]
    audit = line["shard_audit"]
    assert audit["ranks_seen"]["process_group"] == 8
    assert audit["ranks_seen"]["mailboxes_opened_per_rank"] == [7] * 8 or audit["ranks_seen"]["mailboxes_opened_per_rank"] == [8] * 8
    assert len(audit["per_rank_ms"]) == 8 and sum(r["observations"] for r in audit["per_rank_ms"]) == 31843
    assert all(t == "ipc-mailbox" for t in audit["transport_per_rank"])
    assert audit["peer_access"]["matrix"] == [[True]] * 8       # eight ranks, one visible device each
    assert line["parity_rel"] is not None and line["parity_rel"] < 1e-6


def test_bench_with_a_killed_rank_exits_non_zero():
    """(7a) one of the eight ranks dies after the warm-up (GR_TEST_KILL_RANK): its peers sit in a mailbox all-reduce that waits
    for it.  Bounded waits (2 s here) return GR_ERR_COMM, the job ends with a non-zero exit code and no JSON line — it does not
    hang and it does not report a number."""
    env = dict(os.environ, GR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GR_TEST_KILL_RANK="5", GR_IPC_TIMEOUT_MS="2000")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "8", "--workload", "ladybug-49", "--steps", "4",
           "--warmup", "1", "--repeats", "1", "--no-also", "--no-cpu-baseline"]
    t0 = time.time()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 300


@pytest.mark.parametrize("how", ["GR_COMM=rccl", "GR_COMM_TRANSPORT=0", "GR_COMM_TRANSPORT=2"])
def test_bench_rccl_paths_with_a_single_rank_communicator(how):
    """(7b) the two ways a run ends up on RCCL alone, as far as one device allows (RCCL cannot put two ranks on one device: world
    size 1, GR_BENCH_FORCE_COMM=1): RCCL asked for outright (GR_COMM=rccl, or gr_bal_tuning.comm_transport = 0 through the mailbox
    hand-shake), and the mailbox-refused path (comm_transport = 2: the peer mapping treated as refused, every rank agrees to drop
    to RCCL).  The audit record names the transport that really ran; the trace equals the oracle's."""
    k, v = how.split("=")
    env = dict(os.environ, GR_BENCH_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), **{k: v})
    env.pop("WORLD_SIZE", None); env.pop("GR_BENCH_SHARE_GPU", None)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "1", "--workload", "ladybug-49", "--steps", "6",
           "--warmup", "1", "--repeats", "1", "--no-also"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _bench_line(out)
    audit = line["shard_audit"]
    assert audit["transport_per_rank"] == ["rccl"] and audit["ranks_seen"]["rccl_comm_count"] == [1]
    assert audit["ranks_seen"]["mailboxes_opened_per_rank"] == [0]
    assert line["parity_rel"] is not None and line["parity_rel"] < 1e-6
