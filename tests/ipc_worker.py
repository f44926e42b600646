"""One rank of tests/test_gpu_ipc.py: a fresh process that shares GPU 0 with its peers.

argv: rank world port outfile.  The process group (gloo, CPU) only carries the 64-byte IPC handles; every
all-reduce of the solve goes through the peers' mailboxes in HBM (csrc/comm.hpp IpcComm, no RCCL: RCCL cannot
put two ranks on one GPU)."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import graphite_amd as ga
    from graphite_amd import dist as gdist, synth, _lib
    res = {}
    prob = synth.make_config("mini-50")
    shard = gdist.partition_by_landmark(prob, rank, world)
    slot = int(os.environ.get("GR_TEST_IPC_SLOT", 1 << 16))
    if os.environ.get("GR_TEST_IPC_DELAY"):
        # one rank arrives late at its first collective (after the start-up self-test): within the wait bound (gr_bal_tuning
        # .ipc_timeout_ms, here through GR_IPC_TIMEOUT_MS) the all-reduce simply takes that long; beyond it the waiting
        # ranks get GR_ERR_COMM — at once also for every later collective — instead of rank-local values
        import time
        late, seconds = os.environ["GR_TEST_IPC_DELAY"].split(":")
        g = ga.BalProblem(shard.cameras, shard.points, shard.obs, shard.cam_idx, shard.pt_idx, dtype=np.float64, shard=True)
        assert gdist.init_comm_ipc(g, rank, world, slot_bytes=slot, rccl_fallback=False)
        lib = _lib.lib()
        dist.barrier()
        if rank == int(late):
            time.sleep(float(seconds))
        v = np.full(5, float(rank + 1))
        t0 = time.perf_counter()
        rc1 = int(lib.gr_bal_comm_allreduce_host(g.h, v.ctypes.data_as(C.POINTER(C.c_double)), C.c_size_t(5)))
        t1 = time.perf_counter()
        w = np.full(3, 1.0)
        rc2 = int(lib.gr_bal_comm_allreduce_host(g.h, w.ctypes.data_as(C.POINTER(C.c_double)), C.c_size_t(3)))
        t2 = time.perf_counter()
        with open(out, "w") as f:
            json.dump({"rc1": rc1, "rc2": rc2, "v": v.tolist(), "w": w.tolist(), "t_first": t1 - t0, "t_second": t2 - t1,
                       "timeout_ms": g.get_tuning()["ipc_timeout_ms"]}, f)
        dist.barrier()
        g.close()
        dist.destroy_process_group()
        return
    for dtype, tag in ((np.float64, "f64"), (np.float32, "f32")):
        g = ga.BalProblem(shard.cameras, shard.points, shard.obs, shard.cam_idx, shard.pt_idx, dtype=dtype, shard=True)
        used = gdist.init_comm_ipc(g, rank, world, slot_bytes=slot, rccl_fallback=False)
        assert used
        if tag == "f64":
            # raw all-reduces against the host: ragged sizes, both mailbox sets, back to back
            lib = _lib.lib()
            sums = []
            for n in (1, 2, 63, 64, 65, 1000, slot // 8):
                v = np.random.default_rng(1000 * n + rank).standard_normal(n)
                _lib.check(lib.gr_bal_comm_allreduce_host(g.h, v.ctypes.data_as(C.POINTER(C.c_double)), C.c_size_t(n)))
                sums.append(v.tolist())
            res["sums"] = sums
            v = np.zeros(slot // 8 + 1)
            rc = lib.gr_bal_comm_allreduce_host(g.h, v.ctypes.data_as(C.POINTER(C.c_double)), C.c_size_t(v.size))
            res["oversize_rc"] = int(rc)
        # "pcg": the default on landmark shards = the inner iteration's message FUSED into the operator / update launches
        # (gr_bal_tuning.shard_fused, 2 launches per inner iteration); "pcg_unfused": the same solve with a kernel of its own
        # for the all-reduce (4 launches per inner iteration)
        for solver, sname in ((ga.SOLVER_PCG, "pcg"), (ga.SOLVER_PCG, "pcg_unfused"), (ga.SOLVER_PCG_SCHUR_IMPLICIT, "implicit")):
            if tag == "f32" and sname != "pcg":
                continue
            g.set_tuning(shard_fused=0 if sname == "pcg_unfused" else -1)
            g.set_params(shard.cameras, shard.points)
            ct, lt, st = g.levenberg_marquardt(solver=solver, iterations=8)
            cams, pts = g.get_params()
            res[f"{tag}_{sname}"] = {"chi2": [float(x) for x in ct], "lambda": [float(x) for x in lt],
                                     "pcg_iterations": int(st["pcg_iterations"]), "collectives": int(st["collectives"]),
                                     "kernel_launches": int(st["kernel_launches"]),
                                     "cams": np.asarray(cams, np.float64).tolist(), "pts": np.asarray(pts, np.float64).tolist()}
        if tag == "f64":
            # set_tuning between two sharded LM calls, with the ranks on DIFFERENT sides of a timed choice — rank 0 re-times
            # the point-record layout (two runs of eight operator launches inside solver_update_structure), the others have it forced
            # and time nothing.  Timing launches must be rank-local (never the fused message); the solve that follows must fuse again
            # and reproduce the first call.
            g.set_tuning(shard_fused=-1, point_records=(-1 if rank == 0 else 0))
            g.set_params(shard.cameras, shard.points)
            ct, lt, st = g.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=8)
            res["f64_pcg_retuned"] = {"chi2": [float(x) for x in ct], "pcg_iterations": int(st["pcg_iterations"]), "collectives": int(st["collectives"]),
                                      "fused_messages": int(st["fused_messages"]), "comm": g.comm_info()}
        dist.barrier()
        g.close()
    with open(out, "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
