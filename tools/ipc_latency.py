"""Latency of the IPC one-shot all-reduce between processes sharing GPU 0 (no xGMI hop on a 1-GPU box: this is
the floor of the mechanism — push kernel + flag + reduce kernel — not a multi-GPU number).
usage: python tools/ipc_latency.py [world]   (spawns `world` children of itself)"""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(rank, world, port):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import graphite_amd as ga
    from graphite_amd import dist as gdist, synth
    prob = synth.make_config("ladybug-1723")
    shard = gdist.partition_by_landmark(prob, rank, world)
    g = ga.BalProblem(shard.cameras, shard.points, shard.obs, shard.cam_idx, shard.pt_idx, dtype=np.float64, shard=True)
    assert gdist.init_comm_ipc(g, rank, world, slot_bytes=1 << 21, rccl_fallback=False)
    f = g.lib.gr_bal_diag_time
    f.restype = C.c_double
    for n in (8, 512, 1778 * 9 // 2, 1723 * 9, 32768, 1778 * 90 // 2, 1723 * 90):  # (1778 x 9 and x 90 fp32 scalars = Venice's camera vector / linearisation group, in doubles)
        dist.barrier()
        us = f(g.h, C.c_int(8), C.c_int(n), C.c_int(200))
        if rank == 0:
            print(f"ipc all-reduce x{world} on one GPU: {n:6d} doubles ({n * 8 / 1024:.1f} KiB): {us:.2f} us", flush=True)
    dist.barrier()
    ct, lt, st = g.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=10)
    if rank == 0:
        print("sharded LM over IPC:", round(st["iterations_run"] / max(st["loop_seconds"], 1e-9), 1), "LM it/s;", st["collectives"], "collectives;",
              st["pcg_iterations"], "pcg iterations", flush=True)
    dist.barrier()
    g.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        child(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    else:
        world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), str(port)]) for r in range(world)]
        rc = 0
        for p in ps:
            try:
                rc |= p.wait(timeout=300)
            except subprocess.TimeoutExpired:
                p.kill()
                rc = 1
        sys.exit(rc)
