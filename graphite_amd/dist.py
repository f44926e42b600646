"""Landmark sharding of a BAL problem over the GPUs of one node.

One process per GPU.  Points are cut into `world` contiguous ranges balanced by observation
count; a rank owns its points and ALL their observations; cameras are replicated.  Only
camera-space sums cross ranks (RCCL all-reduce inside libgraphite_mi355x.so)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .synth import BalProblem as HostProblem


def point_ranges(pt_idx, num_points, world):
    """Contiguous point ranges with (nearly) equal observation counts: list of (p0, p1)."""
    deg = np.bincount(pt_idx, minlength=num_points).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(deg)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * r / world, side="left")))
    cuts.append(num_points)
    cuts = np.maximum.accumulate(np.array(cuts))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def partition_by_landmark(prob: HostProblem, rank: int, world: int) -> HostProblem:
    """The shard of `prob` owned by `rank`: all cameras, points [p0, p1) renumbered from 0 and
    exactly their observations (original relative order kept)."""
    Nc, Np, No = prob.shape
    ranges = point_ranges(prob.pt_idx, Np, world)
    # validated for EVERY rank on every rank: all of them raise, or none does (a rank that raised alone would
    # leave its peers waiting in init_comm's broadcast)
    empty = [r for r, (a, b) in enumerate(ranges) if b <= a]
    if empty:
        raise ValueError(f"ranks {empty} of {world} would own no points ({Np} points)")
    p0, p1 = ranges[rank]
    sel = np.nonzero((prob.pt_idx >= p0) & (prob.pt_idx < p1))[0]
    shard = HostProblem(prob.cameras, prob.points[p0:p1].copy(), prob.obs[sel].copy(),
                        prob.cam_idx[sel].copy(), (prob.pt_idx[sel] - p0).astype(np.int32), f"{prob.name}[{rank}/{world}]")
    shard.point_range = (p0, p1)
    shard.obs_index = sel
    return shard


def init_comm(problem, rank: int, world: int):
    """Create the RCCL communicator of `problem` (a graphite_amd.BalProblem built with shard=True).
    The 128-byte ncclUniqueId is made on rank 0 and broadcast with torch.distributed."""
    import torch
    import torch.distributed as dist
    lib = _lib.lib()
    uid = (C.c_char * 128)()
    if rank == 0:
        _lib.check(lib.gr_comm_unique_id(uid))
    t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    raw = bytes(t.cpu().numpy().tobytes())
    buf = (C.c_char * 128).from_buffer_copy(raw)
    _lib.check(lib.gr_bal_comm_init(problem.h, buf, C.c_int(rank), C.c_int(world)))


def init_comm_ipc(problem, rank: int, world: int, slot_bytes: int = 4 << 20, rccl_fallback: bool = True):
    """One-shot peer all-reduce over IPC-mapped mailboxes (csrc/comm.hpp IpcComm), RCCL for messages beyond `slot_bytes`.
    The 64-byte mailbox handles are gathered with torch.distributed (any backend).  Returns True when the mailboxes are in
    use, False when the start-up verification failed somewhere and every rank fell back to RCCL."""
    import torch
    import torch.distributed as dist
    lib = _lib.lib()
    mine = (C.c_char * 64)()
    _lib.check(lib.gr_bal_comm_ipc_mailbox(problem.h, C.c_size_t(slot_bytes), C.c_int(world), mine))
    on_gpu = dist.get_backend() == "nccl"
    t = torch.frombuffer(bytearray(mine.raw), dtype=torch.uint8).clone()
    if on_gpu:
        t = t.cuda()
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    raw = b"".join(bytes(g.cpu().numpy().tobytes()) for g in gathered)
    handles = (C.c_char * (64 * world)).from_buffer_copy(raw)
    uid = None
    if rccl_fallback:
        u = (C.c_char * 128)()
        if rank == 0:
            _lib.check(lib.gr_comm_unique_id(u))
        tu = torch.frombuffer(bytearray(u.raw), dtype=torch.uint8).clone()
        if on_gpu:
            tu = tu.cuda()
        dist.broadcast(tu, src=0)
        uid = (C.c_char * 128).from_buffer_copy(bytes(tu.cpu().numpy().tobytes()))
    used = C.c_int(0)
    _lib.check(lib.gr_bal_comm_init_ipc(problem.h, handles, C.c_int(rank), C.c_int(world), uid, C.byref(used)))
    return bool(used.value)


def init_local_group(problems):
    """TEST ONLY: in-process group of shards on one GPU (one host thread per shard afterwards)."""
    lib = _lib.lib()
    arr = (C.c_void_p * len(problems))(*[p.h for p in problems])
    _lib.check(lib.gr_bal_comm_init_local(arr, C.c_int(len(problems))))
