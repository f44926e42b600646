"""Landmark sharding of a BAL problem over the GPUs of one node.

One process per GPU.  Points — ordered by first observing camera — are cut into `world` contiguous ranges balanced by observation
count; a rank owns its points and ALL their observations; cameras are replicated.  Only
camera-space sums cross ranks (RCCL all-reduce inside libgraphite_mi355x.so)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .synth import BalProblem as HostProblem


POINT_WEIGHT = 4.0  # one point costs a rank about as much as four observations (the point-space vector work of an inner iteration;
#                     measured on the 8 shards of Final-13682 fp64: 2.2 ns per point, 0.48 ns per observation and LM iteration);
#                     fp32 (Venice-1778, 8 shards: 1.5 left the point-heavy last shard 15 % behind, 4.0 the observation-heavy first one 5 %): 3.0


def point_weight_for(dtype):
    return POINT_WEIGHT if np.dtype(dtype) == np.float64 else 3.0


def point_ranges(pt_idx, num_points, world, point_weight=POINT_WEIGHT):
    """Contiguous point ranges of (nearly) equal cost = observations + point_weight x points: list of (p0, p1)."""
    deg = np.bincount(pt_idx, minlength=num_points).astype(np.float64) + float(point_weight)
    cum = np.concatenate([[0], np.cumsum(deg)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * r / world, side="left")))
    cuts.append(num_points)
    cuts = np.maximum.accumulate(np.array(cuts))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def locality_order(cam_idx, pt_idx, num_points):
    """Points ordered by their FIRST observing camera (ties: original id).  Cut into contiguous ranges of this order, a shard's
    observations fall on the cameras of one band instead of on all of them: the camera runs of its camera-major order keep the
    length they have in the unsharded problem (a wave = one camera), and only the band's camera rows are non-zero in its messages."""
    first = np.full(num_points, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(first, pt_idx, cam_idx.astype(np.int64))
    return np.argsort(first, kind="stable")


def partition_by_landmark(prob: HostProblem, rank: int, world: int, locality: bool = True, point_weight: float = POINT_WEIGHT) -> HostProblem:
    """The shard of `prob` owned by `rank`: all cameras, `world` contiguous ranges of the points — in locality_order (default) or in
    the caller's numbering — renumbered from 0, and exactly their observations (original relative order kept).
    shard.point_ids: the original ids of the shard's points, in the shard's numbering; shard.obs_index: its observations."""
    Nc, Np, No = prob.shape
    cache = getattr(prob, "_shard_cache", None)
    if cache is None or cache[0] != (world, locality, point_weight):
        if locality:
            order = locality_order(prob.cam_idx, prob.pt_idx, Np)
            new_of_old = np.empty(Np, np.int64)
            new_of_old[order] = np.arange(Np)
            pt_new = new_of_old[prob.pt_idx]
        else:
            order = np.arange(Np)
            pt_new = prob.pt_idx.astype(np.int64)
        cache = ((world, locality, point_weight), order, pt_new, point_ranges(pt_new, Np, world, point_weight))
        try:
            prob._shard_cache = cache  # the ranks of one process (tests, projections) cut the same problem `world` times
        except AttributeError:
            pass
    _, order, pt_new, ranges = cache
    # validated for EVERY rank on every rank: all of them raise, or none does (a rank that raised alone would
    # leave its peers waiting in init_comm's broadcast)
    empty = [r for r, (a, b) in enumerate(ranges) if b <= a]
    if empty:
        raise ValueError(f"ranks {empty} of {world} would own no points ({Np} points)")
    p0, p1 = ranges[rank]
    sel = np.nonzero((pt_new >= p0) & (pt_new < p1))[0]
    ids = order[p0:p1]
    shard = HostProblem(prob.cameras, prob.points[ids].copy(), prob.obs[sel].copy(),
                        prob.cam_idx[sel].copy(), (pt_new[sel] - p0).astype(np.int32), f"{prob.name}[{rank}/{world}]")
    shard.point_ids = ids
    shard.point_range = (p0, p1) if not locality else None
    shard.obs_index = sel
    return shard


def contributor_masks(prob: HostProblem, world: int, locality: bool = True, point_weight: float = POINT_WEIGHT):
    """(world, Nc) booleans: rank r holds observations of camera c under partition_by_landmark's cut (what the ranks of a real
    run agree on at gr_bal_solver_update_structure; tools/shard_projection.py hands them to its one-rank runs)."""
    Nc, Np, No = prob.shape
    partition_by_landmark(prob, 0, world, locality, point_weight)  # fills the cache
    _, order, pt_new, ranges = prob._shard_cache
    cuts = np.array([a for a, _ in ranges] + [ranges[-1][1]])
    rank_of_obs = np.searchsorted(cuts, pt_new, side="right") - 1
    cnt = np.bincount(rank_of_obs * Nc + prob.cam_idx.astype(np.int64), minlength=world * Nc)
    return cnt.reshape(world, Nc) > 0


def set_contributors(problem, has, own_rank: int):
    """gr_bal_comm_set_contributors for a ONE-rank run that plays rank `own_rank` of `has` (contributor_masks): bit 0 = this shard,
    bits 1.. = the other ranks in rank order (the virtual ranks' slots)."""
    world = has.shape[0]
    if world > 32:
        return  # the masks are 32-bit; beyond that every rank pushes every row (as the engine itself decides)
    others = [q for q in range(world) if q != own_rank]
    mask = has[own_rank].astype(np.uint32)
    for b, q in enumerate(others):
        mask |= has[q].astype(np.uint32) << np.uint32(b + 1)
    mask = np.ascontiguousarray(mask, np.uint32)
    _lib.check(_lib.lib().gr_bal_comm_set_contributors(problem.h, mask.ctypes.data_as(C.c_void_p), C.c_int64(len(mask))))


def assemble_points(shards, pts_per_shard):
    """The ranks' point blocks (each in its shard's numbering) put back into the caller's numbering: (Np, 3)."""
    ids = np.concatenate([np.asarray(s.point_ids) for s in shards])
    cat = np.concatenate([np.asarray(p).reshape(-1, 3) for p in pts_per_shard])
    out = np.empty_like(cat)
    out[ids] = cat
    return out


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
