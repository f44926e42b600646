"""N > 1 path on CPU: world_size-2 gloo processes check the landmark partition and that the
camera-space sums the GPU path all-reduces (unscaled Hcc, bc, chi2, operator camera rows)
add up to the unsharded quantities.  The per-rank arithmetic here is the ORACLE (test
infrastructure), the partition / collective layout is the product's (graphite_amd.dist)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_point_ranges_partition_everything():
    from graphite_amd import dist as gdist, synth
    prob = synth.make_config("mini-50")
    Nc, Np, No = prob.shape
    for world in (1, 2, 3, 8):
        ranges = gdist.point_ranges(prob.pt_idx, Np, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == Np
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        shards = [gdist.partition_by_landmark(prob, r, world) for r in range(world)]
        assert sum(s.shape[2] for s in shards) == No and sum(s.shape[1] for s in shards) == Np
        seen = np.concatenate([s.obs_index for s in shards])
        assert np.array_equal(np.sort(seen), np.arange(No))           # every observation exactly once
        counts = [s.shape[2] + gdist.POINT_WEIGHT * s.shape[1] for s in shards]  # the balanced quantity: observations + POINT_WEIGHT x points
        assert max(counts) - min(counts) <= max(2 * (np.bincount(prob.pt_idx).max() + gdist.POINT_WEIGHT), 0.2 * (No + gdist.POINT_WEIGHT * Np) / world)
        for s in shards:
            assert s.pt_idx.min() == 0 and s.pt_idx.max() == s.shape[1] - 1
            assert np.array_equal(s.cameras, prob.cameras)


def test_locality_cut_masks_and_reassembly():
    """Points cut by first observing camera: the shards' contributor masks are what contributor_masks computes, a camera is held by
    fewer ranks than under the caller's numbering, and assemble_points puts the ranks' point blocks back in the caller's order."""
    from graphite_amd import dist as gdist, synth
    prob = synth.make_config("ladybug-49")
    Nc, Np, No = prob.shape
    world = 4
    shards = [gdist.partition_by_landmark(prob, r, world) for r in range(world)]
    has = gdist.contributor_masks(prob, world)
    for r, s in enumerate(shards):
        assert np.array_equal(np.bincount(s.cam_idx, minlength=Nc) > 0, has[r])
        assert np.array_equal(s.points, prob.points[s.point_ids])
        first = np.full(s.shape[1], Nc, np.int64)
        np.minimum.at(first, s.pt_idx, s.cam_idx.astype(np.int64))
        assert np.all(np.diff(first) >= 0)                             # inside a shard the points are still ordered by first camera
    assert np.array_equal(gdist.assemble_points(shards, [s.points for s in shards]), prob.points)
    prob2 = synth.make_config("ladybug-49")
    has_plain = np.stack([np.bincount(gdist.partition_by_landmark(prob2, r, world, locality=False).cam_idx, minlength=Nc) > 0 for r in range(world)])
    assert has.sum() < has_plain.sum()                                  # fewer (rank, camera) pairs: shorter messages, longer camera runs


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    from graphite_amd import dist as gdist, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = synth.make_config("mini-50")
    Nc = prob.shape[0]
    s = gdist.partition_by_landmark(prob, rank, world)
    o = oracle.BalOracle(s.cameras, s.points, s.obs, s.cam_idx, s.pt_idx)
    o.set_scale_system(False)          # unscaled blocks are what travels
    o.linearize()
    o.hessian_update()
    payload = np.concatenate([o.get("Hcc"), o.get("b")[:9 * Nc], [o.chi2()]])
    t = torch.from_numpy(payload.copy())
    dist.all_reduce(t)                 # the same three sums gr_bal_linearize all-reduces with RCCL
    # camera rows of the operator: y_c = sum_obs Jc^T (Jc p_c + Jp p_l)
    rng = np.random.default_rng(0)
    p_full = rng.normal(size=9 * Nc + 3 * prob.shape[1])
    p_loc = np.concatenate([p_full[:9 * Nc], p_full[9 * Nc:].reshape(-1, 3)[s.point_ids].ravel()])
    Jc = o.get("Jc").reshape(-1, 9, 2)
    Jp = o.get("Jp").reshape(-1, 3, 2)
    u = np.einsum("fde,fd->fe", Jc, p_loc[:9 * Nc].reshape(Nc, 9)[s.cam_idx]) + \
        np.einsum("fde,fd->fe", Jp, p_loc[9 * Nc:].reshape(-1, 3)[s.pt_idx])
    rows = np.zeros((Nc, 9))
    np.add.at(rows, s.cam_idx, np.einsum("fde,fe->fd", Jc, u))
    tr = torch.from_numpy(rows.ravel().copy())
    dist.all_reduce(tr)
    if rank == 0:
        q.put((t.numpy(), tr.numpy()))
    dist.destroy_process_group()


def test_camera_space_sums_over_gloo():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import oracle
    from graphite_amd import synth
    oracle.build()
    world, port = 2, 29500 + os.getpid() % 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    summed, rows = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    prob = synth.make_config("mini-50")
    Nc = prob.shape[0]
    o = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx)
    o.set_scale_system(False)
    o.linearize()
    o.hessian_update()
    ref = np.concatenate([o.get("Hcc"), o.get("b")[:9 * Nc], [o.chi2()]])
    assert np.allclose(summed, ref, rtol=1e-11, atol=1e-9)
    rng = np.random.default_rng(0)
    p_full = rng.normal(size=o.n)
    Jc = o.get("Jc").reshape(-1, 9, 2)
    Jp = o.get("Jp").reshape(-1, 3, 2)
    u = np.einsum("fde,fd->fe", Jc, p_full[:9 * Nc].reshape(Nc, 9)[prob.cam_idx]) + \
        np.einsum("fde,fd->fe", Jp, p_full[9 * Nc:].reshape(-1, 3)[prob.pt_idx])
    full = np.zeros((Nc, 9))
    np.add.at(full, prob.cam_idx, np.einsum("fde,fe->fd", Jc, u))
    assert np.allclose(rows, full.ravel(), rtol=1e-10, atol=1e-8)
