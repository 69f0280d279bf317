"""World-size-2 gloo test (CPU) of the data-parallel exchange step: view
sharding, the flat gradient bucket SUM all-reduce and the three densification
statistics reductions must reproduce the 1-rank result on the same global batch."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy_problem():
    """Stand-in for render(): a differentiable per-view function of shared parameters
    (the collectives do not care what produced the gradients)."""
    g = torch.Generator().manual_seed(3)
    params = [torch.randn(50, 3, generator=g).requires_grad_(True),
              torch.randn(7, generator=g).requires_grad_(True),
              torch.randn(4, 5, generator=g).requires_grad_(True),
              torch.randn(3, generator=g).requires_grad_(True)]       # never used: grad stays None
    views = [torch.randn(50, 3, generator=g) for _ in range(4)]
    return params, views


def _view_loss(params, view):
    return ((params[0] * view).sum(dim=1).tanh().mean() + (params[1] ** 2).sum() * view.mean()
            + (params[2].sum() * view[0, 0]))


def _view_stats(view):
    radii = (view[:, 0] * 10).to(torch.int32).clamp(min=0)
    vis = radii > 0
    vgrad = view * 0.1
    return radii, vis, vgrad


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "4d-gaussian-splatting_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fdgs import dp
    params, views = _toy_problem()
    mine = dp.shard_views(views)
    assert len(mine) == len(views) // world
    scale = dp.local_loss_scale(len(mine), len(views))
    loss = sum(_view_loss(params, v) for v in mine) * scale
    loss.backward()
    bucket = dp.GradBucket(params)
    bucket.allreduce_()
    stats = [_view_stats(v) for v in mine]
    radii = torch.stack([s[0] for s in stats]).max(0).values
    vis = torch.stack([s[1] for s in stats]).any(0)
    vgrad = sum(s[2] for s in stats)
    radii, vis, vgrad = dp.reduce_densification_stats(radii, vis, vgrad)
    # inside local_only() a rank of the group is a group of its own: it keeps every view and exchanges nothing
    with dp.local_only():
        assert not dp.is_dist() and dp.rank_world() == (0, 1)
        assert len(dp.shard_views(views)) == len(views) and dp.local_loss_scale(len(views), len(views)) == 1.0 / len(views)
    assert dp.is_dist() and dp.rank_world() == (rank, world)
    if rank == 0:
        torch.save({"grads": [p.grad for p in params], "radii": radii, "vis": vis, "vgrad": vgrad}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_one_rank(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    params, views = _toy_problem()
    loss = sum(_view_loss(params, v) for v in views) / len(views)
    loss.backward()
    for p, g in zip(params, got["grads"]):
        if p.grad is None:      # a tensor no rank has a gradient for stays without one (the optimizer skips it)
            assert g is None
        else:
            assert torch.allclose(g, p.grad, rtol=1e-5, atol=1e-7)
    stats = [_view_stats(v) for v in views]
    assert torch.equal(got["radii"], torch.stack([s[0] for s in stats]).max(0).values)      # integer: bitwise
    assert torch.equal(got["vis"], torch.stack([s[1] for s in stats]).any(0))
    assert torch.allclose(got["vgrad"], sum(s[2] for s in stats), rtol=1e-5, atol=1e-7)


def test_shard_views_partition():
    sys.path.insert(0, os.path.join(ROOT, "4d-gaussian-splatting_amd"))
    from fdgs import dp
    views = list(range(8))
    for world in (1, 2, 4, 8):
        parts = [dp.shard_views(views, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == views and all(len(p) == 8 // world for p in parts)


def test_arena_ranges_for_every_stage_and_exchange_order():
    """The bookkeeping of the in-place exchange (fdgs/pipeline.py): regions in order, 256-byte aligned, and the
    range that is all-reduced for coarse / fine stage x early exchange done / not done; the early range and the late
    range partition the stage's gradients, whatever the Gaussian count."""
    sys.path.insert(0, os.path.join(ROOT, "4d-gaussian-splatting_amd"))
    from fdgs.pipeline import arena_layout, exchange_bounds
    D = 2365440 + 94395                                    # the deformation field's trainable floats
    for N in (1, 63, 6000, 300000, 326023, 2000000):
        regions, m, total = arena_layout(N, D)
        names = [r[0] for r in regions]
        assert names == ["dflat", "xyz", "means2D", "scaling", "rotation", "opacity", "shs"]
        sizes = dict(dflat=D, xyz=3 * N, means2D=3 * N, scaling=3 * N, rotation=4 * N, opacity=N, shs=48 * N)
        for a, b in zip(names, names[1:] + ["end"]):
            assert m[a] % 64 == 0 and m[b] - m[a] >= sizes[a] and m[b] - m[a] < sizes[a] + 64
        assert m["end"] == total
        early = (m["means2D"], m["end"])
        for fine in (False, True):
            whole = exchange_bounds(m, fine, False)
            late = exchange_bounds(m, fine, True)
            assert whole == ((m["dflat"] if fine else m["xyz"]), m["end"])
            assert late[0] == whole[0] and late[1] == early[0] and early[1] == whole[1]     # late + early = whole
            assert late[1] - late[0] >= (D if fine else 0) + 3 * N
        # bytes one rank exchanges per fine step: the 59 N + D floats of the survey, plus means2D and padding
        assert 4 * (total - m["dflat"]) >= 4 * (59 * N + D)


def _arena_worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "4d-gaussian-splatting_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fdgs import dp
    from fdgs.pipeline import arena_layout, exchange_bounds
    N, D = 1000, 777
    _, m, total = arena_layout(N, D)
    views = list(range(8))
    mine = dp.shard_views(views)
    assert mine == views[rank::world] and len(mine) == 8 // world
    # every view contributes (view + 1) to every float of the arena; a rank's arena holds the sum over its views
    arena = torch.full((total,), float(sum(v + 1 for v in mine)))
    radii = torch.full((N,), max(mine), dtype=torch.int32)
    # the early range travels asynchronously (with the radii) while the late range is still being produced ...
    early = arena[m["means2D"]:m["end"]]
    work = dp.all_reduce_async(early)
    work_r = dp.all_reduce_async(radii, dist.ReduceOp.MAX)
    # ... then the rest in place, for the fine stage
    first, last = exchange_bounds(m, True, True)
    dp.all_reduce_(arena[first:last], dist.ReduceOp.SUM)
    work.wait()
    work_r.wait()
    assert torch.equal(arena, torch.full((total,), 36.0)), (rank, arena.unique())
    assert int(radii.min()) == 7 and int(radii.max()) == 7
    # coarse stage, no early exchange: everything but the deformation field in one call
    arena = torch.full((total,), float(rank + 1))
    first, last = exchange_bounds(m, False, False)
    dp.all_reduce_(arena[first:last], dist.ReduceOp.SUM)
    assert float(arena[0]) == rank + 1 and float(arena[m["xyz"]]) == world * (world + 1) / 2
    if rank == 0:
        torch.save({"ok": True, "world": world}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_arena_exchange_with_four_and_eight_ranks(tmp_path):
    """The exchange step's calls over gloo with the world sizes of the scaling runs: the early range and the radii
    asynchronously, the late range in place; a global batch of 8 views split 2 and 1 per rank."""
    for world in (4, 8):
        out = str(tmp_path / f"arena{world}.pt")
        mp.spawn(_arena_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert torch.load(out) == {"ok": True, "world": world}
