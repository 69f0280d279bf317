#!/usr/bin/env python
"""Benchmark of the 4DGS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one training iteration of the reference's loop body
(src/runner/fdgs_trainer.py:112-205) on a synthetic DyNeRF-shaped scene: per
view  deformation forward -> activations -> rasterizer forward -> L1 loss ->
backward through both -> densification statistics -> Adam step.

--gpus 1 (default): one view per iteration (the reference's batch_size 1,
configs/train/dynerf.yaml:18).  --gpus N > 1: the global batch is a fixed 8
views / time stamps per iteration (--global-batch), rank r renders views
r, r+N, ... (8/N per rank: the reference's batch loop fdgs_trainer.py:239-248
split over ranks), gradients and statistics are all-reduced over RCCL; `value`
is then iterations of 8 views per second (strong scaling).  --global-batch 0
means one view per rank (weak scaling).

Run as `python bench.py --gpus N` without a launcher, this process touches no
GPU: it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a child, forwards rank 0's JSON line and exits with the child's status.
Under a launcher (WORLD_SIZE set) it is one of the ranks.  Rank 0 prints ONE
JSON line.
"""
import argparse
import gc
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "4d-gaussian-splatting_amd"))
sys.path.insert(0, ROOT)

torch = None      # imported in main(), after the decision between launcher and rank
dist = None


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=25,
                    help="untimed steps; the default walks all 20 training views once so that buffer sizes settle")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="views per iteration over all ranks (default: 1 at --gpus 1, 8 at --gpus > 1; 0 = one per rank)")
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--width", type=int, default=1352)
    ap.add_argument("--height", type=int, default=1014)
    ap.add_argument("--render-frames", type=int, default=60, help="frames of the inference-only FPS legs (0 = skip)")
    ap.add_argument("--scale-mult", type=float, default=1.0, help="multiplier on the Gaussian scales (0.6 for the 2M case)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all-calls", action="store_true",
                    help="event pairs around every entry point inside the timed region (default: only around the "
                         "north-star kernel's; the others are timed in a leg of their own)")
    ap.add_argument("--fp32-mfma-steps", type=int, default=10,
                    help="steps of the extra leg that forms the deformation matrix products with fp32 MFMA (0 = skip)")
    ap.add_argument("--batch8-steps", type=int, default=6,
                    help="at --gpus 1: steps of the extra leg with 8 views per iteration, the one-GPU point of the "
                         "fixed-global-batch scaling curve (0 = skip)")
    ap.add_argument("--one-gpu-steps", type=int, default=5,
                    help="at --gpus > 1: steps of the last leg, the same global batch on rank 0 alone while the other "
                         "ranks wait (the one-GPU point of the scaling curve, measured inside the job; 0 = skip)")
    ap.add_argument("--cpu-tile-stride", type=int, default=48,
                    help="cpu baseline blends every k-th tile, and every (k/2)-th for a second estimate")
    ap.add_argument("--dry-run", action="store_true",
                    help="come up as ranks (gloo, no GPU, no kernels), prove the group and print the line's skeleton")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N with no WORLD_SIZE in the environment
# ----------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Starts N fresh rank processes (one per GPU) and forwards rank 0's JSON line.  Nothing here initialises
    the GPU and nothing replaces this process: the ranks are children, their exit status is ours."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # RCCL needs dmabuf IPC on this driver
    env.setdefault("OMP_NUM_THREADS", "16")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)    # stderr passes through
    line = None
    for out in child.stdout:
        text = out.strip()
        if text.startswith("{") and '"metric"' in text:
            line = text
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if rc != 0:
        raise SystemExit(f"bench.py: the {args.gpus}-rank job failed with status {rc}")
    if line is None:
        raise SystemExit("bench.py: rank 0 printed no result line")
    if json.loads(line).get("n_gpus") != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the result line reports {json.loads(line).get('n_gpus')}")
    print(line, flush=True)


def dry_run(args, world, rank):
    """The group comes up over gloo on the CPU and every rank is counted; no kernel runs."""
    import torch as t
    import torch.distributed as d
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    d.init_process_group(backend="gloo", rank=rank, world_size=world)
    seen = t.zeros(world, dtype=t.int64)
    seen[rank] = 1 + rank
    d.all_reduce(seen)
    assert seen.tolist() == [1 + r for r in range(world)], seen
    B = resolve_global_batch(args, world)
    if rank == 0:
        print(json.dumps({"metric": "train it/s", "value": None, "unit": "it/s", "n_gpus": world, "dry_run": True,
                          "steps": args.steps, "warmup": args.warmup, "ranks_seen": int((seen > 0).sum()),
                          "scaling": scaling_label(args, world),
                          "config": {"global_batch_views": B, "views_per_gpu": B // world}}), flush=True)
    d.barrier()
    d.destroy_process_group()


def resolve_global_batch(args, world):
    B = args.global_batch
    if B is None:
        B = 1 if world == 1 else 8
    if B == 0:
        B = world
    if B % world != 0:
        raise SystemExit(f"--global-batch {B} does not divide over {world} ranks")
    return B


def scaling_label(args, world):
    return "weak" if (world == 1 or args.global_batch == 0) else "strong"


# ----------------------------------------------------------------------------------------------------------
def build_scene(args, dev):
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import DeformHolder
    g = ss.GaussianParams(args.gaussians, device=dev, scale_mult=args.scale_mult)
    torch.manual_seed(ss.SEED)                # train.py:27: every rank builds the same replica
    net = DeformationFields(ss.DYNERF_NET_KWARGS)
    ss.perturb_deformation_(net)
    net = net.to(dev)
    with torch.no_grad():
        xyz = g._xyz.detach()
        net.deformation_net.set_aabb(xyz.max(0).values.tolist(), xyz.min(0).values.tolist())   # scene.py:26-33
    deforms = DeformHolder(net)
    cams = ss.make_cameras(21, args.width, args.height, device=dev)
    bg = torch.ones(3, device=dev)            # white_background: true (dynerf.yaml:57)
    # ground truth: a render of a second, differently seeded scene (dense L1 gradients)
    g2 = ss.GaussianParams(args.gaussians, seed=ss.SEED + 100, device=dev, requires_grad=False,
                           scale_mult=args.scale_mult)
    with torch.no_grad():
        for cam in cams:
            cam.original_image = render(cam, g2, deforms, bg, stage="coarse")["render"].clamp(0, 1).clone()
    del g2
    return g, net, deforms, cams, bg


def host_threads():
    """Threads for the CPU baseline: the CPUs this process may actually use.  os.cpu_count() counts the whole
    machine; the scheduler affinity and the OMP_NUM_THREADS the box exports (its per-GPU CPU share) bound it."""
    total = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = total
    env = os.environ.get("OMP_NUM_THREADS")
    share = int(env) if env and env.isdigit() else None
    threads = min(x for x in (total, affinity, share) if x)
    return total, affinity, share, threads


def _median_after_warmup(fn, repeats=3):
    fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def cpu_baseline(args, g, net, cam, bg):
    """The CPU oracle (oracle/deform_ref.py + oracle/raster_ref.py, torch CPU) on the same inputs, each leg
    timed as the median of 3 runs after one warm-up: deformation forward+backward and rasterizer
    preprocess/sort + their backward on all points, blending forward+backward on every k-th tile, divided by
    that sample's share of the tile-list entries - at two strides (k and k/2), so that the extrapolation's
    stability shows."""
    from oracle import deform_ref, raster_ref as rr
    total, affinity, share, threads = host_threads()
    torch.set_num_threads(threads)
    cores = torch.get_num_threads()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for k in sd:
        if k.startswith("deformation_net.") and not k.endswith("aabb"):
            sd[k].requires_grad_(True)
    ins = [t.detach().cpu().requires_grad_(True) for t in (g._xyz, g._scaling, g._rotation, g._opacity)]
    shs = g.get_features.detach().cpu().requires_grad_(True)
    N = ins[0].shape[0]
    S = rr.Settings(args.height, args.width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg.cpu(), 1.0,
                    cam.world_view_transform.cpu(), cam.full_proj_transform.cpu(), 3, cam.camera_center.cpu())
    gt = cam.original_image.cpu()
    tvec = torch.full((N, 1), float(cam.time))
    gx, gy = rr.tile_grid(S)
    st = {}

    def deform_fwd():
        st["def"] = deform_ref.deform_forward(sd, ins[0], ins[1], ins[2], ins[3], shs, tvec)

    def pre_sort():
        pts, sc, ro, op, sh = st["def"]
        st["m2"] = torch.zeros(N, 3, requires_grad=True)
        pre = rr.preprocess(pts, torch.exp(sc), torch.nn.functional.normalize(ro), torch.sigmoid(op), sh, S, st["m2"])
        keys, vals, _ = rr.build_keys(pre, S)
        skeys, svals = rr.sort_keys(keys, vals)
        st["pre"], st["svals"], st["ranges"] = pre, svals, rr.tile_ranges(skeys, gx * gy)

    t_def_f = _median_after_warmup(deform_fwd)
    t_pre = _median_after_warmup(pre_sort)
    pre, svals, ranges = st["pre"], st["svals"], st["ranges"]
    mids = [pre[k] for k in ("px", "py", "conA", "conB", "conC", "opacity", "rgb")]
    lens = (ranges[:, 1] - ranges[:, 0]).to(torch.float64)

    def blend_legs(stride):
        tiles = list(range(0, gx * gy, stride))
        frac = float(lens[tiles].sum() / max(float(lens.sum()), 1.0))
        mask = torch.zeros(gy * 16, gx * 16, dtype=torch.bool)
        for t in tiles:
            ty, tx = divmod(t, gx)
            mask[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
        mask = mask[:args.height, :args.width]
        box = {}

        def fwd():
            color, _, _, _ = rr.blend(pre, svals, ranges, S, tiles)
            box["loss"] = (torch.abs(color - gt) * mask).sum() / (3.0 * args.height * args.width)

        def bwd():
            box["gmid"] = torch.autograd.grad(box["loss"], mids, retain_graph=True)

        tf = _median_after_warmup(fwd)
        tb = _median_after_warmup(bwd)
        return {"stride": stride, "tiles": len(tiles), "entry_fraction": frac, "fwd_s": tf, "bwd_s": tb,
                "extrapolated_s": (tf + tb) / max(frac, 1e-9)}, box["gmid"]

    b1, gmid = blend_legs(args.cpu_tile_stride)
    b2, _ = blend_legs(max(1, args.cpu_tile_stride // 2))
    leaves = ins + [shs] + [v for v in sd.values() if v.requires_grad]

    def rest_bwd():     # preprocess + activations + deformation backward, all points
        for t in leaves:
            t.grad = None
        torch.autograd.backward(mids, gmid, retain_graph=True)

    t_rest_b = _median_after_warmup(rest_bwd)
    from fdgs.train_step import TrainStep

    def reg():          # grid regulariser of the fine stage, fwd + bwd
        deform_ref.compute_regulation(sd, *TrainStep.DYNERF_REGULATION).backward()

    t_reg = _median_after_warmup(reg)
    blend_s = 0.5 * (b1["extrapolated_s"] + b2["extrapolated_s"])
    est = t_def_f + t_pre + blend_s + t_rest_b + t_reg
    return {"value": 1.0 / est, "unit": "it/s", "cores": cores, "threads": cores, "cpu_count": total,
            "cpu_affinity": affinity, "omp_num_threads_env": share, "kind": "port",
            "sample": (f"median of 3 after 1 warm-up per leg on {cores} threads (machine: {total} CPUs, this process may "
                       f"use {affinity}, the box's per-GPU share OMP_NUM_THREADS={share}); deformation fwd+bwd, grid "
                       f"regulariser and rasterizer preprocess/sort on all {N} points; blending fwd+bwd on every "
                       f"{b1['stride']}th tile ({b1['tiles']} of {gx * gy} tiles, {b1['entry_fraction']:.4f} of the "
                       f"tile-list entries) and on every {b2['stride']}th ({b2['tiles']} tiles, {b2['entry_fraction']:.4f}), "
                       f"each divided by its fraction: {b1['extrapolated_s']:.2f} s and {b2['extrapolated_s']:.2f} s, mean used"),
            "legs_s": {"deform_fwd": t_def_f, "preprocess_sort": t_pre, "blend_fwd_bwd_extrapolated": blend_s,
                       "preprocess_deform_bwd": t_rest_b, "regulariser": t_reg},
            "blend_estimates": [b1, b2],
            "blend_estimates_rel_diff": abs(b1["extrapolated_s"] - b2["extrapolated_s"]) / max(blend_s, 1e-9),
            "estimated_step_seconds": est}


def blend_row_evaluations(p):
    """How many (list entry, 16-pixel row) evaluations the blend backward made for the view whose buffers the
    pipeline `p` still holds: every entry of a tile's list is evaluated on the 4 x 4-pixel rows (four per 8 x 8
    quadrant) that the forward's verdict byte says its 1/255-alpha ellipse reaches, up to that row's last
    contributor.  One such evaluation is one step of one DPP row in blend_bwd_kernel's inner loop."""
    H, W = p.n_contrib.shape
    gx, gy = (W + 15) // 16, (H + 15) // 16
    nc = torch.zeros(gy * 16, gx * 16, dtype=torch.int32, device=p.n_contrib.device)
    nc[:H, :W] = p.n_contrib
    # y = (tile, quadrant, row, pixel) x = (tile, quadrant, row, pixel): the last contributor of every row
    last = nc.view(gy, 2, 2, 4, gx, 2, 2, 4).amax(dim=(3, 7)).permute(0, 3, 1, 4, 2, 5).reshape(gy * gx, 4, 4)
    ranges = p.ranges.long()
    lens = ranges[:, 1] - ranges[:, 0]
    R = int(lens.sum())
    if R == 0:
        return 0
    tile = torch.repeat_interleave(torch.arange(gy * gx, device=lens.device), lens)
    first = torch.repeat_interleave(ranges[:, 0], lens)
    order = torch.argsort(first, stable=True)           # the lists lie in tile order; keep it explicit
    tile, first = tile[order], first[order]
    pos = (torch.arange(R, device=lens.device) - first).to(torch.int32)       # position inside the tile's list
    verdict = p.keys_a.view(torch.uint8)[:4 * R].view(R, 4).to(torch.int32)    # one byte per quadrant, bit = row
    total = 0
    for row in range(4):
        reach = ((verdict >> row) & 1).bool()
        total += int((reach & (pos[:, None] < last[tile, :, row])).sum())
    return total


def main():
    global torch, dist
    argv = sys.argv[1:]
    args = parse(argv)
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        return launch_ranks(args, argv)       # before anything touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_run:
        return dry_run(args, world, rank)
    import torch as _torch
    import torch.distributed as _dist
    torch, dist = _torch, _dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # FDGS_BENCH_SHARE_GPU=1: rehearsal of the N-rank job on a one-GPU machine - every rank on cuda:0, collectives
    # over gloo staged through host memory (two RCCL ranks cannot share a device).  Not a measurement.
    share_gpu = os.environ.get("FDGS_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    grouped = world > 1 or os.environ.get("FDGS_DP_SINGLE_RANK") == "1"    # the latter: RCCL rehearsal on one GPU
    # RCCL prints a version banner to file descriptor 1 when its first communicator comes up; the contract of this
    # script is ONE line on stdout.  Everything written to descriptor 1 before the result line goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if grouped:
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            os.environ.setdefault("NCCL_ALGO", "Ring")      # see fdgs/dp.py: no packed fp32 in RCCL's ring sum kernels
            dist.init_process_group(backend="nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    from fdgs import _lib, dp, rasterizer
    from fdgs.renderer import render
    from fdgs.train_step import TrainStep

    from fdgs import deformation as _deformation
    split_is_default = _deformation.MATMUL == _lib.MATMUL_SPLIT_BF16
    g, net, deforms, cams, bg = build_scene(args, dev)
    step = TrainStep(g, deforms, bg)
    train_cams = cams[:20]
    B = resolve_global_batch(args, world)
    per_rank = B // world

    def batch(i, size=B):
        """The whole batch of iteration i, the same list on every rank (TrainStep shards it): `size` views, each
        with its own camera and time stamp (SynthCamera.time), walking the 20 training views."""
        return [train_cams[(i * size + j) % len(train_cams)] for j in range(size)]

    def sync():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if not grouped:
            return seconds
        t = torch.tensor([seconds], device=dev, dtype=torch.float64)
        dp.all_reduce_(t, dist.ReduceOp.MAX)
        return float(t.item())

    for i in range(args.warmup):
        step(batch(i), iteration=1 + i)
    sync()
    # a generation-2 collection of the cyclic garbage collector walks every live Python object (about 45 ms in a
    # process with torch loaded, more than ten training steps); the steps themselves create no cycles, so
    # everything alive after set-up is moved out of the collector's reach
    gc.collect()
    gc.freeze()
    with torch.no_grad():       # visible Gaussians of the first training view (recorded, not timed)
        visible = int((render(train_cams[rank % len(train_cams)], g, deforms, bg)["radii"] > 0).sum())
    sync()
    # HIP events over the timed region around the north-star kernel's call only; the other entry points are timed the
    # same way in a leg of their own below (events around all eleven calls of a step cost the step 2-3 %)
    _lib.PROFILE = {}
    _lib.PROFILE_ONLY = None if args.profile_all_calls else {"fdgs_raster_blend_bwd"}
    rendered = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(batch(args.warmup + i), iteration=1 + args.warmup + i)     # with the learning-rate schedule, as _run does
        rendered.append(rasterizer.LAST_NUM_RENDERED)
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    prof, _lib.PROFILE = _lib.PROFILE, None
    row_evals = None
    if rank == 0 and step.use_pipeline and step.pipeline.capacity > 0:
        try:
            torch.cuda.synchronize()
            row_evals = blend_row_evaluations(step.pipeline)        # of the last timed view
        except Exception:                                           # a diagnostic figure: the line must still come out
            row_evals = None
    if _lib.PROFILE_ONLY is not None:      # kernel_ms of every entry point: the same steps continued, not timed as a whole
        _lib.PROFILE, _lib.PROFILE_ONLY = {}, None
        for i in range(min(args.steps, 40)):
            step(batch(args.warmup + args.steps + i), iteration=1 + args.warmup + args.steps + i)
        sync()
        prof_all, _lib.PROFILE = _lib.PROFILE, None
        prof_all["fdgs_raster_blend_bwd"] = prof.get("fdgs_raster_blend_bwd", [])
        prof = prof_all

    # extra leg, not the headline: same step with the deformation matrix products on v_mfma_f32_32x32x2_f32
    # instead of the exact three-way bf16 split (both take, accumulate and return fp32)
    fp32_mfma = None
    if args.fp32_mfma_steps > 0 and split_is_default:
        from fdgs import deformation
        split_mode = deformation.MATMUL
        deformation.MATMUL = _lib.MATMUL_F32
        for i in range(2):
            step(batch(i))
        sync()
        t2 = time.perf_counter()
        for i in range(args.fp32_mfma_steps):
            step(batch(2 + i))
        sync()
        dt2 = max_over_ranks(time.perf_counter() - t2)
        deformation.MATMUL = split_mode
        fp32_mfma = args.fp32_mfma_steps / dt2

    # extra leg at one GPU: 8 views per iteration, the N = 1 point of the fixed-global-batch scaling curve
    batch8 = None
    if world == 1 and B == 1 and args.batch8_steps > 0:
        for i in range(2):
            step(batch(i, 8))
        sync()
        t3 = time.perf_counter()
        for i in range(args.batch8_steps):
            step(batch(2 + i, 8))
        sync()
        batch8 = args.batch8_steps / (time.perf_counter() - t3)

    # inference-only legs: deformation + forward rasterizer, frames strided over ranks, no collective
    fps = fps_d2h = fps_d2h_host = fps_pipelined = None
    if args.render_frames > 0:
        import numpy as np
        frames = list(range(rank, args.render_frames, world))
        with torch.no_grad():
            render(cams[20], g, deforms, bg)
            sync()
            t1 = time.perf_counter()
            for f in frames:
                cams[20].time = f / 300.0
                render(cams[20], g, deforms, bg)
            sync()
            fps = args.render_frames / max_over_ranks(time.perf_counter() - t1)
            # the reference's own FPS figure (render.py:74-89) includes, per frame, the 8-bit [H,W,3] copy of the image
            # on the host.  (a) fdgs.playback.render_sequence: conversion on the device, 3 bytes per pixel copied on a
            # copy stream into pinned memory (identical bytes, tests/test_render_gpu.py; allocation of the pinned
            # array included); (b) the reference's literal host code: the float
            # image copied and converted by numpy while the GPU waits.
            import copy
            from fdgs.playback import render_sequence
            views = []
            for f in frames:
                v = copy.copy(cams[20])
                v.time = f / 300.0
                views.append(v)
            render_sequence(views[:2], g, deforms, bg)
            sync()
            t1 = time.perf_counter()
            images, _ = render_sequence(views, g, deforms, bg)
            sync()
            fps_d2h = args.render_frames / max_over_ranks(time.perf_counter() - t1)
            del images
            # the same loop without the 8-bit frames: what the pipelining of consecutive frames is worth by itself
            # (render_fps above is the plain loop of render() calls, one host wait per frame with an idle GPU)
            sync()
            t1 = time.perf_counter()
            render_sequence(views, g, deforms, bg, to8b=False)
            sync()
            fps_pipelined = args.render_frames / max_over_ranks(time.perf_counter() - t1)

            def to8b(x):
                return (255 * np.clip(x.cpu().numpy(), 0, 1)).astype(np.uint8)
            host_frames = frames[:max(1, min(len(frames), 60))]
            images = [to8b(render(cams[20], g, deforms, bg)["render"]).transpose(1, 2, 0)]
            sync()
            t1 = time.perf_counter()
            for f in host_frames:
                cams[20].time = f / 300.0
                images.append(to8b(render(cams[20], g, deforms, bg)["render"]).transpose(1, 2, 0))
            sync()
            fps_d2h_host = len(host_frames) * world / max_over_ranks(time.perf_counter() - t1)
            del images

    # last leg of an N-rank run: the SAME batch of B views on rank 0 alone (no exchange step), the one-GPU point of
    # the strong-scaling curve measured inside this job.  The other ranks wait; the replicas differ afterwards,
    # so nothing that steps follows.
    one_gpu = one_gpu_error = None
    if world > 1 and args.one_gpu_steps > 0:
        sync()
        if rank == 0:
            try:
                with dp.local_only():
                    for i in range(2):
                        step(batch(i))
                    torch.cuda.synchronize()
                    t4 = time.perf_counter()
                    for i in range(args.one_gpu_steps):
                        step(batch(2 + i))
                    torch.cuda.synchronize()
                    one_gpu = args.one_gpu_steps / (time.perf_counter() - t4)
            except Exception as exc:       # a reference figure only: the measured line must still come out
                one_gpu_error = repr(exc)
        sync()

    if rank == 0:
        def avg_ms(name):
            ev = prof.get(name, [])
            # fdgs_raster_bin_sort is made in two parts per view (the counts before the wait for the list length,
            # the rest after): the figure is the two together
            views = len(prof.get("fdgs_raster_blend_fwd", [])) if name == "fdgs_raster_bin_sort" else len(ev)
            return sum(s.elapsed_time(e) for s, e in ev) / max(views, 1)
        per_call = {k: round(avg_ms(k), 4) for k in prof}
        R = sum(rendered) / max(len(rendered), 1)
        px = args.width * args.height
        # blend backward: 44 B per list entry read + 20 B per pixel read + 40 B per entry of gradient accumulation
        alg_bytes = 84.0 * R + 20.0 * px
        dur = avg_ms("fdgs_raster_blend_bwd") * 1e-3
        achieved = alg_bytes / dur / 1e9 if dur > 0 else 0.0
        # HBM traffic per launch from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # summarised by tools/pmc_summary.py into profiles/); only valid for the default workload
        traffic = None
        valu_roofline = None
        if (args.gaussians, args.width, args.height) == (300000, 1352, 1014):
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1:]:
                traffic = json.load(open(f)).get("blend_bwd_kernel", {}).get("traffic_bytes_per_launch")
            # vector-instruction issue: instructions per launch from the SQ counter pass of this same command
            # (tools/sq_summary.py), over the launch time measured here
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.json")))[-1:]:
                sq = json.load(open(f)).get("blend_bwd_kernel")
                if sq and dur > 0:
                    peak = sq["valu_issue_peak_ginst_per_s"]
                    rate = sq["valu_insts_per_launch"] / dur / 1e9
                    # peak: the issue rate a microbenchmark of independent v_fma_f32 reaches on this part
                    # (tools/microbench/valu_rate.hip); spec_peak: the 157.3 TFLOP/s fp32 vector figure = one wave64
                    # FMA per SIMD every 2 cycles at 2.4 GHz on 1024 SIMDs
                    spec = 157.3e12 / 2.0 / 64.0 / 1e9
                    valu_roofline = {"kernel": "blend_bwd_kernel", "bound": "valu-issue", "achieved": rate,
                                     "peak": peak, "unit": "G wave-instructions/s", "frac": rate / peak,
                                     "spec_peak": spec, "frac_of_spec_peak": rate / spec,
                                     "valu_insts_per_launch": sq["valu_insts_per_launch"],
                                     "counters_from": os.path.basename(f)}
                    if row_evals:
                        # the last timed view's (entry, 16-pixel row) evaluations, scaled to the mean list length of
                        # the timed views.  A wave instruction serves four rows at once, so a loop whose four rows
                        # were always busy would spend a quarter of its instructions per step on each evaluation;
                        # the figure includes the idle rows' share and everything outside the inner loop.
                        evals = row_evals * (R / max(rendered[-1], 1))
                        valu_roofline["entry_row_evaluations_per_launch"] = evals
                        valu_roofline["wave_insts_per_entry_row_evaluation"] = sq["valu_insts_per_launch"] / evals
        mlp_s = (avg_ms("fdgs_deform_fwd") + avg_ms("fdgs_deform_bwd")) * 1e-3 or 1.0
        mlp_flop = 3 * 187136.0 * args.gaussians          # fp32 semantics, forward + backward, per view
        out = {
            "metric": "train it/s", "value": args.steps / elapsed, "unit": "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": scaling_label(args, world),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: DyNeRF-shaped synthetic scene, single-view training step"
                                    if B == 1 else
                                    f"BASELINE configs[2]: DyNeRF-shaped synthetic scene, {B} views / time stamps per "
                                    f"iteration split over {world} GPU(s), RCCL gradient all-reduce")
                                   + " (fine stage: HexPlane deformation + tile rasterizer fwd+bwd + L1 loss and PSNR"
                                     " + grid regulariser + densification statistics + Adam)",
                       "gaussians": args.gaussians, "image": [args.width, args.height], "sh_degree": 3,
                       "global_batch_views": B, "views_per_gpu": per_rank, "visible_gaussians": visible,
                       "tile_list_entries": R,
                       "parallelism": f"dp{world}",
                       "deform_matmul": ("fp32 in / fp32 accumulate / fp32 out; products formed from an exact "
                                         "three-way bf16 split of both operands, six bf16 MFMA terms (error below "
                                         "one fp32 rounding per product)" if split_is_default else "fp32 MFMA")},
            "views_per_s": B * args.steps / elapsed,
            "train_it_s_fp32_mfma": fp32_mfma,
            "train_it_s_global_batch_8": batch8,
            # N-rank runs: the same global batch timed on rank 0 alone at the end of this job, and value over it
            "train_it_s_same_batch_one_gpu": one_gpu,
            "speedup_over_one_gpu": (args.steps / elapsed / one_gpu) if one_gpu else None,
            "render_fps": fps,
            "render_fps_pipelined": fps_pipelined,
            "render_fps_with_d2h": fps_d2h,
            "render_fps_with_d2h_host_conversion": fps_d2h_host,
            "kernel_ms": per_call,
            "kernel_ms_note": "HIP events around each C-ABI call on the stream it is enqueued on; fdgs_plane_regulation "
                              "runs on a second stream beside fdgs_deform_fwd (FDGS_ASIDE), so its figure is time in "
                              "flight while sharing the chip (0.026 ms alone), not time added to the step",
            # deformation MLP, forward + backward: 3 x 187,136 flop per point in fp32 semantics.  The split mode
            # issues six bf16 MFMA terms per product: issued bf16 flop = 6 x that, against the dense bf16 peak;
            # the fp32-equivalent figure against the fp32 matrix-core peak is kept beside it.  The two calls
            # include the HexPlane gather / scatter.
            "roofline_deform_mlp": {"bound": "mfma",
                                    "achieved": (6 if split_is_default else 1) * mlp_flop / mlp_s / 1e12,
                                    "peak": 2500.0 if split_is_default else 157.3,
                                    "unit": "TFLOP/s issued (bf16 terms)" if split_is_default else "TFLOP/s (fp32 MFMA)",
                                    "frac": ((6 * mlp_flop / mlp_s / 1e12 / 2500.0) if split_is_default
                                             else mlp_flop / mlp_s / 1e12 / 157.3),
                                    "fp32_equivalent_tflops": mlp_flop / mlp_s / 1e12,
                                    "fp32_equivalent_frac_of_157.3": mlp_flop / mlp_s / 1e12 / 157.3,
                                    "calls": "fdgs_deform_fwd + fdgs_deform_bwd"},
            "roofline": {"kernel": "blend_bwd_kernel", "bound": "hbm", "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dur * 1e3},
            "roofline_valu": valu_roofline,
        }
        if one_gpu_error:
            out["train_it_s_same_batch_one_gpu_error"] = one_gpu_error
        if share_gpu:
            out["rehearsal"] = "all ranks on one GPU over gloo: not a measurement"
        # the HBM-bound stages beside the north-star kernel: algorithmic bytes per call over the call's time
        # (N Gaussians, V visible, P parameters)
        N, V = float(args.gaussians), float(visible)
        n_param = float(sum(p.numel() for p in step.params))

        def hbm(name, nbytes):
            t = avg_ms(name) * 1e-3
            return {"call": name, "bound": "hbm", "achieved": nbytes / t / 1e9 if t > 0 else 0.0, "peak": 8000.0,
                    "unit": "GB/s", "frac": (nbytes / t / 1e9 / 8000.0) if t > 0 else 0.0}
        out["roofline_hbm_calls"] = [
            hbm("fdgs_raster_preprocess_bwd", 40.0 * V + 236.0 * N + 248.0 * N),
            hbm("fdgs_adam_step", 28.0 * n_param),
            hbm("fdgs_l1_loss_fwd", 12.0 * 3 * px),     # two images read, the gradient image written
            hbm("fdgs_raster_blend_fwd", 44.0 * R + 28.0 * px),
            hbm("fdgs_raster_bin_sort", 24.0 * V + 12.0 * R + 24.0 * R + 8.0 * R),
        ]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, g, net, train_cams[0], bg)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
