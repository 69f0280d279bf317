#!/usr/bin/env python
"""Benchmark of the 4DGS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one training iteration of the reference's loop body
(src/runner/fdgs_trainer.py:112-205) on a synthetic DyNeRF-shaped scene: per
view  deformation forward -> activations -> rasterizer forward -> L1 loss ->
backward through both -> densification statistics -> Adam step.  At N GPUs the
global batch is N views, one per rank (weak scaling), gradients and statistics
are all-reduced over RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "4d-gaussian-splatting_amd"))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=25,
                    help="untimed steps; the default walks all 20 training views once so that buffer sizes settle")
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--width", type=int, default=1352)
    ap.add_argument("--height", type=int, default=1014)
    ap.add_argument("--render-frames", type=int, default=60, help="frames of the inference-only FPS leg (0 = skip)")
    ap.add_argument("--scale-mult", type=float, default=1.0, help="multiplier on the Gaussian scales (0.6 for the 2M case)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp32-mfma-steps", type=int, default=10,
                    help="steps of the extra leg that forms the deformation matrix products with fp32 MFMA (0 = skip)")
    ap.add_argument("--cpu-tile-stride", type=int, default=48, help="cpu baseline blends every k-th tile")
    return ap.parse_args()


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


def cpu_baseline(args, g, net, cam, bg):
    """The CPU oracle (oracle/deform_ref.py + oracle/raster_ref.py, torch CPU) on the same inputs:
    deformation forward+backward on all points, rasterizer preprocess/sort on all points, blending
    forward+backward on every k-th tile, extrapolated by the number of pixel-Gaussian pairs."""
    from oracle import deform_ref, raster_ref as rr
    torch.set_num_threads(min(16, os.cpu_count() or 1))
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
    t0 = time.time()
    pts, sc, ro, op, sh = deform_ref.deform_forward(sd, ins[0], ins[1], ins[2], ins[3], shs,
                                                    torch.full((N, 1), float(cam.time)))
    t_def_f = time.time() - t0
    m2 = torch.zeros(N, 3, requires_grad=True)
    t0 = time.time()
    pre = rr.preprocess(pts, torch.exp(sc), torch.nn.functional.normalize(ro), torch.sigmoid(op), sh, S, m2)
    keys, vals, _ = rr.build_keys(pre, S)
    skeys, svals = rr.sort_keys(keys, vals)
    gx, gy = rr.tile_grid(S)
    ranges = rr.tile_ranges(skeys, gx * gy)
    t_pre = time.time() - t0
    tiles = list(range(0, gx * gy, args.cpu_tile_stride))
    lens = (ranges[:, 1] - ranges[:, 0]).to(torch.float64)
    frac = float(lens[tiles].sum() / max(float(lens.sum()), 1.0))
    t0 = time.time()
    color, _, _, _ = rr.blend(pre, svals, ranges, S, tiles)
    t_blend_f = time.time() - t0
    mask = torch.zeros(gy * 16, gx * 16, dtype=torch.bool)
    for t in tiles:
        ty, tx = divmod(t, gx)
        mask[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
    mask = mask[:args.height, :args.width]
    loss = (torch.abs(color - gt) * mask).sum() / (3.0 * args.height * args.width)
    mids = [pre[k] for k in ("px", "py", "conA", "conB", "conC", "opacity", "rgb")]
    t0 = time.time()
    gmid = torch.autograd.grad(loss, mids, retain_graph=True)     # blending backward, sampled tiles only
    t_blend_b = time.time() - t0
    t0 = time.time()
    torch.autograd.backward(mids, gmid)                            # preprocess + deformation backward, all points
    t_rest_b = time.time() - t0
    t0 = time.time()                                               # grid regulariser of the fine stage, fwd + bwd
    from fdgs.train_step import TrainStep
    deform_ref.compute_regulation(sd, *TrainStep.DYNERF_REGULATION).backward()
    t_reg = time.time() - t0
    est = t_def_f + t_pre + (t_blend_f + t_blend_b) / max(frac, 1e-9) + t_rest_b + t_reg
    measured = t_def_f + t_pre + t_blend_f + t_blend_b + t_rest_b + t_reg
    return {"value": 1.0 / est, "unit": "it/s", "cores": cores, "kind": "port",
            "sample": (f"deformation fwd+bwd, grid regulariser and rasterizer preprocess/sort on all {N} points; blending fwd+bwd on "
                       f"every {args.cpu_tile_stride}th tile ({len(tiles)} of {gx * gy} tiles, {frac:.4f} of the "
                       f"tile-list entries), blending time divided by that fraction; measured {measured:.1f} s"),
            "seconds_measured": measured, "estimated_step_seconds": est}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    grouped = world > 1 or os.environ.get("FDGS_DP_SINGLE_RANK") == "1"    # the latter: RCCL rehearsal on one GPU
    if grouped:
        dist.init_process_group(backend="nccl", device_id=dev)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from fdgs import _lib, rasterizer
    from fdgs.renderer import render
    from fdgs.train_step import TrainStep

    from fdgs import deformation as _deformation
    split_is_default = _deformation.MATMUL == _lib.MATMUL_SPLIT_BF16
    g, net, deforms, cams, bg = build_scene(args, dev)
    step = TrainStep(g, deforms, bg)
    train_cams = cams[:20]

    def batch(i):
        return [train_cams[(i * world + r) % len(train_cams)] for r in range(world)]

    def sync():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

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
    _lib.PROFILE = {}
    rendered = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(batch(args.warmup + i), iteration=1 + args.warmup + i)     # with the learning-rate schedule, as _run does
        rendered.append(rasterizer.LAST_NUM_RENDERED)
    sync()
    elapsed = time.perf_counter() - t0
    prof, _lib.PROFILE = _lib.PROFILE, None
    if grouped:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
        dt2 = time.perf_counter() - t2
        deformation.MATMUL = split_mode
        if grouped:
            t = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        fp32_mfma = world * args.fp32_mfma_steps / dt2

    # inference-only leg: deformation + forward rasterizer, frames strided over ranks, no collective
    fps = None
    if args.render_frames > 0:
        frames = list(range(rank, args.render_frames, world))
        with torch.no_grad():
            render(cams[20], g, deforms, bg)
            sync()
            t1 = time.perf_counter()
            for f in frames:
                cams[20].time = f / 300.0
                render(cams[20], g, deforms, bg)
            sync()
            dt = time.perf_counter() - t1
        if grouped:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        fps = args.render_frames / dt

    if rank == 0:
        def avg_ms(name):
            ev = prof.get(name, [])
            return sum(s.elapsed_time(e) for s, e in ev) / max(len(ev), 1)
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
            # what actually bounds this kernel: vector-instruction issue.  Instructions per launch from the SQ
            # counter pass of this same command (tools/sq_summary.py), over the launch time measured here, against
            # 1024 SIMDs x one wave64 vector instruction per 4 cycles at 2.4 GHz
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.json")))[-1:]:
                sq = json.load(open(f)).get("blend_bwd_kernel")
                if sq and dur > 0:
                    peak = sq["valu_issue_peak_ginst_per_s"]
                    rate = sq["valu_insts_per_launch"] / dur / 1e9
                    valu_roofline = {"kernel": "blend_bwd_kernel", "bound": "valu-issue", "achieved": rate,
                                     "peak": peak, "unit": "G wave-instructions/s", "frac": rate / peak,
                                     "valu_insts_per_launch": sq["valu_insts_per_launch"]}
        out = {
            "metric": "train it/s", "value": world * args.steps / elapsed, "unit": "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: DyNeRF-shaped synthetic scene, single-view training step "
                                   "per GPU (fine stage: HexPlane deformation + tile rasterizer fwd+bwd + L1 loss and PSNR + grid regulariser + densification statistics + Adam)",
                       "gaussians": args.gaussians, "image": [args.width, args.height], "sh_degree": 3,
                       "global_batch_views": world, "views_per_gpu": 1, "visible_gaussians": visible,
                       "tile_list_entries": R,
                       "parallelism": f"dp{world}",
                       "deform_matmul": ("fp32 in / fp32 accumulate / fp32 out; products formed from an exact "
                                         "three-way bf16 split of both operands, six bf16 MFMA terms (error below "
                                         "one fp32 rounding per product)" if split_is_default else "fp32 MFMA")},
            "train_it_s_fp32_mfma": fp32_mfma,
            "render_fps": fps,
            "kernel_ms": per_call,
            # deformation MLP, forward + backward: 3 x 187,136 flop per point (fp32 semantics) over the two calls,
            # against the dense fp32 matrix-core peak (157.3 TFLOP/s); the calls include the HexPlane gather/scatter
            "roofline_deform_mlp": (lambda t: {"bound": "mfma", "achieved": 3 * 187136.0 * args.gaussians / t / 1e12,
                                               "peak": 157.3, "unit": "TFLOP/s (fp32-equivalent)",
                                               "frac": 3 * 187136.0 * args.gaussians / t / 1e12 / 157.3,
                                               "calls": "fdgs_deform_fwd + fdgs_deform_bwd"})(
                (avg_ms("fdgs_deform_fwd") + avg_ms("fdgs_deform_bwd")) * 1e-3 or 1.0),
            "roofline": {"kernel": "blend_bwd_kernel", "bound": "hbm", "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dur * 1e3},
            "roofline_valu": valu_roofline,
        }
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
            hbm("fdgs_l1_loss_bwd", 12.0 * 3 * px),
            hbm("fdgs_raster_blend_fwd", 44.0 * R + 28.0 * px),
        ]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, g, net, train_cams[0], bg)
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
