"""Worker of tests/test_dp_gpu.py: runs TrainStep on a global batch of two views for a few steps with the
optimizer update replaced by a recorder, and saves every step's complete gradient vector and the statistics.
Launched once with two ranks (gloo, both on cuda:0) and once alone; the gradients must agree."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "4d-gaussian-splatting_amd")]


def main(out_dir):
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import TrainStep, DeformHolder
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    # FDGS_TEST_BACKEND=nccl with FDGS_DP_SINGLE_RANK=1: the exchange step through RCCL in a group of one rank
    backend = os.environ.get("FDGS_TEST_BACKEND", "gloo")
    grouped = world > 1 or os.environ.get("FDGS_DP_SINGLE_RANK") == "1"
    views_per_step = int(os.environ.get("FDGS_TEST_VIEWS", "2"))
    tag = os.environ.get("FDGS_TEST_TAG", "")
    if grouped:
        dist.init_process_group(backend=backend, **({"device_id": dev} if backend == "nccl" else {}))
    torch.manual_seed(ss.SEED + rank)            # different RNG state per rank: TrainStep must equalise the replicas
    n, w, h = (int(v) for v in os.environ.get("FDGS_TEST_SCENE", "6000,160,120").split(","))
    g = ss.GaussianParams(n, device=dev, scale_mult=3.0)
    net = ss.perturb_deformation_(DeformationFields(ss.DYNERF_NET_KWARGS)).to(dev)
    net.deformation_net.set_aabb(list(ss.SCENE_MAX), list(ss.SCENE_MIN))
    deforms = DeformHolder(net)
    cams = ss.make_cameras(6, w, h, device=dev)
    bg = torch.ones(3, device=dev)
    g2 = ss.GaussianParams(n, seed=ss.SEED + 100, device=dev, requires_grad=False, scale_mult=3.0)
    with torch.no_grad():
        for cam in cams:
            cam.original_image = render(cam, g2, deforms, bg, stage="coarse")["render"].clamp(0, 1).clone()
    step = TrainStep(g, deforms, bg)
    recorded = []

    def record():
        recorded.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1)
                                   for p in step.params]).cpu())
    step.optimizer.step = record
    for i in range(3):
        step([cams[(2 * i + v) % len(cams)] for v in range(views_per_step)])
    torch.cuda.synchronize()
    out = {f"grad{i}": t for i, t in enumerate(recorded)}
    out.update(accum=step.xyz_gradient_accum.cpu(), denom=step.denom.cpu(), maxr=step.max_radii2D.cpu(),
               w=net.deformation_net.pos_deform[1].weight.detach().cpu())
    torch.save(out, os.path.join(out_dir, f"w{world}_r{rank}{tag}.pt"))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
