"""Seeded rasterizer test scenes shared by the CPU and GPU tests."""
import math

import torch

from fdgs import scene_synth as ss
from oracle import raster_ref as rr

CASES = {
    # name: (n_gaussians, width, height, scale_mult, sh_degree, bg)
    "small": (3000, 128, 100, 4.0, 3, (1.0, 1.0, 1.0)),
    "ragged": (1500, 70, 45, 8.0, 2, (0.0, 0.3, 0.7)),     # partial edge tiles, big splats, early termination
    "deg0": (800, 64, 64, 3.0, 0, (0.0, 0.0, 0.0)),
    "medium": (20000, 256, 256, 2.0, 3, (1.0, 1.0, 1.0)),
    # Gaussian counts on the edges of the preprocess workgroup (256) and of the scan chunk (2048)
    "n256": (256, 48, 32, 5.0, 1, (0.2, 0.2, 0.2)),
    "n2049": (2049, 96, 80, 4.0, 3, (1.0, 1.0, 1.0)),
    "n4096": (4096, 112, 64, 3.0, 2, (0.0, 0.0, 0.0)),
}


def make_case(name, cam_index=1):
    n, w, h, sm, deg, bg = CASES[name]
    cam = ss.make_cameras(3, w, h)[cam_index]
    g = ss.GaussianParams(n, scale_mult=sm, requires_grad=False)
    means3D = g._xyz.clone()
    # a few Gaussians behind / very near the camera and far outside the frustum
    means3D[0] = torch.tensor([0.0, 0.0, -4.5])
    means3D[1] = torch.tensor([0.0, 0.0, -3.9])
    means3D[2] = torch.tensor([30.0, 0.0, 0.0])
    opac = torch.sigmoid(g._opacity)
    opac[3:40] = 0.999            # force alpha clamp + early termination
    scales = torch.exp(g._scaling)
    scales[3:40] *= 6.0
    rots = torch.nn.functional.normalize(g._rotation)
    shs = g.get_features.clone()
    S = rr.Settings(h, w, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.tensor(bg), 1.0,
                    cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center)
    return dict(means3D=means3D, shs=shs, opacities=opac, scales=scales, rotations=rots, S=S)
