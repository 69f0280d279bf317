"""Seeded synthetic DyNeRF-shaped scenes (no dataset files needed).

Shapes and distributions follow what the reference's DyNeRF run produces:
image 1352x1014 (src/dataset/neural_3D_dataset_NDC.py:220-223), 300 frames with
time = idx/300 (:266,335), scene box [-2.5,-2,-1]..[2.5,2,1]
(configs/train/dynerf.yaml:69-70), SH degree 3 (:80), seed 6666 (train.py:27),
znear 0.01 / zfar 100 (src/scene/camera.py:42-43), opacity initialised around
logit(0.1) (src/model/gaussian_model.py:61).  The focal length (730 px at 1352
wide) is an assumption: the reference reads it from poses_bounds.npy.
"""
import math

import numpy as np
import torch

SCENE_MIN = (-2.5, -2.0, -1.0)
SCENE_MAX = (2.5, 2.0, 1.0)
SEED = 6666


def world_to_view(R, t):
    """4x4 float32 world->view matrix from a camera-to-world rotation R and a
    world->camera translation t, as src/utils/graphics_utils.py:59-70 builds it
    (translate = 0, scale = 1)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return np.float32(np.linalg.inv(np.linalg.inv(Rt)))


def perspective(znear, zfar, fovx, fovy):
    """src/utils/graphics_utils.py:72-92."""
    tan_y = math.tan(fovy / 2)
    tan_x = math.tan(fovx / 2)
    top, right = tan_y * znear, tan_x * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class SynthCamera:
    """Carries the fields render() reads from the reference's Camera
    (src/scene/camera.py:6-53): FoVx, FoVy, image_width/height, time,
    world_view_transform (W2C transposed), full_proj_transform, camera_center,
    original_image."""

    def __init__(self, R, T, fovx, fovy, width, height, time, uid=0, device="cpu"):
        self.uid = uid
        self.R, self.T = R, T
        self.FoVx, self.FoVy = fovx, fovy
        self.image_width, self.image_height = int(width), int(height)
        self.time = float(time)
        self.znear, self.zfar = 0.01, 100.0
        # the same operations on the same (transposed, non-contiguous) views as src/scene/camera.py:48-53, so that
        # the four tensors are bit-identical to the reference's; made contiguous afterwards
        wvt = torch.tensor(world_to_view(R, T)).transpose(0, 1)
        proj = perspective(self.znear, self.zfar, fovx, fovy).transpose(0, 1)
        self.full_proj_transform = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
        self.camera_center = wvt.inverse()[3, :3].contiguous()
        self.world_view_transform = wvt.contiguous()
        self.projection_matrix = proj
        self.original_image = None
        self.to(device)

    def to(self, device):
        """Pre-stage the three camera tensors on the device once, so render()
        does no per-call host->device copies."""
        self.world_view_transform = self.world_view_transform.to(device)
        self.full_proj_transform = self.full_proj_transform.to(device)
        self.camera_center = self.camera_center.to(device)
        if self.original_image is not None:
            self.original_image = self.original_image.to(device)
        return self


def look_at(cam_pos, target=(0.0, 0.0, 0.0)):
    """Camera-to-world rotation (columns right, down, forward) and world->camera translation."""
    c = np.asarray(cam_pos, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - c
    fwd /= np.linalg.norm(fwd)
    right = np.cross(np.array([0.0, 1.0, 0.0]), fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)
    T = -R.T @ c
    return R, T


def make_cameras(n_cams=21, width=1352, height=1014, focal=None, n_frames=300, device="cpu"):
    """Forward-facing rig: n_cams poses on a 1.0 x 0.6 arc at z = -4 looking at the origin."""
    focal = 730.0 * width / 1352.0 if focal is None else focal
    fovx = 2 * math.atan(width / (2 * focal))
    fovy = 2 * math.atan(height / (2 * focal))
    cams = []
    for i in range(n_cams):
        u = i / max(n_cams - 1, 1) - 0.5
        pos = (1.0 * u, 0.6 * math.sin(2 * math.pi * u) * 0.5, -4.0)
        R, T = look_at(pos)
        cams.append(SynthCamera(R, T, fovx, fovy, width, height, time=(i * 7 % n_frames) / n_frames,
                                uid=i, device=device))
    return cams


class GaussianParams:
    """Tensor-layout contract of the reference's GaussianModel
    (src/model/gaussian_model.py:12-89): raw (pre-activation) parameters
    _xyz [N,3], _features_dc [N,1,3], _features_rest [N,15,3], _scaling [N,3]
    (log), _rotation [N,4] (w first, un-normalised), _opacity [N,1] (logit)."""

    def __init__(self, n, seed=SEED, device="cpu", scale_mult=1.0, requires_grad=True, sh_degree=3):
        g = torch.Generator().manual_seed(seed)
        lo = torch.tensor(SCENE_MIN)
        hi = torch.tensor(SCENE_MAX)
        xyz = lo + (hi - lo) * torch.rand(n, 3, generator=g)
        scaling = math.log(0.01 * scale_mult) + 0.5 * torch.randn(n, 3, generator=g)
        rotation = torch.randn(n, 4, generator=g)
        opacity = 1.5 * torch.randn(n, 1, generator=g)
        f_dc = torch.randn(n, 1, 3, generator=g)
        f_rest = 0.1 * torch.randn(n, 15, 3, generator=g)

        def P(t):
            return t.to(device).contiguous().requires_grad_(requires_grad)
        self._xyz, self._scaling, self._rotation, self._opacity = P(xyz), P(scaling), P(rotation), P(opacity)
        self._features_dc, self._features_rest = P(f_dc), P(f_rest)
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree
        self.scaling_activation = torch.exp
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def get_covariance(self, scaling_modifier=1):
        """Upper triangle of R S S^T R^T as src/model/gaussian_model.py:38-42,92-93
        (rotation normalised inside, src/utils/general_utils.py:101-133)."""
        s = scaling_modifier * torch.exp(self._scaling)
        q = self._rotation / self._rotation.norm(dim=1, keepdim=True)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
        L = R * s[:, None, :]
        cov = L @ L.transpose(1, 2)
        return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]


DYNERF_NET_KWARGS = dict(   # configs/train/dynerf.yaml:85-112
    net_width=128, timebase_pe=4, defor_depth=0, posebase_pe=10, scale_rotation_pe=2, opacity_pe=2,
    grid_pe=0, no_grid=False, bounds=1.6,
    kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 16,
                    "resolution": [64, 64, 64, 150]},
    multires=[1, 2], empty_voxel=False, static_mlp=False)


def perturb_deformation_(module, seed=SEED + 1, std=0.01):
    """After the reference-style init (time planes exactly one), add N(0, std^2)
    so that time planes differ from 1 and every gradient is non-trivial."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            if p.requires_grad:
                p.add_((std * torch.randn(p.shape, generator=g)).to(p.device))
    return module
