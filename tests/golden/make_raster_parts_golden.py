"""Generates tests/golden/raster_parts.npz: outputs of the reference-side functions that feed the rasterizer -
the only parts of that path whose arithmetic lives in the reference itself (the rasterizer proper is an external,
absent package):
  * Camera (src/scene/camera.py:6-53) with getWorld2View2 / getProjectionMatrix (src/utils/graphics_utils.py:59-92):
    world_view_transform, projection_matrix, full_proj_transform, camera_center
  * eval_sh (src/utils/sh_utils.py:57-112), degrees 0-3, and the +0.5 / clamp of gaussian_renderer/__init__.py:95-96
  * GaussianModel.get_covariance (src/model/gaussian_model.py:43-50,92-93 with build_scaling_rotation /
    strip_symmetric, src/utils/general_utils.py:86-133)
Run in the authoring container only: python tests/golden/make_raster_parts_golden.py
Absent third-party modules the import chain names (plyfile, simple_knn, tkinter, diff_gaussian_rasterization,
torchvision, cv2) are empty placeholders; the
name `torch` inside src.utils.general_utils is bound to a proxy that rewrites device="cuda" to "cpu"."""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class CpuTorch:
    def __getattr__(self, name):
        obj = getattr(torch, name)
        if not callable(obj) or isinstance(obj, type):
            return obj

        def call(*a, **k):
            if k.get("device") == "cuda":
                k["device"] = "cpu"
            return obj(*a, **k)
        return call


def main():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules.setdefault(name, m)
        return m
    stub("tkinter", W="w")
    stub("plyfile", PlyData=object, PlyElement=object)
    stub("simple_knn", _C=stub("simple_knn._C", distCUDA2=None))
    stub("diff_gaussian_rasterization", GaussianRasterizationSettings=object, GaussianRasterizer=object)
    stub("torchvision"); stub("torchvision.transforms"); stub("cv2")     # named by src/scene/__init__.py's chain
    sys.path.insert(0, "/root/reference")
    import src.utils.general_utils as gen
    from src.utils.sh_utils import eval_sh
    from src.scene.camera import Camera
    from src.model.gaussian_model import GaussianModel
    gen.torch = CpuTorch()

    rng = np.random.default_rng(6666)
    out = {}
    # cameras: random rotations (QR of a Gaussian matrix, det +1), translations, fields of view
    Rs, Ts, fovs = [], [], []
    for i in range(5):
        q, r = np.linalg.qr(rng.standard_normal((3, 3)))
        q = q * np.sign(np.diag(r))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        Rs.append(q); Ts.append(rng.standard_normal(3) * 2.0)
        fovs.append((0.6 + 0.25 * i, 0.5 + 0.2 * i))
    Rs[0] = np.eye(3); Ts[0] = np.array([0.0, 0.0, 4.0]); fovs[0] = (2 * math.atan(1352 / (2 * 730.0)), 2 * math.atan(1014 / (2 * 730.0)))
    out["cam_R"] = np.stack(Rs); out["cam_T"] = np.stack(Ts); out["cam_fov"] = np.array(fovs)
    wvt, proj, full, center = [], [], [], []
    for R, T, (fx, fy) in zip(Rs, Ts, fovs):
        cam = Camera(colmap_id=0, R=R, T=T, FoVx=fx, FoVy=fy, image=torch.zeros(3, 4, 6), gt_alpha_mask=None,
                     image_name="x", uid=0, data_device="cpu", time=0.25)
        wvt.append(cam.world_view_transform.numpy()); proj.append(cam.projection_matrix.numpy())
        full.append(cam.full_proj_transform.numpy()); center.append(cam.camera_center.numpy())
        assert (cam.image_width, cam.image_height) == (6, 4)
    out["cam_world_view_transform"] = np.stack(wvt); out["cam_projection_matrix"] = np.stack(proj)
    out["cam_full_proj_transform"] = np.stack(full); out["cam_camera_center"] = np.stack(center)

    # spherical harmonics, laid out as the renderer passes them (gaussian_renderer/__init__.py:92-96)
    n = 96
    feats = torch.from_numpy(rng.standard_normal((n, 16, 3)).astype(np.float32))       # get_features layout
    xyz = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32) * 2)
    campos = torch.from_numpy(center[1])
    out["sh_features"] = feats.numpy(); out["sh_xyz"] = xyz.numpy(); out["sh_campos"] = campos.numpy()
    shs_view = feats.transpose(1, 2).view(-1, 3, 16)
    dir_pp = xyz - campos.repeat(n, 1)
    dirs = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    for deg in range(4):
        sh2rgb = eval_sh(deg, shs_view, dirs)
        out["sh_rgb_deg%d" % deg] = sh2rgb.numpy()
        out["sh_color_deg%d" % deg] = torch.clamp_min(sh2rgb + 0.5, 0.0).numpy()

    # 3-D covariance
    g = GaussianModel(sh_degree=3, percent_dense=0.01)
    g._scaling = torch.from_numpy((rng.standard_normal((n, 3)) * 0.7 - 2.0).astype(np.float32))
    g._rotation = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32))
    out["cov_scaling"] = g._scaling.numpy(); out["cov_rotation"] = g._rotation.numpy()
    for tag, mod in (("1", 1.0), ("07", 0.7)):
        out["cov_mod" + tag] = g.get_covariance(mod).numpy()
    path = os.path.join(HERE, "raster_parts.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
