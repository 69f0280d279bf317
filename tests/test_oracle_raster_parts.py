"""The reference-side arithmetic that feeds the rasterizer, against tests/golden/raster_parts.npz (outputs of the
reference's own Camera / getWorld2View2 / getProjectionMatrix, eval_sh and GaussianModel.get_covariance, generated
by tests/golden/make_raster_parts_golden.py).  Checked here: the oracle's restatements (oracle/raster_ref.py) and
the host-side mirrors shipped in the package (fdgs/scene_synth.py, fdgs/sh.py).  The rasterizer stages after
these live in an external package the reference does not vendor and stay unpinned."""
import os

import numpy as np
import pytest
import torch

from oracle import raster_ref as rr

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster_parts.npz"))
N_CAMS = GOLD["cam_R"].shape[0]


@pytest.mark.parametrize("i", range(N_CAMS))
def test_camera_tensors(i):
    from fdgs import scene_synth as ss
    R, T, (fx, fy) = GOLD["cam_R"][i], GOLD["cam_T"][i], GOLD["cam_fov"][i]
    wvt, full, center = rr.camera_tensors(R, T, float(fx), float(fy))
    assert np.array_equal(wvt.numpy(), GOLD["cam_world_view_transform"][i])
    assert np.array_equal(full.numpy(), GOLD["cam_full_proj_transform"][i])
    assert np.array_equal(center.numpy(), GOLD["cam_camera_center"][i])
    assert np.array_equal(rr.projection_matrix(0.01, 100.0, float(fx), float(fy)).transpose(0, 1).numpy(),
                          GOLD["cam_projection_matrix"][i])
    cam = ss.SynthCamera(R, T, float(fx), float(fy), 6, 4, 0.25)
    assert np.array_equal(cam.world_view_transform.numpy(), GOLD["cam_world_view_transform"][i])
    assert np.array_equal(cam.projection_matrix.numpy(), GOLD["cam_projection_matrix"][i])
    assert np.array_equal(cam.full_proj_transform.numpy(), GOLD["cam_full_proj_transform"][i])
    assert np.array_equal(cam.camera_center.numpy(), GOLD["cam_camera_center"][i])


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colour(deg):
    from fdgs.sh import eval_sh
    feats = torch.from_numpy(GOLD["sh_features"])
    xyz = torch.from_numpy(GOLD["sh_xyz"])
    campos = torch.from_numpy(GOLD["sh_campos"])
    d = xyz - campos
    d = d / d.norm(dim=1, keepdim=True)
    want = GOLD["sh_rgb_deg%d" % deg]
    got = rr.eval_sh_color(deg, feats, d[:, 0:1], d[:, 1:2], d[:, 2:3])
    assert np.abs(got.numpy() - want).max() <= 1e-6 * max(np.abs(want).max(), 1.0)
    mirror = eval_sh(deg, feats.transpose(1, 2).reshape(-1, 3, 16), d)
    assert np.abs(mirror.numpy() - want).max() <= 1e-6 * max(np.abs(want).max(), 1.0)
    assert np.abs(torch.clamp_min(mirror + 0.5, 0.0).numpy() - GOLD["sh_color_deg%d" % deg]).max() <= 1e-6 * 4


@pytest.mark.parametrize("tag,mod", [("1", 1.0), ("07", 0.7)])
def test_covariance(tag, mod):
    from fdgs import scene_synth as ss
    sc = torch.from_numpy(GOLD["cov_scaling"])
    ro = torch.from_numpy(GOLD["cov_rotation"])
    want = GOLD["cov_mod" + tag]
    got = torch.stack(rr.covariance3d(torch.exp(sc), torch.nn.functional.normalize(ro), mod), dim=1).numpy()
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    g = ss.GaussianParams(sc.shape[0], requires_grad=False)
    g._scaling, g._rotation = sc, ro
    assert np.abs(g.get_covariance(mod).numpy() - want).max() <= 2e-6 * np.abs(want).max()
