"""Self-checks of oracle/raster_ref.py (the reference holds no fixtures for the
rasterizer, so the oracle is pinned by its own invariants): structural
properties of keys / sort / ranges, float64 agreement, autograd vs central
differences, and the BASELINE config-0 plumbing run (50k Gaussians, 256x256,
deformation + rasterizer forward/backward on CPU)."""
import math

import pytest
import torch

import raster_cases as rc
from oracle import deform_ref, raster_ref as rr


def test_keys_sort_ranges_invariants():
    c = rc.make_case("small")
    color, radii, depth, aux = rr.rasterize(c["means3D"], None, c["shs"], c["opacities"], c["scales"],
                                            c["rotations"], c["S"], return_aux=True)
    S = c["S"]
    gx, gy = rr.tile_grid(S)
    keys, skeys, svals, ranges = aux["keys"], aux["sorted_keys"], aux["sorted_vals"], aux["ranges"]
    R = keys.numel()
    assert R == int(aux["pre"]["tiles_touched"].sum()) == int(aux["offsets"][-1])
    assert bool((skeys[1:] >= skeys[:-1]).all())                       # sortedness
    assert torch.equal(torch.sort(keys).values, skeys)                  # permutation of the input
    tile = skeys >> 32
    assert int(tile.max()) < gx * gy
    lens = (ranges[:, 1] - ranges[:, 0]).long()
    assert int(lens.sum()) == R                                         # ranges partition the list
    for t in torch.nonzero(lens)[:50, 0].tolist():
        assert bool((tile[ranges[t, 0]:ranges[t, 1]] == t).all())
    # depth order inside a tile, ties broken by Gaussian index (stable sort of index-ordered duplicates)
    d = aux["pre"]["depth"][svals.long()]
    same = tile[1:] == tile[:-1]
    assert bool((d[1:][same] >= d[:-1][same]).all())
    tie = same & (d[1:] == d[:-1])
    assert bool((svals[1:][tie] > svals[:-1][tie]).all())
    # culled Gaussians (behind the camera, outside the frustum) have radius 0 and no keys
    assert int(radii[0]) == 0 and int(radii[2]) == 0
    assert color.shape == (3, S.image_height, S.image_width) and depth.shape == (1, S.image_height, S.image_width)
    assert bool((aux["final_T"] >= rr.T_STOP * 0.99).all()) and float(color.min()) >= 0.0


def test_float64_agrees_with_float32():
    c = rc.make_case("deg0")
    c32 = rr.rasterize(c["means3D"], None, c["shs"], c["opacities"], c["scales"], c["rotations"], c["S"])[0]
    d = {k: c[k].double() for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    c64 = rr.rasterize(d["means3D"], None, d["shs"], d["opacities"], d["scales"], d["rotations"], c["S"])[0]
    # a handful of pixels may sit on an alpha / transmittance threshold; everything else agrees closely
    diff = (c32.double() - c64).abs()
    assert float((diff > 1e-4).float().mean()) < 1e-3 and float(diff.median()) < 1e-6


def test_autograd_matches_central_differences():
    torch.manual_seed(0)
    c = rc.make_case("deg0")
    n = 200
    ins = {k: c[k][40:40 + n].double() for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    ins["opacities"] = ins["opacities"].clamp(max=0.9)         # keep away from the straight-through clamp
    S = c["S"]
    w = torch.rand(3, S.image_height, S.image_width, dtype=torch.float64)

    def f(v):
        return (rr.rasterize(v["means3D"], None, v["shs"], v["opacities"], v["scales"], v["rotations"], S)[0] * w).sum()

    leaf = {k: t.clone().requires_grad_(True) for k, t in ins.items()}
    f(leaf).backward()
    eps = 1e-6
    for k in ("means3D", "shs", "opacities", "scales"):
        dirn = torch.randn_like(ins[k])
        dirn = dirn / dirn.norm()
        plus = dict(ins); plus[k] = ins[k] + eps * dirn
        minus = dict(ins); minus[k] = ins[k] - eps * dirn
        fd = float((f(plus) - f(minus)) / (2 * eps))
        an = float((leaf[k].grad * dirn).sum())
        assert abs(fd - an) <= 2e-4 * max(1.0, abs(an)), (k, fd, an)


def test_means2D_gradient_convention():
    """means2D carries dL/d(pixel position) scaled by 0.5*W, 0.5*H; z stays 0."""
    c = rc.make_case("deg0")
    m3 = c["means3D"].clone().requires_grad_(True)
    m2 = torch.zeros(m3.shape[0], 3, requires_grad=True)
    pre = rr.preprocess(m3, c["scales"], c["rotations"], c["opacities"], c["shs"], c["S"], m2)
    (pre["px"].sum() + 2.0 * pre["py"].sum()).backward()
    assert torch.allclose(m2.grad[:, 0], torch.full_like(m2.grad[:, 0], 0.5 * c["S"].image_width))
    assert torch.allclose(m2.grad[:, 1], torch.full_like(m2.grad[:, 1], 1.0 * c["S"].image_height))
    assert float(m2.grad[:, 2].abs().max()) == 0.0


@pytest.mark.timeout(600)
def test_config0_plumbing_50k_256():
    """BASELINE config 0: 50k Gaussians, 256x256, one time stamp, deformation + rasterizer fwd/bwd on CPU."""
    import deform_cases as dc
    from fdgs import scene_synth as ss
    N, W, H = 50000, 256, 256
    g = ss.GaussianParams(N, scale_mult=2.0)
    cam = ss.make_cameras(3, W, H)[1]
    sd = dc.make_state("dynerf")
    for k in dc.trainable_keys(sd):
        sd[k].requires_grad_(True)
    t = torch.full((N, 1), 0.37)
    pts, sc, ro, op, sh = deform_ref.deform_forward(sd, g._xyz, g._scaling, g._rotation, g._opacity, g.get_features, t)
    S = rr.Settings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.ones(3), 1.0,
                    cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center)
    m2 = torch.zeros(N, 3, requires_grad=True)
    color, radii, depth = rr.rasterize(pts, m2, sh, torch.sigmoid(op), torch.exp(sc),
                                       torch.nn.functional.normalize(ro), S)
    loss = torch.abs(color - 0.5).mean()
    loss.backward()
    assert int((radii > 0).sum()) > 0.9 * N
    for p in (g._xyz, g._scaling, g._rotation, g._opacity, g._features_dc, g._features_rest, m2):
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0
    gk = "deformation_net.grid.grids.1.0"
    assert torch.isfinite(sd[gk].grad).all() and float(sd[gk].grad.abs().max()) > 0
