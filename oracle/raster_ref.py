"""CPU (pure PyTorch) restatement of the tile rasterizer the reference calls.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this
module; it is the checker used by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.

PARITY UNPINNED.  The arithmetic of this path lives in the third-party CUDA
package `diff_gaussian_rasterization` (depth-returning fork, version unknown),
which the reference imports (src/gaussian_renderer/__init__.py:14) but does
not vendor (.gitignore:11) and for which it holds no tests or golden vectors.
This file restates the published 3D Gaussian Splatting tile-rasterizer
algorithm (Kerbl et al., SIGGRAPH 2023) stage by stage and is anchored on
what the reference itself fixes:
  * the call-site contract        src/gaussian_renderer/__init__.py:38-51,57,104-113
  * camera matrix conventions     src/scene/camera.py:42-53, src/utils/graphics_utils.py:59-92
  * SH basis and signs            src/utils/sh_utils.py:26-112 (+0.5 / clamp mirror at
                                  src/gaussian_renderer/__init__.py:95-96)
  * rotation / covariance         src/utils/general_utils.py:101-133 (Sigma = R S S^T R^T)
The last three - the parts whose arithmetic is in the reference's own files - are pinned by
tests/golden/raster_parts.npz (outputs of the reference's Camera, eval_sh and GaussianModel.get_covariance,
made by tests/golden/make_raster_parts_golden.py; checked in tests/test_oracle_raster_parts.py).  The stages
after them are not.  It is therefore the normative oracle for this build: integer stages (radii,
tile rectangles, keys, sort order, tile ranges) are compared bit for bit, the
image within 1e-5 max-abs, gradients within 1e-4 relative.

Every float expression below is written as an explicit sequence of single
IEEE operations (no matmul, no fused ops) in the order the device kernels use,
so that float32 results are reproducible bit for bit on any IEEE machine that
does not contract multiplies and adds.

Conventions fixed here (none is constrained by any file of the reference):
  * means2D carries no value; its gradient is dL/d(pixel position) scaled by
    0.5*W (x) and 0.5*H (y) (the derivative w.r.t. NDC), z component 0.
  * depth output = sum_i depth_i * alpha_i * T_i, shape [1,H,W]; it carries no
    gradient.
  * alpha = min(0.99, opacity*G) is straight-through in the backward pass (the
    clamp does not zero the gradient), as in the published implementation.
  * the +-1.3*tanfov clamp of the view-space position used for the projection
    Jacobian is differentiated exactly (zero w.r.t. the clamped component,
    limit*d/dz w.r.t. depth).
"""
from typing import NamedTuple, Optional

import torch

TILE = 16
NEAR_Z = 0.2
COV_DILATION = 0.3
ALPHA_MAX = 0.99
ALPHA_MIN = 1.0 / 255.0
T_STOP = 1e-4

# src/utils/sh_utils.py:26-43
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


class Settings(NamedTuple):
    """Field list fixed by src/gaussian_renderer/__init__.py:38-51."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False


def tile_grid(S):
    return (S.image_width + TILE - 1) // TILE, (S.image_height + TILE - 1) // TILE


def eval_sh_color(deg, sh, dx, dy, dz):
    """src/utils/sh_utils.py:57-100 for sh laid out [N,16,3]; x,y,z are the
    components of the unit view direction, each [N,1]."""
    x, y, z = dx, dy, dz
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - (SH_C1 * y) * sh[:, 1] + (SH_C1 * z) * sh[:, 2] - (SH_C1 * x) * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + (SH_C2[0] * xy) * sh[:, 4] + (SH_C2[1] * yz) * sh[:, 5]
                   + (SH_C2[2] * (2.0 * zz - xx - yy)) * sh[:, 6]
                   + (SH_C2[3] * xz) * sh[:, 7] + (SH_C2[4] * (xx - yy)) * sh[:, 8])
            if deg > 2:
                res = (res + ((SH_C3[0] * y) * (3.0 * xx - yy)) * sh[:, 9]
                       + ((SH_C3[1] * xy) * z) * sh[:, 10]
                       + ((SH_C3[2] * y) * (4.0 * zz - xx - yy)) * sh[:, 11]
                       + ((SH_C3[3] * z) * (2.0 * zz - 3.0 * xx - 3.0 * yy)) * sh[:, 12]
                       + ((SH_C3[4] * x) * (4.0 * zz - xx - yy)) * sh[:, 13]
                       + ((SH_C3[5] * z) * (xx - yy)) * sh[:, 14]
                       + ((SH_C3[6] * x) * (xx - 3.0 * yy)) * sh[:, 15])
    return res


def covariance3d(scales, rotations, scale_modifier=1.0):
    """3-D covariance Sigma = (R S)(R S)^T as its six upper-triangle entries (xx, xy, xz, yy, yz, zz): what
    GaussianModel.get_covariance builds (src/model/gaussian_model.py:43-50 with build_scaling_rotation and
    strip_symmetric, src/utils/general_utils.py:86-133) for activated scales and a unit quaternion (r, x, y, z)."""
    dt = scales.dtype
    mod = torch.tensor(scale_modifier, dtype=dt)
    sx, sy, sz = mod * scales[:, 0], mod * scales[:, 1], mod * scales[:, 2]
    qr, qx, qy, qz = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    R00 = 1.0 - 2.0 * (qy * qy + qz * qz)
    R01 = 2.0 * (qx * qy - qr * qz)
    R02 = 2.0 * (qx * qz + qr * qy)
    R10 = 2.0 * (qx * qy + qr * qz)
    R11 = 1.0 - 2.0 * (qx * qx + qz * qz)
    R12 = 2.0 * (qy * qz - qr * qx)
    R20 = 2.0 * (qx * qz - qr * qy)
    R21 = 2.0 * (qy * qz + qr * qx)
    R22 = 1.0 - 2.0 * (qx * qx + qy * qy)
    M00, M01, M02 = R00 * sx, R01 * sy, R02 * sz
    M10, M11, M12 = R10 * sx, R11 * sy, R12 * sz
    M20, M21, M22 = R20 * sx, R21 * sy, R22 * sz
    c00 = M00 * M00 + M01 * M01 + M02 * M02
    c01 = M00 * M10 + M01 * M11 + M02 * M12
    c02 = M00 * M20 + M01 * M21 + M02 * M22
    c11 = M10 * M10 + M11 * M11 + M12 * M12
    c12 = M10 * M20 + M11 * M21 + M12 * M22
    c22 = M20 * M20 + M21 * M21 + M22 * M22
    return c00, c01, c02, c11, c12, c22


def preprocess(means3D, scales, rotations, opacities, shs, S, means2D=None, colors_precomp=None, cov3D_precomp=None):
    """Stage 1, per Gaussian.  scales are already exp-activated, rotations
    already normalised, opacities already sigmoid-activated
    (src/gaussian_renderer/__init__.py:85-87).  Matrices are the row-vector
    forms of src/scene/camera.py:48-52: p_view = [x y z 1] @ viewmatrix."""
    dt = means3D.dtype
    N = means3D.shape[0]
    W, H = S.image_width, S.image_height
    gx, gy = tile_grid(S)
    V = S.viewmatrix.to(dt).reshape(16)
    P = S.projmatrix.to(dt).reshape(16)
    cam = S.campos.to(dt)
    x, y, z = means3D[:, 0], means3D[:, 1], means3D[:, 2]

    pvx = V[0] * x + V[4] * y + V[8] * z + V[12]
    pvy = V[1] * x + V[5] * y + V[9] * z + V[13]
    pvz = V[2] * x + V[6] * y + V[10] * z + V[14]
    in_front = pvz.detach() > NEAR_Z
    one = torch.ones_like(pvz)

    phx = P[0] * x + P[4] * y + P[8] * z + P[12]
    phy = P[1] * x + P[5] * y + P[9] * z + P[13]
    phw = P[3] * x + P[7] * y + P[11] * z + P[15]
    p_w = 1.0 / (torch.where(in_front, phw, one) + 1e-7)   # culled rows: value unused, keeps autograd finite
    ppx = phx * p_w
    ppy = phy * p_w

    if cov3D_precomp is not None:      # src/gaussian_renderer/__init__.py:70-71 (compute_cov3D_python)
        c00, c01, c02, c11, c12, c22 = (cov3D_precomp[:, k] for k in range(6))
    else:
        c00, c01, c02, c11, c12, c22 = covariance3d(scales, rotations, S.scale_modifier)

    # EWA projection to a 2-D covariance
    tanx = torch.tensor(S.tanfovx, dtype=dt)
    tany = torch.tensor(S.tanfovy, dtype=dt)
    limx = 1.3 * tanx
    limy = 1.3 * tany
    # true IEEE divisions (python_scalar / tensor would be reciprocal-then-multiply)
    fx = torch.tensor(float(W), dtype=dt) / (2.0 * tanx)
    fy = torch.tensor(float(H), dtype=dt) / (2.0 * tany)
    tz = torch.where(in_front, pvz, one)
    tx = torch.minimum(limx, torch.maximum(-limx, pvx / tz)) * tz
    ty = torch.minimum(limy, torch.maximum(-limy, pvy / tz)) * tz
    J00 = fx / tz
    J02 = -(fx * tx) / (tz * tz)
    J11 = fy / tz
    J12 = -(fy * ty) / (tz * tz)
    # Wr[i][j] = V[4*j + i]: world -> view rotation
    T00 = J00 * V[0] + J02 * V[2]
    T01 = J00 * V[4] + J02 * V[6]
    T02 = J00 * V[8] + J02 * V[10]
    T10 = J11 * V[1] + J12 * V[2]
    T11 = J11 * V[5] + J12 * V[6]
    T12 = J11 * V[9] + J12 * V[10]
    U00 = T00 * c00 + T01 * c01 + T02 * c02
    U01 = T00 * c01 + T01 * c11 + T02 * c12
    U02 = T00 * c02 + T01 * c12 + T02 * c22
    U10 = T10 * c00 + T11 * c01 + T12 * c02
    U11 = T10 * c01 + T11 * c11 + T12 * c12
    U12 = T10 * c02 + T11 * c12 + T12 * c22
    a = U00 * T00 + U01 * T01 + U02 * T02 + COV_DILATION
    b = U00 * T10 + U01 * T11 + U02 * T12
    c = U10 * T10 + U11 * T11 + U12 * T12 + COV_DILATION
    det = a * c - b * b
    det_ok = det.detach() != 0
    det_inv = 1.0 / torch.where(det_ok, det, one)
    conA = c * det_inv
    conB = -b * det_inv
    conC = a * det_inv
    with torch.no_grad():
        mid = 0.5 * (a + c)
        disc = torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        lam = torch.maximum(mid + disc, mid - disc)
        rad_f = torch.ceil(3.0 * torch.sqrt(lam))
        rad_f = torch.nan_to_num(rad_f, nan=0.0, posinf=16777216.0).clamp(max=16777216.0)

    px = ((ppx + 1.0) * float(W) - 1.0) * 0.5
    py = ((ppy + 1.0) * float(H) - 1.0) * 0.5

    with torch.no_grad():
        def lo(p, g):
            return torch.nan_to_num((p - rad_f) / float(TILE), nan=0.0).clamp(0.0, float(g)).to(torch.int32)

        def hi(p, g):
            return torch.nan_to_num((p + rad_f + float(TILE - 1)) / float(TILE), nan=0.0).clamp(0.0, float(g)).to(torch.int32)

        rminx, rminy = lo(px, gx), lo(py, gy)
        rmaxx, rmaxy = hi(px, gx), hi(py, gy)
        area = (rmaxx - rminx) * (rmaxy - rminy)
        visible = in_front & det_ok & (area > 0)
        radii = torch.where(visible, rad_f.to(torch.int32), torch.zeros_like(area))
        tiles_touched = torch.where(visible, area, torch.zeros_like(area))

    if colors_precomp is not None:     # src/gaussian_renderer/__init__.py:89-101: used as is
        rgb = colors_precomp
        clamped = torch.zeros_like(rgb, dtype=torch.bool)
    else:
        # colour from SH (view direction from the camera centre to the mean)
        ddx, ddy, ddz = x - cam[0], y - cam[1], z - cam[2]
        nrm = torch.sqrt(ddx * ddx + ddy * ddy + ddz * ddz)
        ux, uy, uz = (ddx / nrm)[:, None], (ddy / nrm)[:, None], (ddz / nrm)[:, None]
        raw = eval_sh_color(S.sh_degree, shs, ux, uy, uz) + 0.5
        clamped = raw.detach() < 0
        rgb = torch.where(clamped, torch.zeros_like(raw), raw)

    if means2D is not None:   # gradient carrier (value is zero)
        px = px + (0.5 * W) * means2D[:, 0]
        py = py + (0.5 * H) * means2D[:, 1]

    return dict(depth=pvz, radii=radii, px=px, py=py, conA=conA, conB=conB, conC=conC,
                opacity=opacities.reshape(N), rgb=rgb, clamped=clamped, visible=visible,
                rect=torch.stack([rminx, rminy, rmaxx, rmaxy], 1), tiles_touched=tiles_touched,
                cov3D=torch.stack([c00, c01, c02, c11, c12, c22], 1))


def build_keys(pre, S):
    """Stages 2+3: inclusive scan of tiles_touched, then one (key, value) per
    (Gaussian, touched tile), Gaussians in index order, tiles row-major.
    key = (tile_id << 32) | bits(depth_f32) as int64."""
    gx, _ = tile_grid(S)
    tt = pre["tiles_touched"].to(torch.int64)
    offsets = torch.cumsum(tt, 0)
    R = int(offsets[-1]) if tt.numel() else 0
    gid = torch.repeat_interleave(torch.arange(tt.numel()), tt)
    start = (offsets - tt)[gid]
    j = torch.arange(R) - start
    rect = pre["rect"].to(torch.int64)
    wdt = (rect[:, 2] - rect[:, 0])[gid]
    tyy = rect[gid, 1] + j // wdt.clamp(min=1)
    txx = rect[gid, 0] + j % wdt.clamp(min=1)
    tile = tyy * gx + txx
    dbits = pre["depth"].detach().to(torch.float32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    keys = (tile << 32) | dbits[gid]
    return keys, gid.to(torch.int32), offsets.to(torch.int32)


def sort_keys(keys, vals):
    """Stage 4: stable ascending sort."""
    sk, order = torch.sort(keys, stable=True)
    return sk, vals[order]


def tile_ranges(sorted_keys, n_tiles):
    """Stage 5: ranges[t] = [first, last+1) of entries whose key's high word is t; (0,0) if none."""
    tile = (sorted_keys >> 32).to(torch.int64)
    ranges = torch.zeros(n_tiles, 2, dtype=torch.int32)
    if tile.numel():
        idx = torch.arange(tile.numel(), dtype=torch.int32)
        first = torch.ones_like(tile, dtype=torch.bool)
        first[1:] = tile[1:] != tile[:-1]
        last = torch.ones_like(tile, dtype=torch.bool)
        last[:-1] = tile[1:] != tile[:-1]
        ranges[tile[first], 0] = idx[first]
        ranges[tile[last], 1] = idx[last] + 1
    return ranges


def blend_tile(pre, ids, S, tx, ty):
    """Stage 6 for one 16x16 tile; ids = sorted Gaussian indices of the tile.
    Returns colour [256,3], depth [256], final_T [256], n_contrib [256] (row-major pixels)."""
    dt = pre["px"].dtype
    yy, xx = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
    pxf = (tx * TILE + xx).reshape(-1, 1).to(dt)
    pyf = (ty * TILE + yy).reshape(-1, 1).to(dt)
    ids = ids.long()
    K = ids.numel()
    bg = S.bg.to(dt)
    if K == 0:
        return bg.expand(TILE * TILE, 3).clone(), torch.zeros(TILE * TILE, dtype=dt), \
            torch.ones(TILE * TILE, dtype=dt), torch.zeros(TILE * TILE, dtype=torch.int32)
    dx = pre["px"][ids][None, :] - pxf
    dy = pre["py"][ids][None, :] - pyf
    A, B, C = pre["conA"][ids][None, :], pre["conB"][ids][None, :], pre["conC"][ids][None, :]
    power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
    G = torch.exp(torch.clamp(power, max=0.0))
    araw = pre["opacity"][ids][None, :] * G
    alpha = araw - (araw - ALPHA_MAX).clamp(min=0.0).detach()        # straight-through min(0.99, .)
    contrib = (power.detach() <= 0) & (alpha.detach() >= ALPHA_MIN)
    a_eff = torch.where(contrib, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    T_incl = torch.cumprod(one_m, dim=1)
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
    alive = contrib & (T_incl.detach() >= T_STOP)
    w = torch.where(alive, a_eff * T_excl, torch.zeros_like(a_eff))
    color = w @ pre["rgb"][ids]
    depth = (w.detach() @ pre["depth"][ids].detach()[:, None])[:, 0]
    final_T = torch.where(alive, one_m, torch.ones_like(one_m)).prod(dim=1)
    pos = torch.arange(1, K + 1, dtype=torch.int32)[None, :]
    n_contrib = torch.where(alive, pos, torch.zeros_like(pos)).max(dim=1).values
    return color + final_T[:, None] * bg[None, :], depth, final_T, n_contrib


def blend(pre, sorted_vals, ranges, S, tiles=None):
    """Stage 6 over the image (or a subset of tile ids, the rest left at zero)."""
    gx, gy = tile_grid(S)
    dt = pre["px"].dtype
    H, W = S.image_height, S.image_width
    tile_ids = range(gx * gy) if tiles is None else tiles
    col = {}
    for t in tile_ids:
        ty, tx = divmod(int(t), gx)
        r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
        col[int(t)] = blend_tile(pre, sorted_vals[r0:r1], S, tx, ty)
    zc = (torch.zeros(TILE * TILE, 3, dtype=dt), torch.zeros(TILE * TILE, dtype=dt),
          torch.zeros(TILE * TILE, dtype=dt), torch.zeros(TILE * TILE, dtype=torch.int32))
    outs = []
    for k in range(4):
        rows = []
        for ty in range(gy):
            row = [col.get(ty * gx + tx, zc)[k].reshape(TILE, TILE, -1) for tx in range(gx)]
            rows.append(torch.cat(row, dim=1))
        full = torch.cat(rows, dim=0)[:H, :W]
        outs.append(full)
    color = outs[0].permute(2, 0, 1).contiguous()
    depth = outs[1].permute(2, 0, 1).contiguous()
    return color, depth, outs[2][..., 0], outs[3][..., 0]


def rasterize(means3D, means2D, shs, opacities, scales, rotations, S, tiles=None, return_aux=False,
              colors_precomp=None, cov3D_precomp=None):
    """The call of src/gaussian_renderer/__init__.py:104-113: returns (color [3,H,W],
    radii [N] int32, depth [1,H,W]).  Either shs or colors_precomp [N,3]; either
    (scales, rotations) or cov3D_precomp [N,6] (xx,xy,xz,yy,yz,zz, scale modifier applied).
    Differentiable (autograd) w.r.t. every floating-point input given."""
    pre = preprocess(means3D, scales, rotations, opacities, shs, S, means2D, colors_precomp, cov3D_precomp)
    keys, vals, offsets = build_keys(pre, S)
    skeys, svals = sort_keys(keys, vals)
    gx, gy = tile_grid(S)
    ranges = tile_ranges(skeys, gx * gy)
    color, depth, final_T, n_contrib = blend(pre, svals, ranges, S, tiles)
    if return_aux:
        return color, pre["radii"], depth, dict(pre=pre, keys=keys, vals=vals, offsets=offsets,
                                                sorted_keys=skeys, sorted_vals=svals, ranges=ranges,
                                                final_T=final_T, n_contrib=n_contrib)
    return color, pre["radii"], depth


# ---------------------------------------------------------------------------
# Camera construction, restating src/utils/graphics_utils.py:59-92 and
# src/scene/camera.py:42-53 (used by tests to check the shipped scene generator).
# ---------------------------------------------------------------------------
def world2view2(R, t):
    """getWorld2View2 with translate=0, scale=1 (graphics_utils.py:59-70); numpy in, float32 4x4 out."""
    import numpy as np
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection_matrix(znear, zfar, fovX, fovY):
    """getProjectionMatrix (graphics_utils.py:72-92)."""
    import math
    tanY = math.tan(fovY / 2)
    tanX = math.tan(fovX / 2)
    top, right = tanY * znear, tanX * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_tensors(R, t, fovx, fovy, znear=0.01, zfar=100.0):
    """(world_view_transform, full_proj_transform, camera_center) as src/scene/camera.py:48-53."""
    wvt = torch.tensor(world2view2(R, t)).transpose(0, 1)
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    return wvt, full, center
