"""Drop-in for the `diff_gaussian_rasterization` module the reference imports at
src/gaussian_renderer/__init__.py:14 (depth-returning fork: the call at
:104-113 unpacks `rendered_image, radii, depth`).

`GaussianRasterizationSettings` keeps the field list of the call site at
src/gaussian_renderer/__init__.py:38-51; `GaussianRasterizer(raster_settings)`
is an nn.Module called with the eight keyword tensors of :104-113.  All
compute happens in libfdgs.so (hand-written gfx950 kernels) through the C ABI;
PyTorch only owns the buffers and the stream.
"""
import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import lib, ptr, check, stream_ptr


class GaussianRasterizationSettings(NamedTuple):
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
    prefiltered: bool
    debug: bool


LAST_NUM_RENDERED = 0     # tile-list length R of the most recent forward (bench bookkeeping)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


class RasterState:
    """Device buffers produced by the forward stages (kept for backward / inspection)."""
    __slots__ = ("geom", "radii", "tiles_touched", "point_offsets", "num_rendered", "keys_sorted", "vals_sorted",
                 "ranges", "final_T", "n_contrib", "color", "depth")


def rasterize_forward(means3D, shs, opacities, scales, rotations, S, keep_keys=False):
    """Runs stages 1-6 through the C ABI and returns a RasterState."""
    _lib.require_gpu(means3D, shs, opacities, scales, rotations, S.bg, S.viewmatrix, S.projmatrix, S.campos)
    dev = means3D.device
    N = means3D.shape[0]
    H, W = int(S.image_height), int(S.image_width)
    gx, gy = (W + _lib.TILE - 1) // _lib.TILE, (H + _lib.TILE - 1) // _lib.TILE
    st = RasterState()
    st.geom = torch.empty(N, _lib.GEOM_FLOATS, device=dev, dtype=torch.float32)
    st.radii = torch.empty(N, device=dev, dtype=torch.int32)
    st.tiles_touched = torch.empty(N, device=dev, dtype=torch.int32)
    st.point_offsets = torch.empty(N, device=dev, dtype=torch.int32)
    scan_ws = torch.empty(max(1, lib.fdgs_raster_scan_workspace_bytes(N) // 4), device=dev, dtype=torch.int32)
    vm, pm, cp, bg = _f32c(S.viewmatrix), _f32c(S.projmatrix), _f32c(S.campos), _f32c(S.bg)
    sp = stream_ptr()

    a = _lib.PreprocessFwdArgs(N, int(S.sh_degree), W, H, float(S.tanfovx), float(S.tanfovy), float(S.scale_modifier),
                               ptr(vm), ptr(pm), ptr(cp), ptr(means3D), ptr(scales), ptr(rotations), ptr(opacities),
                               ptr(shs), ptr(st.geom), ptr(st.radii), ptr(st.tiles_touched), ptr(st.point_offsets),
                               ptr(scan_ws))
    _lib.call("fdgs_raster_preprocess_fwd", a)

    # the data-dependent list length R: one small device->host read (stream sync)
    R = int(st.point_offsets[-1].item()) if N > 0 else 0
    st.num_rendered = R
    global LAST_NUM_RENDERED
    LAST_NUM_RENDERED = R
    keys_a = torch.empty(max(R, 1), device=dev, dtype=torch.int64)
    vals_a = torch.empty(max(R, 1), device=dev, dtype=torch.int32)
    st.keys_sorted = torch.empty(max(R, 1), device=dev, dtype=torch.int64)
    st.vals_sorted = torch.empty(max(R, 1), device=dev, dtype=torch.int32)
    st.ranges = torch.empty(gx * gy, 2, device=dev, dtype=torch.int32)
    ws_bytes = lib.fdgs_raster_sort_workspace_bytes(R)
    sort_ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    b = _lib.BinSortArgs(N, W, H, R, ptr(st.geom), ptr(st.point_offsets), ptr(keys_a), ptr(vals_a),
                         ptr(st.keys_sorted), ptr(st.vals_sorted), ptr(st.ranges), ptr(sort_ws), ws_bytes)
    _lib.call("fdgs_raster_bin_sort", b)
    if not keep_keys:
        st.keys_sorted = None

    st.color = torch.empty(3, H, W, device=dev, dtype=torch.float32)
    st.depth = torch.empty(1, H, W, device=dev, dtype=torch.float32)
    st.final_T = torch.empty(H, W, device=dev, dtype=torch.float32)
    st.n_contrib = torch.empty(H, W, device=dev, dtype=torch.int32)
    c = _lib.BlendFwdArgs(W, H, ptr(st.ranges), ptr(st.vals_sorted), ptr(st.geom), ptr(bg), ptr(st.color),
                          ptr(st.depth), ptr(st.final_T), ptr(st.n_contrib))
    _lib.call("fdgs_raster_blend_fwd", c)
    return st


def rasterize_backward(st, grad_color, means3D, shs, scales, rotations, S):
    dev = means3D.device
    N = means3D.shape[0]
    H, W = int(S.image_height), int(S.image_width)
    vm, pm, cp, bg = _f32c(S.viewmatrix), _f32c(S.projmatrix), _f32c(S.campos), _f32c(S.bg)
    sp = stream_ptr()
    dgeom = torch.zeros(N, _lib.GEOM_FLOATS, device=dev, dtype=torch.float32)
    gc = grad_color.to(torch.float32).contiguous()
    a = _lib.BlendBwdArgs(W, H, ptr(st.ranges), ptr(st.vals_sorted), ptr(st.geom), ptr(bg), ptr(st.final_T),
                          ptr(st.n_contrib), ptr(gc), ptr(dgeom))
    _lib.call("fdgs_raster_blend_bwd", a)
    d_means3D = torch.empty(N, 3, device=dev, dtype=torch.float32)
    d_means2D = torch.empty(N, 3, device=dev, dtype=torch.float32)
    d_scales = torch.empty(N, 3, device=dev, dtype=torch.float32)
    d_rot = torch.empty(N, 4, device=dev, dtype=torch.float32)
    d_op = torch.empty(N, 1, device=dev, dtype=torch.float32)
    d_shs = torch.empty(N, 16, 3, device=dev, dtype=torch.float32)
    b = _lib.PreprocessBwdArgs(N, int(S.sh_degree), W, H, float(S.tanfovx), float(S.tanfovy),
                               float(S.scale_modifier), ptr(vm), ptr(pm), ptr(cp), ptr(means3D), ptr(scales),
                               ptr(rotations), ptr(shs), ptr(st.geom), ptr(dgeom), ptr(d_means3D), ptr(d_means2D),
                               ptr(d_scales), ptr(d_rot), ptr(d_op), ptr(d_shs))
    _lib.call("fdgs_raster_preprocess_bwd", b)
    return d_means3D, d_means2D, d_shs, d_op, d_scales, d_rot, dgeom


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, opacities, scales, rotations, S):
        m3, sh = _f32c(means3D), _f32c(shs)
        op, sc, ro = _f32c(opacities).reshape(-1), _f32c(scales), _f32c(rotations)
        st = rasterize_forward(m3, sh, op, sc, ro, S)
        ctx.S = S
        ctx.st = st
        ctx.save_for_backward(m3, sh, sc, ro)
        ctx.op_shape = opacities.shape
        ctx.mark_non_differentiable(st.radii, st.depth)
        return st.color, st.radii, st.depth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, _grad_depth):
        m3, sh, sc, ro = ctx.saved_tensors
        d_m3, d_m2, d_sh, d_op, d_sc, d_ro, _ = rasterize_backward(ctx.st, grad_color, m3, sh, sc, ro, ctx.S)
        ctx.st = None
        return d_m3, d_m2, d_sh, d_op.reshape(ctx.op_shape), d_sc, d_ro, None


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if shs is None or colors_precomp is not None:
            raise NotImplementedError("colors_precomp (convert_SHs_python / override_color) is not built yet: "
                                      "pass shs and let the rasterizer evaluate them")
        if cov3D_precomp is not None or scales is None or rotations is None:
            raise NotImplementedError("cov3D_precomp (compute_cov3D_python) is not built yet: pass scales and rotations")
        if shs.shape[1:] != (16, 3):
            raise ValueError("shs must be [N,16,3] (SH degree 3 storage)")
        return _RasterizeGaussians.apply(means3D, means2D, shs, opacities, scales, rotations, self.raster_settings)
