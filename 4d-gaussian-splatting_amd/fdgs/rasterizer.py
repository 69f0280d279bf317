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

# Capacity of the R-sized buffers (keys, values, sort workspace).  R changes from view to view and from step
# to step; sizing the buffers to R exactly makes the caching allocator see a new size almost every call, and
# a size it has no block for costs a device allocation (tens of milliseconds here).  The capacity is therefore
# R rounded up to a quarter-octave bucket and never shrinks within the process.
_LIST_CAPACITY = 0


def _list_capacity(R):
    global _LIST_CAPACITY
    if R > _LIST_CAPACITY:
        step = 1 << max(R.bit_length() - 3, 10)
        _LIST_CAPACITY = (R + step - 1) // step * step
    return _LIST_CAPACITY


def _f32c(t):
    if t.requires_grad:
        t = t.detach()
    return t if (t.dtype is torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


class RasterState:
    """Device buffers produced by the forward stages (kept for backward / inspection)."""
    __slots__ = ("geom", "radii", "tiles_touched", "point_offsets", "num_rendered", "keys_sorted", "vals_sorted",
                 "ranges", "tile_order", "final_T", "n_contrib", "color", "depth", "cam", "reach_masks")


class _PendingRaster:
    """A forward pass between its two halves: the preprocess call is enqueued and the list length requested."""
    __slots__ = ("st", "ready", "r_host", "bufs", "args", "dims", "list_buffers", "call_args", "inputs", "counted",
                 "keep_masks")


def rasterize_begin(means3D, shs, opacities, scales, rotations, S, colors_precomp=None, cov3D_precomp=None,
                    activations=0, keep_masks=False):
    """First half of rasterize_forward: stages 1-2 are enqueued, the list length is on its way to the host, every
    buffer whose size is known is allocated; nothing waits.  rasterize_finish(pending) does the rest.  (A playback
    loop enqueues the next frame's deformation and first half before it finishes the current frame, so the GPU has
    work while the host waits for the length.)
    keep_masks: the blend kernel records which pixel blocks every list entry reaches (RasterState.reach_masks, four
    bytes per entry in the buffer the unsorted keys occupied until the sort); the backward then reads them instead
    of repeating the test.  For passes that a backward follows."""
    _lib.require_gpu(means3D, shs, opacities, scales, rotations, colors_precomp, cov3D_precomp, S.bg, S.viewmatrix,
                     S.projmatrix, S.campos)
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
    st.cam = (vm, pm, cp, bg)

    a = _lib.PreprocessFwdArgs(N, int(S.sh_degree), W, H, float(S.tanfovx), float(S.tanfovy), float(S.scale_modifier),
                               ptr(vm), ptr(pm), ptr(cp), ptr(means3D), ptr(scales), ptr(rotations), ptr(opacities),
                               ptr(shs), ptr(colors_precomp), ptr(cov3D_precomp), ptr(st.geom), ptr(st.radii), ptr(st.tiles_touched), ptr(st.point_offsets),
                               ptr(scan_ws), int(activations))
    _lib.call("fdgs_raster_preprocess_fwd", a)
    p = _PendingRaster()
    p.st = st
    p.inputs = (means3D, shs, opacities, scales, rotations, colors_precomp, cov3D_precomp, scan_ws)   # alive until finish
    # the data-dependent list length R: one 12-byte device->host read into pinned memory, requested here and waited
    # for on an event in the second half (a synchronous copy of a device tensor goes through a staging buffer and a
    # full stream synchronisation)
    p.ready = p.r_host = None
    if N > 0:
        p.r_host = torch.empty(3, dtype=torch.int32, pin_memory=True)
        p.r_host.copy_(scan_ws[:3], non_blocking=True)
        p.ready = torch.cuda.Event()
        p.ready.record()

    # everything whose size is known is allocated while the preprocess kernels run, before the host waits
    st.color = torch.empty(3, H, W, device=dev, dtype=torch.float32)
    st.depth = torch.empty(1, H, W, device=dev, dtype=torch.float32)
    st.final_T = torch.empty(H, W, device=dev, dtype=torch.float32)
    st.n_contrib = torch.empty(H, W, device=dev, dtype=torch.int32)
    st.ranges = torch.empty(gx * gy, 2, device=dev, dtype=torch.int32)
    st.tile_order = torch.empty(gx * gy, device=dev, dtype=torch.int32)     # tiles by decreasing list length
    # ... the R-sized ones too, at the capacity of the previous calls (redone in the rare case R outgrew it)
    def list_buffers(cap):
        ws_bytes = lib.fdgs_raster_sort_workspace_bytes(cap)
        return (cap, torch.empty(cap, device=dev, dtype=torch.int64), torch.empty(cap, device=dev, dtype=torch.int32),
                torch.empty(cap, device=dev, dtype=torch.int64), torch.empty(cap, device=dev, dtype=torch.int32),
                torch.empty(ws_bytes, device=dev, dtype=torch.uint8), ws_bytes)

    def call_args(bufs):
        _, keys_a, vals_a, keys_s, vals_s, sort_ws, ws_bytes = bufs
        b = _lib.BinSortArgs(N, W, H, 0, ptr(st.geom), ptr(st.point_offsets), ptr(keys_a), ptr(vals_a), ptr(keys_s),
                             ptr(vals_s), ptr(st.ranges), ptr(sort_ws), ws_bytes, 0, 0, ptr(st.tile_order))
        c = _lib.BlendFwdArgs(W, H, ptr(st.ranges), ptr(vals_s), ptr(st.geom), ptr(bg), ptr(st.color), ptr(st.depth),
                              ptr(st.final_T), ptr(st.n_contrib), ptr(st.tile_order),
                              ptr(keys_a) if keep_masks else None)
        return b, c
    p.list_buffers, p.call_args = list_buffers, call_args
    p.keep_masks = bool(keep_masks)
    p.bufs = list_buffers(_LIST_CAPACITY) if _LIST_CAPACITY > 0 else None
    # ... and so are the argument blocks of the two calls that follow the wait: the GPU is idle until they are made
    p.args = call_args(p.bufs) if p.bufs is not None else (None, None)
    # the counting part of the tile lists needs no list length: it runs while the length travels to the host
    p.counted = False
    if p.bufs is not None and N > 0:
        p.args[0].phase = _lib.BINSORT_COUNT
        _lib.call("fdgs_raster_bin_sort", p.args[0])
        p.counted = True
    return p


def rasterize_begin_call(S, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                         cov3D_precomp=None, activations=0):
    """rasterize_begin with the keyword arguments of GaussianRasterizer.forward (no gradients): the inputs are
    brought to contiguous float32 as the autograd route does."""
    opt = lambda t: None if t is None else _f32c(t)
    return rasterize_begin(_f32c(means3D), opt(shs), _f32c(opacities).reshape(-1), opt(scales), opt(rotations), S,
                           colors_precomp=opt(colors_precomp), cov3D_precomp=opt(cov3D_precomp),
                           activations=int(activations))


def rasterize_finish(p, keep_keys=False):
    """Second half: waits for the list length, then stages 3-6.  Returns the RasterState."""
    st, bufs = p.st, p.bufs
    b, c = p.args
    if p.ready is not None:
        p.ready.synchronize()
        dmin, dmax, R = (int(v) & 0xffffffff for v in p.r_host.tolist())
    else:
        dmin, dmax, R = 0, 0, 0
    st.num_rendered = R
    global LAST_NUM_RENDERED
    LAST_NUM_RENDERED = R
    counted = p.counted
    if bufs is None or R > bufs[0]:
        bufs = p.list_buffers(_list_capacity(max(R, 1)))
        b, c = p.call_args(bufs)
        counted = False                 # another workspace: the counts are taken again
    st.keys_sorted, st.vals_sorted = bufs[3], bufs[4]
    st.reach_masks = bufs[1] if p.keep_masks else None      # the unsorted keys' buffer, free once the lists are built
    b.num_rendered, b.depth_bits_min, b.depth_bits_max = R, dmin, dmax
    b.phase = _lib.BINSORT_FINISH if counted else _lib.BINSORT_ALL
    if R == 0:          # no list entries: the list pointer is NULL (the library then also accepts the NULL geom of an empty scene)
        c.point_list = None
    _lib.call("fdgs_raster_bin_sort", b)
    if not keep_keys:
        st.keys_sorted = None
    _lib.call("fdgs_raster_blend_fwd", c)
    p.inputs = p.bufs = p.args = p.list_buffers = p.call_args = None
    return st


def rasterize_forward(means3D, shs, opacities, scales, rotations, S, keep_keys=False, colors_precomp=None,
                      cov3D_precomp=None, activations=0, keep_masks=False):
    """Runs stages 1-6 through the C ABI and returns a RasterState.  Either shs or colors_precomp [N,3],
    either (scales, rotations) or cov3D_precomp [N,6].  `activations` (ACT_* flags of fdgs._lib) marks
    scales / rotations / opacities as raw parameters whose activation the preprocess kernel applies."""
    return rasterize_finish(rasterize_begin(means3D, shs, opacities, scales, rotations, S, colors_precomp=colors_precomp,
                                            cov3D_precomp=cov3D_precomp, activations=activations,
                                            keep_masks=keep_masks), keep_keys=keep_keys)


def rasterize_backward(st, grad_color, means3D, shs, scales, rotations, S, colors_precomp=None, cov3D_precomp=None,
                       activations=0):
    dev = means3D.device
    N = means3D.shape[0]
    H, W = int(S.image_height), int(S.image_width)
    vm, pm, cp, bg = st.cam
    dgeom = torch.zeros(N, _lib.GEOM_FLOATS, device=dev, dtype=torch.float32)
    gc = _f32c(grad_color)
    a = _lib.BlendBwdArgs(W, H, ptr(st.ranges), ptr(st.vals_sorted) if st.num_rendered > 0 else None, ptr(st.geom), ptr(bg), ptr(st.final_T),
                          ptr(st.n_contrib), ptr(gc), ptr(dgeom), ptr(st.tile_order),
                          ptr(getattr(st, "reach_masks", None)) if st.num_rendered > 0 else None)
    _lib.call("fdgs_raster_blend_bwd", a)
    d_means3D = torch.empty(N, 3, device=dev, dtype=torch.float32)
    d_means2D = torch.empty(N, 3, device=dev, dtype=torch.float32)
    d_scales = d_rot = d_shs = d_colors = d_cov = None
    if cov3D_precomp is None:
        d_scales = torch.empty(N, 3, device=dev, dtype=torch.float32)
        d_rot = torch.empty(N, 4, device=dev, dtype=torch.float32)
    else:
        d_cov = torch.empty(N, 6, device=dev, dtype=torch.float32)
    d_op = torch.empty(N, 1, device=dev, dtype=torch.float32)
    if colors_precomp is None:
        d_shs = torch.empty(N, 16, 3, device=dev, dtype=torch.float32)
    else:
        d_colors = torch.empty(N, 3, device=dev, dtype=torch.float32)
    b = _lib.PreprocessBwdArgs(N, int(S.sh_degree), W, H, float(S.tanfovx), float(S.tanfovy),
                               float(S.scale_modifier), ptr(vm), ptr(pm), ptr(cp), ptr(means3D), ptr(scales),
                               ptr(rotations), ptr(shs), ptr(colors_precomp), ptr(cov3D_precomp), ptr(st.geom),
                               ptr(dgeom), ptr(d_means3D), ptr(d_means2D), ptr(d_scales), ptr(d_rot), ptr(d_op),
                               ptr(d_shs), ptr(d_colors), ptr(d_cov), int(activations))
    _lib.call("fdgs_raster_preprocess_bwd", b)
    return dict(means3D=d_means3D, means2D=d_means2D, shs=d_shs, colors=d_colors, opacities=d_op, scales=d_scales,
                rotations=d_rot, cov3D=d_cov, geom=dgeom)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, S,
                activations=0):
        opt = lambda t: None if t is None else _f32c(t)
        m3, sh, cp = _f32c(means3D), opt(shs), opt(colors_precomp)
        op, sc, ro, cv = _f32c(opacities).reshape(-1), opt(scales), opt(rotations), opt(cov3D_precomp)
        st = rasterize_forward(m3, sh, op, sc, ro, S, colors_precomp=cp, cov3D_precomp=cv, activations=activations,
                               keep_masks=any(ctx.needs_input_grad))
        ctx.S = S
        ctx.activations = activations
        # the context keeps what the backward reads (geom, list, ranges, final_T, n_contrib) and none of the
        # outputs: an output held by its own grad_fn's context would be a reference cycle through C++ that Python's
        # collector cannot see, leaking the images and lists of every grad-enabled render that is never backpropagated
        color, radii, depth = st.color, st.radii, st.depth
        st.color = st.radii = st.depth = st.tiles_touched = st.point_offsets = None
        ctx.st = st
        ctx.inputs = (m3, sh, cp, sc, ro, cv)
        ctx.op_shape = opacities.shape
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, _grad_depth):
        m3, sh, cp, sc, ro, cv = ctx.inputs
        g = rasterize_backward(ctx.st, grad_color, m3, sh, sc, ro, ctx.S, colors_precomp=cp, cov3D_precomp=cv,
                               activations=ctx.activations)
        ctx.st = None
        ctx.inputs = None
        return (g["means3D"], g["means2D"], g["shs"], g["colors"], g["opacities"].reshape(ctx.op_shape), g["scales"],
                g["rotations"], g["cov3D"], None, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, activations=0):
        """The reference's call (src/gaussian_renderer/__init__.py:104-113).  `activations` is an extension:
        ACT_* flags of fdgs._lib say which of scales / rotations / opacities are still raw parameters; their
        exp / normalize / sigmoid then runs inside the preprocess kernels (forward and backward)."""
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide exactly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if shs is not None and tuple(shs.shape[1:]) != (16, 3):
            raise ValueError("shs must be [N,16,3] (SH degree 3 storage)")
        return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                         cov3D_precomp, self.raster_settings, int(activations))
