"""The reference's image losses (src/model/losses.py:7-8 L1, :13-53 SSIM), its per-iteration PSNR
(src/model/metrics.py:7-28) and densification statistics
(src/runner/fdgs_trainer.py:181-182, src/model/gaussian_model.py:182-184) as single passes through
libfdgs.so instead of chains of element-wise PyTorch launches."""
import torch

from . import _lib
from ._lib import lib, ptr


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, network_output, gt, scale, want_psnr=False):
        _lib.require_gpu(network_output, gt)
        a = network_output.detach()
        a = a if (a.dtype is torch.float32 and a.is_contiguous()) else a.to(torch.float32).contiguous()
        b = gt.detach()
        if b.shape != a.shape:
            b = b.expand_as(a)
        b = b if (b.dtype is torch.float32 and b.is_contiguous()) else b.to(torch.float32).contiguous()
        n = a.numel()
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        partial = torch.empty(lib.fdgs_l1_loss_partials(n) * (2 if want_psnr else 1), device=a.device,
                              dtype=torch.float32)
        psnr = torch.empty((), device=a.device, dtype=torch.float32) if want_psnr else None
        _lib.call("fdgs_l1_loss_fwd", _lib.L1FwdArgs(n, ptr(a), ptr(b), float(scale), ptr(loss), ptr(partial),
                                                     ptr(psnr)))
        ctx.save_for_backward(a, b)
        ctx.scale = float(scale)
        ctx.out_shape = network_output.shape
        if want_psnr:
            ctx.mark_non_differentiable(psnr)
            return loss, psnr
        return loss

    @staticmethod
    def backward(ctx, grad_loss, *unused):
        a, b = ctx.saved_tensors
        g = grad_loss if (grad_loss.dtype is torch.float32 and grad_loss.is_contiguous()) else \
            grad_loss.to(torch.float32).contiguous()
        grad_a = torch.empty_like(a)
        _lib.call("fdgs_l1_loss_bwd", _lib.L1BwdArgs(a.numel(), ptr(a), ptr(b), ctx.scale, ptr(g), ptr(grad_a)))
        return grad_a.view(ctx.out_shape), None, None, None


def l1_loss(network_output, gt, scale=1.0):
    """scale * mean |network_output - gt|; the gradient flows to network_output only (gt is data)."""
    return _L1Loss.apply(network_output, gt, scale)


def l1_loss_psnr(network_output, gt, scale=1.0):
    """(l1_loss, psnr) from one pass over the two images: the reference computes both every iteration
    (src/runner/fdgs_trainer.py:130-131).  psnr = 20 log10(1 / sqrt(mean squared difference)) over all the
    elements given (one image; the reference then averages over the images of the batch)."""
    return _L1Loss.apply(network_output, gt, scale, True)


def psnr(img1, img2):
    """src/model/metrics.py:7-28 without a mask: [B, 1] values, one per image of a [B, C, H, W] stack."""
    if img1.dim() == 3:
        img1, img2 = img1.unsqueeze(0), img2.unsqueeze(0)
    with torch.no_grad():
        return torch.stack([_L1Loss.apply(x, y, 1.0, True)[1] for x, y in zip(img1, img2)]).reshape(-1, 1)


# ---- SSIM (src/model/losses.py:13-53) --------------------------------------------------------------------

def _gaussian_window(window_size, sigma=1.5):
    """src/model/losses.py:13-15, same operations in the same precision."""
    from math import exp
    gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


def _ssim_window_args(a, window_size):
    if window_size % 2 != 1 or window_size // 2 > _lib.SSIM_MAX_RADIUS:
        raise ValueError("window_size must be odd and at most %d" % (2 * _lib.SSIM_MAX_RADIUS + 1))
    a.radius = window_size // 2
    for i, w in enumerate(_gaussian_window(window_size).tolist()):
        a.window[i] = w


class _SsimPlaneSums(torch.autograd.Function):
    """[planes] sums of the SSIM map of each image plane; differentiable in img1."""

    @staticmethod
    def forward(ctx, img1, img2, window_size):
        _lib.require_gpu(img1, img2)
        x = img1.detach()
        x = x if (x.dtype is torch.float32 and x.is_contiguous()) else x.to(torch.float32).contiguous()
        y = img2.detach()
        y = y if (y.dtype is torch.float32 and y.is_contiguous()) else y.to(torch.float32).contiguous()
        h, w = x.shape[-2:]
        planes = x.numel() // (h * w)
        need_grad = img1.requires_grad
        a = _lib.SsimFwdArgs()
        a.planes, a.height, a.width = planes, h, w
        a.img1, a.img2 = x.data_ptr(), y.data_ptr()
        _ssim_window_args(a, window_size)
        a.c1, a.c2 = 0.01 ** 2, 0.03 ** 2
        sums = torch.empty(planes, device=x.device, dtype=torch.float32)
        partial = torch.empty(planes * lib.fdgs_ssim_tiles(h, w), device=x.device, dtype=torch.float32)
        dmaps = torch.empty((3,) + tuple(x.shape), device=x.device, dtype=torch.float32) if need_grad else None
        a.plane_sum, a.partial, a.dmaps = sums.data_ptr(), partial.data_ptr(), ptr(dmaps)
        _lib.call("fdgs_ssim_fwd", a)
        if need_grad:
            ctx.save_for_backward(x, y, dmaps)
        ctx.window_size = window_size
        ctx.in_shape = img1.shape
        return sums

    @staticmethod
    def backward(ctx, grad_sums):
        x, y, dmaps = ctx.saved_tensors
        h, w = x.shape[-2:]
        a = _lib.SsimBwdArgs()
        a.planes, a.height, a.width = x.numel() // (h * w), h, w
        a.img1, a.img2, a.dmaps = x.data_ptr(), y.data_ptr(), dmaps.data_ptr()
        _ssim_window_args(a, ctx.window_size)
        g = grad_sums if (grad_sums.dtype is torch.float32 and grad_sums.is_contiguous()) else \
            grad_sums.to(torch.float32).contiguous()
        grad = torch.empty_like(x)
        a.plane_scale, a.grad_img1 = g.data_ptr(), grad.data_ptr()
        _lib.call("fdgs_ssim_bwd", a)
        return grad.view(ctx.in_shape), None, None


def ssim(img1, img2, window_size=11, size_average=True):
    """src/model/losses.py:23-53, same signature: [B, C, H, W] (or [C, H, W]) stacks; the mean of the SSIM map
    over everything, or per image with size_average=False.  Gradient with respect to img1."""
    if img1.shape != img2.shape or img1.dim() < 3:
        raise ValueError("ssim: two stacks of equal shape [..., C, H, W]")
    h, w = img1.shape[-2:]
    sums = _SsimPlaneSums.apply(img1, img2, window_size)
    if size_average:
        return sums.sum() / float(sums.numel() * h * w)
    if img1.dim() != 4:
        raise ValueError("size_average=False needs [B, C, H, W]")
    return sums.view(img1.shape[0], -1).sum(1) / float(img1.shape[1] * h * w)


@torch.no_grad()
def add_densification_stats(radii, viewspace_grad, xyz_gradient_accum, denom, max_radii2D=None):
    """In place, for the Gaussians with radii > 0: max_radii2D = max(max_radii2D, radii),
    xyz_gradient_accum += |viewspace_grad[:, :2]|, denom += 1."""
    _lib.require_gpu(radii, viewspace_grad, xyz_gradient_accum, denom, max_radii2D)
    n = radii.shape[0]
    r = radii if (radii.dtype is torch.int32 and radii.is_contiguous()) else radii.to(torch.int32).contiguous()
    g = viewspace_grad if (viewspace_grad.dtype is torch.float32 and viewspace_grad.is_contiguous()) else \
        viewspace_grad.to(torch.float32).contiguous()
    if g.shape != (n, 3) or xyz_gradient_accum.numel() != n or denom.numel() != n:
        raise ValueError("add_densification_stats: radii [N], viewspace_grad [N,3], accumulators with N entries")
    for t in (xyz_gradient_accum, denom, max_radii2D):
        if t is not None and not (t.dtype is torch.float32 and t.is_contiguous()):
            raise ValueError("the accumulators must be contiguous float32 (they are updated in place)")
    _lib.call("fdgs_densify_stats", _lib.DensifyStatsArgs(n, ptr(r), ptr(g), ptr(max_radii2D), ptr(xyz_gradient_accum),
                                                           ptr(denom)))


# ---- grid regulariser (src/model/deformation_wrapper.py:52-92) ---------------------------------------------

_TIME_PLANES = (2, 4, 5)       # planes with a time axis, in itertools.combinations(range(4), 2) order


def _reg_args(planes_per_level, time_smoothness_weight, l1_time_planes_weight, plane_tv_weight):
    planes = []
    a = _lib.PlaneRegArgs()
    k = 0
    for grids in planes_per_level:
        if len(grids) == 3:                     # a static (3-D) field has no regularised planes (:58-59, 69-70)
            continue
        for ci, p in enumerate(grids):
            if k >= _lib.REG_MAX_PLANES:
                raise ValueError("at most 12 planes per call")
            t = p.detach()
            if t.dim() != 4 or t.shape[0] != 1 or t.dtype is not torch.float32 or not t.is_contiguous():
                raise ValueError("planes must be contiguous float32 [1, C, h, w]")
            if k and t.shape[1] != a.channels:
                raise ValueError("all planes must have the same channel count")
            a.channels = t.shape[1]
            a.plane[k] = t.data_ptr(); a.h[k] = t.shape[2]; a.w[k] = t.shape[3]
            is_time = ci in _TIME_PLANES
            a.w_smooth[k] = time_smoothness_weight if is_time else plane_tv_weight
            a.w_l1[k] = l1_time_planes_weight if is_time else 0.0
            planes.append(p)
            k += 1
    a.num_planes = k
    return a, planes


class _PlaneRegulation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, levels, *flat_planes):
        it = iter(flat_planes)
        per_level = [[next(it) for _ in range(n)] for n in levels]
        _lib.require_gpu(*flat_planes)
        a, planes = _reg_args(per_level, *weights)
        dev = flat_planes[0].device
        loss = torch.zeros((), device=dev, dtype=torch.float32)
        if a.num_planes:
            partial = torch.empty(lib.fdgs_plane_reg_partials(a.num_planes), device=dev, dtype=torch.float32)
            a.loss, a.loss_partial = loss.data_ptr(), partial.data_ptr()
            _lib.call("fdgs_plane_regulation", a)
        ctx.weights, ctx.levels = weights, levels
        ctx.save_for_backward(*flat_planes)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        flat_planes = ctx.saved_tensors
        it = iter(flat_planes)
        per_level = [[next(it) for _ in range(n)] for n in ctx.levels]
        a, planes = _reg_args(per_level, *ctx.weights)
        grads = {}
        if a.num_planes:
            g = grad_loss if (grad_loss.dtype is torch.float32 and grad_loss.is_contiguous()) else \
                grad_loss.to(torch.float32).contiguous()
            flat = torch.empty(sum(p.numel() for p in planes), device=g.device, dtype=torch.float32)
            for k, (p, piece) in enumerate(zip(planes, flat.split_with_sizes([p.numel() for p in planes]))):
                grads[id(p)] = piece.view(p.shape)
                a.grad[k] = piece.data_ptr()
            a.grad_scale = g.data_ptr()
            _lib.call("fdgs_plane_regulation", a)
        return (None, None) + tuple(grads.get(id(p)) for p in flat_planes)


def compute_regulation(deformation_fields, time_smoothness_weight, l1_time_planes_weight, plane_tv_weight):
    """DeformationWrapper.compute_regulation for a DeformationFields module (same argument order): a scalar
    with autograd, two launches forward and one backward for all planes."""
    grids = deformation_fields.deformation_net.grid.grids
    flat = [p for gp in grids for p in gp]
    return _PlaneRegulation.apply((float(time_smoothness_weight), float(l1_time_planes_weight), float(plane_tv_weight)),
                                  tuple(len(gp) for gp in grids), *flat)


@torch.no_grad()
def add_regulation_grad_(deformation_fields, time_smoothness_weight, l1_time_planes_weight, plane_tv_weight,
                         want_loss=True):
    """What `loss += compute_regulation(...)` before `loss.backward()` does to the planes' .grad, done after the
    backward in one launch: the regulariser's gradient is added in place to the existing .grad tensors (no
    twelve separate accumulation launches).  Returns the regulariser's value (or None)."""
    grids = deformation_fields.deformation_net.grid.grids
    a, planes = _reg_args([list(gp) for gp in grids], float(time_smoothness_weight), float(l1_time_planes_weight),
                          float(plane_tv_weight))
    if not a.num_planes:
        return None
    _lib.require_gpu(*planes)
    for k, p in enumerate(planes):
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        if not (p.grad.is_contiguous() and p.grad.dtype is torch.float32):
            raise ValueError("plane gradients must be contiguous float32")
        a.grad[k] = p.grad.data_ptr()
    a.accumulate = 1
    loss = None
    if want_loss:
        dev = planes[0].device
        loss = torch.empty((), device=dev, dtype=torch.float32)
        partial = torch.empty(lib.fdgs_plane_reg_partials(a.num_planes), device=dev, dtype=torch.float32)
        a.loss, a.loss_partial = loss.data_ptr(), partial.data_ptr()
    _lib.call("fdgs_plane_regulation", a)
    return loss
