"""The reference's image loss (src/model/losses.py:7-8) and densification statistics
(src/runner/fdgs_trainer.py:181-182, src/model/gaussian_model.py:182-184) as single passes through
libfdgs.so instead of chains of element-wise PyTorch launches."""
import torch

from . import _lib
from ._lib import lib, ptr


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, network_output, gt, scale):
        _lib.require_gpu(network_output, gt)
        a = network_output.detach()
        a = a if (a.dtype is torch.float32 and a.is_contiguous()) else a.to(torch.float32).contiguous()
        b = gt.detach()
        if b.shape != a.shape:
            b = b.expand_as(a)
        b = b if (b.dtype is torch.float32 and b.is_contiguous()) else b.to(torch.float32).contiguous()
        n = a.numel()
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        partial = torch.empty(lib.fdgs_l1_loss_partials(n), device=a.device, dtype=torch.float32)
        _lib.call("fdgs_l1_loss_fwd", _lib.L1FwdArgs(n, ptr(a), ptr(b), float(scale), ptr(loss), ptr(partial)))
        ctx.save_for_backward(a, b)
        ctx.scale = float(scale)
        ctx.out_shape = network_output.shape
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        a, b = ctx.saved_tensors
        g = grad_loss if (grad_loss.dtype is torch.float32 and grad_loss.is_contiguous()) else \
            grad_loss.to(torch.float32).contiguous()
        grad_a = torch.empty_like(a)
        _lib.call("fdgs_l1_loss_bwd", _lib.L1BwdArgs(a.numel(), ptr(a), ptr(b), ctx.scale, ptr(g), ptr(grad_a)))
        return grad_a.view(ctx.out_shape), None, None


def l1_loss(network_output, gt, scale=1.0):
    """scale * mean |network_output - gt|; the gradient flows to network_output only (gt is data)."""
    return _L1Loss.apply(network_output, gt, scale)


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
