"""`from simple_knn._C import distCUDA2` (src/model/gaussian_model.py:5,57): mean squared distance of every point
to its three nearest neighbours, computed by libfdgs.so on the GPU."""
import torch

from fdgs import _lib


def distCUDA2(points):
    _lib.require_gpu(points)
    p = points.detach()
    p = p if (p.dtype is torch.float32 and p.is_contiguous()) else p.to(torch.float32).contiguous()
    if p.dim() != 2 or p.shape[1] != 3:
        raise ValueError("distCUDA2 expects points [N,3]")
    out = torch.empty(p.shape[0], device=p.device, dtype=torch.float32)
    _lib.check(_lib.lib.fdgs_knn_mean_dist2(_lib.ptr(p), p.shape[0], _lib.ptr(out), _lib.stream_ptr()),
               "fdgs_knn_mean_dist2")
    return out
