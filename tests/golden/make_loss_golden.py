"""Generates tests/golden/loss_small.npz by importing the reference's own src/model/losses.py and
src/model/metrics.py (by file path: the package __init__ pulls in modules that need absent packages; these two
files need only torch).  Run in the authoring container only: python tests/golden/make_loss_golden.py"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/src/model"


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    losses, metrics = _load("losses"), _load("metrics")
    g = torch.Generator().manual_seed(6666)
    out = {}
    # a smooth image plus noise (SSIM of pure noise is near zero everywhere), sizes that are no multiple of 32
    for tag, (b, c, h, w) in {"a": (2, 3, 37, 53), "b": (1, 3, 70, 33), "c": (1, 1, 9, 7)}.items():
        yy, xx = torch.meshgrid(torch.linspace(0, 3, h), torch.linspace(0, 2, w), indexing="ij")
        base = (0.5 + 0.4 * torch.sin(3 * xx + yy) * torch.cos(2 * yy)).expand(b, c, h, w)
        gt = (base + 0.05 * torch.randn(b, c, h, w, generator=g)).clamp(0, 1)
        img = (base + 0.15 * torch.randn(b, c, h, w, generator=g)).clamp(0, 1).requires_grad_(True)
        for ws in (11, 5):
            val = losses.ssim(img, gt, window_size=ws)
            grad, = torch.autograd.grad(val, img)
            out["%s_ssim%d" % (tag, ws)] = val.detach().numpy()
            out["%s_ssim%d_grad" % (tag, ws)] = grad.numpy()
            out["%s_ssim%d_per_image" % (tag, ws)] = losses.ssim(img, gt, window_size=ws, size_average=False).detach().numpy()
        l1 = losses.l1_loss(img, gt)
        out[tag + "_l1"] = l1.detach().numpy()
        out[tag + "_l1_grad"] = torch.autograd.grad(l1, img)[0].numpy()
        out[tag + "_psnr"] = metrics.psnr(img.detach(), gt).numpy()
        out[tag + "_img"] = img.detach().numpy()
        out[tag + "_gt"] = gt.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
