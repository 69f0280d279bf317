"""GPU parity of the image losses: fdgs.losses.ssim / psnr / l1_loss_psnr (libfdgs.so) against the golden
vectors from the reference's own losses.py / metrics.py and, at other sizes, against oracle/loss_ref.py.
SSIM value <= 1e-5 absolute, its gradient <= 1e-4 relative (max-norm); PSNR <= 1e-4 dB."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref as lr

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_small.npz"))
VAL_TOL, GRAD_TOL, PSNR_TOL = 1e-5, 1e-4, 1e-4


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("ws", [11, 5])
def test_ssim_against_reference_vectors(tag, ws):
    from fdgs import losses
    img = torch.from_numpy(GOLD[tag + "_img"]).cuda().requires_grad_(True)
    gt = torch.from_numpy(GOLD[tag + "_gt"]).cuda()
    val = losses.ssim(img, gt, window_size=ws)
    grad, = torch.autograd.grad(val, img)
    assert abs(float(val.detach()) - float(GOLD["%s_ssim%d" % (tag, ws)])) <= VAL_TOL
    want = GOLD["%s_ssim%d_grad" % (tag, ws)]
    assert np.abs(grad.cpu().numpy() - want).max() <= GRAD_TOL * np.abs(want).max()
    per = losses.ssim(img, gt, window_size=ws, size_average=False)
    assert np.abs(per.detach().cpu().numpy() - GOLD["%s_ssim%d_per_image" % (tag, ws)]).max() <= VAL_TOL
    # gradient through the per-image form with unequal weights
    if img.shape[0] > 1:
        wts = torch.tensor([0.25, 2.0], device="cuda")
        g2, = torch.autograd.grad((per * wts).sum(), img)
        ic = torch.from_numpy(GOLD[tag + "_img"]).requires_grad_(True)
        want2, = torch.autograd.grad((lr.ssim(ic, gt.cpu(), window_size=ws, size_average=False) * wts.cpu()).sum(), ic)
        assert (g2.cpu() - want2).abs().max() <= GRAD_TOL * want2.abs().max()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_l1_and_psnr_against_reference_vectors(tag):
    from fdgs import losses
    img = torch.from_numpy(GOLD[tag + "_img"]).cuda().requires_grad_(True)
    gt = torch.from_numpy(GOLD[tag + "_gt"]).cuda()
    got = losses.psnr(img.detach(), gt).cpu().numpy()
    assert got.shape == GOLD[tag + "_psnr"].shape
    assert np.abs(got - GOLD[tag + "_psnr"]).max() <= PSNR_TOL
    if img.shape[0] == 1:
        loss, p = losses.l1_loss_psnr(img, gt)
        assert abs(float(loss) - float(GOLD[tag + "_l1"])) <= 1e-6 * float(GOLD[tag + "_l1"])
        assert abs(float(p) - float(GOLD[tag + "_psnr"][0, 0])) <= PSNR_TOL
        loss.backward()
        assert np.abs(img.grad.cpu().numpy() - GOLD[tag + "_l1_grad"]).max() <= 1e-6 * np.abs(GOLD[tag + "_l1_grad"]).max()


@pytest.mark.parametrize("shape,ws", [((1, 3, 260, 300), 11), ((3, 64, 64), 11), ((2, 2, 31, 95), 15), ((1, 1, 1, 1), 3),
                                      ((1, 3, 32, 32), 1)])
def test_ssim_against_oracle(shape, ws):
    from fdgs import losses
    g = torch.Generator().manual_seed(sum(shape) + ws)
    base = torch.rand(shape, generator=g)
    img = (base + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1)
    gt = base
    ic = img.clone().requires_grad_(True)
    want = lr.ssim(ic if ic.dim() == 4 else ic.unsqueeze(0), gt if gt.dim() == 4 else gt.unsqueeze(0), window_size=ws)
    wgrad, = torch.autograd.grad(want, ic)
    ig = img.cuda().requires_grad_(True)
    val = losses.ssim(ig, gt.cuda(), window_size=ws)
    grad, = torch.autograd.grad(val, ig)
    assert abs(float(val) - float(want)) <= VAL_TOL
    assert (grad.cpu() - wgrad).abs().max() <= GRAD_TOL * wgrad.abs().max()
    # no gradient requested: no derivative maps are written, same value
    assert abs(float(losses.ssim(img.cuda(), gt.cuda(), window_size=ws)) - float(want)) <= VAL_TOL


def test_ssim_identical_images_and_full_size():
    from fdgs import losses
    g = torch.Generator().manual_seed(5)
    img = torch.rand(1, 3, 1014, 1352, generator=g).cuda()
    assert abs(float(losses.ssim(img, img.clone())) - 1.0) <= 1e-5
    x = img.clone().requires_grad_(True)
    gt = (img + 0.05 * torch.randn(img.shape, generator=g).cuda()).clamp(0, 1)
    val = losses.ssim(x, gt)
    grad, = torch.autograd.grad(val, x)
    assert 0.0 < float(val) < 1.0 and torch.isfinite(grad).all()
    # directional derivative (along the gradient's signs, so the change is far above the rounding of a mean of
    # four million values) against a central difference of the kernel's own forward
    d = torch.sign(grad)
    eps = 1e-3
    num = (float(losses.ssim(img + eps * d, gt)) - float(losses.ssim(img - eps * d, gt))) / (2 * eps)
    ana = float((grad.double() * d.double()).sum())
    assert abs(num - ana) <= 2e-2 * abs(ana) + 1e-6
    for _ in range(3):
        losses.ssim(x, gt).backward()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        x.grad = None
        losses.ssim(x, gt).backward()
    e.record()
    torch.cuda.synchronize()
    print("ssim fwd+bwd at 3 x 1014 x 1352: %.3f ms" % (s.elapsed_time(e) / 10))
    ws = lr.create_window(11, 3).cuda()
    def torch_ssim(a, b):
        import torch.nn.functional as F
        mu1 = F.conv2d(a, ws, padding=5, groups=3); mu2 = F.conv2d(b, ws, padding=5, groups=3)
        s1 = F.conv2d(a * a, ws, padding=5, groups=3) - mu1 * mu1; s2 = F.conv2d(b * b, ws, padding=5, groups=3) - mu2 * mu2
        s12 = F.conv2d(a * b, ws, padding=5, groups=3) - mu1 * mu2
        return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    for _ in range(2):
        x.grad = None
        torch_ssim(x, gt).backward()
    torch.cuda.synchronize()
    s.record()
    for _ in range(5):
        x.grad = None
        torch_ssim(x, gt).backward()
    e.record()
    torch.cuda.synchronize()
    print("PyTorch conv2d ssim fwd+bwd:      %.3f ms" % (s.elapsed_time(e) / 5))


def test_train_step_with_ssim_term_and_psnr():
    """One training step over two views with lambda_dssim = 0.2 (fdgs_trainer.py:130-138): the image gradient
    that reaches the renderer and the PSNR must be those of the reference's composition, evaluated with
    oracle/loss_ref.py on the same rendered images."""
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import DeformHolder, TrainStep
    dev = torch.device("cuda:0")
    n, w, h = 6000, 160, 120
    g = ss.GaussianParams(n, device=dev, scale_mult=3.0)
    torch.manual_seed(ss.SEED)
    net = ss.perturb_deformation_(DeformationFields(ss.DYNERF_NET_KWARGS)).to(dev)
    net.deformation_net.set_aabb(list(ss.SCENE_MAX), list(ss.SCENE_MIN))
    deforms = DeformHolder(net)
    cams = ss.make_cameras(4, w, h, device=dev)
    bg = torch.ones(3, device=dev)
    g2 = ss.GaussianParams(n, seed=ss.SEED + 100, device=dev, requires_grad=False, scale_mult=3.0)
    with torch.no_grad():
        for cam in cams:
            cam.original_image = render(cam, g2, deforms, bg, stage="coarse")["render"].clamp(0, 1).clone()
    views = [cams[0], cams[2]]
    lam = 0.2
    # the reference's loss on the same renders, with autograd through the images only
    with torch.no_grad():
        imgs = torch.stack([render(c, g, deforms, bg, stage="fine")["render"] for c in views]).cpu()
    gts = torch.stack([c.original_image for c in views]).cpu()
    ic = imgs.clone().requires_grad_(True)
    ref_loss = lr.l1_loss(ic, gts) + lam * (1.0 - lr.ssim(ic, gts))
    want_dimg, = torch.autograd.grad(ref_loss, ic)
    want_psnr = lr.psnr(imgs, gts).mean()
    step = TrainStep(g, deforms, bg, lambda_dssim=lam, regulation=None)
    step.optimizer.step = lambda: None
    seen = []
    import fdgs.renderer as fr
    orig = fr.render

    def spy(*a, **k):
        pkg = orig(*a, **k)
        pkg["render"].register_hook(lambda gr: seen.append(gr.detach().cpu()))
        return pkg
    import fdgs.train_step as ts
    ts.render = spy
    try:
        loss = step(views)
    finally:
        ts.render = orig
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-5
    assert abs(float(step.last_psnr) - float(want_psnr)) <= PSNR_TOL
    got = torch.stack(seen[::-1]) if len(seen) == 2 else None
    assert got is not None
    # hooks fire in backward order: match each to its view by value.  Tolerance 5e-4 of the largest entry: on the
    # saturated white background sigma = E[x^2] - mu^2 cancels to rounding level (1e-7) and is divided by
    # C2 = 9e-4, so any two float32 evaluation orders of the reference's formula differ by about 1e-4 there
    for v in range(2):
        errs = [float((gk - want_dimg[v]).abs().max()) for gk in seen]
        assert min(errs) <= 5e-4 * float(want_dimg[v].abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3 * 1014 * 1352, 4096 * 3 + 17, 5])
def test_l1_forward_writes_the_gradient_of_the_backward_call(n):
    """fdgs_l1_loss_fwd with grad_a set: the same values as fdgs_l1_loss_bwd with an upstream gradient of 1 (bit
    for bit), loss and PSNR unchanged; full image size, a ragged tail and fewer elements than one vector."""
    from fdgs import _lib
    from fdgs._lib import lib, ptr
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(n)
    a = torch.rand(n, generator=gen).to(dev)
    b = torch.rand(n, generator=gen).to(dev)
    b[::7] = a[::7]                                   # exact zeros of the difference
    scale = 0.125
    partial = torch.empty(2 * lib.fdgs_l1_loss_partials(n), device=dev)
    out = {}
    for tag, with_grad in (("plain", False), ("fused", True)):
        loss, ps = torch.empty((), device=dev), torch.empty((), device=dev)
        grad = torch.full((n,), float("nan"), device=dev)
        _lib.call("fdgs_l1_loss_fwd", _lib.L1FwdArgs(n, ptr(a), ptr(b), scale, ptr(loss), ptr(partial), ptr(ps),
                                                     ptr(grad) if with_grad else None))
        out[tag] = (loss.clone(), ps.clone(), grad)
    one = torch.ones(1, device=dev)
    ref = torch.empty(n, device=dev)
    _lib.call("fdgs_l1_loss_bwd", _lib.L1BwdArgs(n, ptr(a), ptr(b), scale, ptr(one), ptr(ref)))
    torch.cuda.synchronize()
    assert torch.equal(out["fused"][2], ref)
    assert torch.equal(out["fused"][0], out["plain"][0]) and torch.equal(out["fused"][1], out["plain"][1])
    assert torch.isnan(out["plain"][2]).all()         # not touched without the pointer
    expect = (scale / n) * torch.sign(a.double() - b.double())
    assert float((ref.double() - expect).abs().max()) <= 1e-6 * scale / n
