"""GPU: the straight C-ABI chain of fdgs/pipeline.py (what TrainStep runs by default) against the render() +
autograd route of the reference's interface: same loss, PSNR, every parameter gradient, view-space gradient,
radii and statistics, for one and for several views per step, in both stages."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n=6000, w=160, h=120):
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import DeformHolder
    dev = torch.device("cuda:0")
    g = ss.GaussianParams(n, device=dev, scale_mult=3.0)
    torch.manual_seed(ss.SEED)
    net = ss.perturb_deformation_(DeformationFields(ss.DYNERF_NET_KWARGS)).to(dev)
    net.deformation_net.set_aabb(list(ss.SCENE_MAX), list(ss.SCENE_MIN))
    deforms = DeformHolder(net)
    cams = ss.make_cameras(6, w, h, device=dev)
    bg = torch.ones(3, device=dev)
    g2 = ss.GaussianParams(n, seed=ss.SEED + 100, device=dev, requires_grad=False, scale_mult=3.0)
    with torch.no_grad():
        for cam in cams:
            cam.original_image = render(cam, g2, deforms, bg, stage="coarse")["render"].clamp(0, 1).clone()
    return g, deforms, cams, bg


def _one_step(use_pipeline, stage, n_views):
    from fdgs.train_step import TrainStep
    g, deforms, cams, bg = _setup()
    step = TrainStep(g, deforms, bg, stage=stage)
    step.use_pipeline = use_pipeline
    rec = {}

    def record():
        for grp in step.optimizer.param_groups:
            for i, p in enumerate(grp["params"]):
                rec[f"{grp['name']}{i}"] = None if p.grad is None else p.grad.detach().clone()
    step.optimizer.step = record
    losses = [step(cams[:n_views], iteration=1 + k).clone() for k in range(2)]     # twice: buffers are reused
    torch.cuda.synchronize()
    rec.update(loss=torch.stack(losses), psnr=step.last_psnr.clone(), accum=step.xyz_gradient_accum.clone(),
               denom=step.denom.clone(), maxr=step.max_radii2D.clone())
    return rec


@pytest.mark.parametrize("stage", ["fine", "coarse"])
@pytest.mark.parametrize("n_views", [1, 3])
def test_direct_chain_matches_autograd_route(stage, n_views):
    a = _one_step(False, stage, n_views)
    b = _one_step(True, stage, n_views)
    assert a.keys() == b.keys()
    assert torch.equal(a["denom"], b["denom"]) and torch.equal(a["maxr"], b["maxr"])
    for k in a:
        if a[k] is None:            # the frozen box corners; in the coarse stage the whole deformation field
            assert b[k] is None, k
            continue
        ref, got = a[k].float(), b[k].float()
        assert got.shape == ref.shape, k
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-12, k
    assert sum(v is not None for v in a.values()) >= (45 if stage == "fine" else 11)


def test_training_step_with_the_dnerf_width():
    """net_width 64 (configs/train/dnerf.yaml:83): the module runs the 128-wide kernels on zero-padded weights, the
    step takes the render() + autograd route (the direct chain is built for the kernels' own width), and training
    moves every tensor of the narrow network - whose shapes stay the reference's."""
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import DeformHolder, TrainStep
    dev = torch.device("cuda:0")
    n, w, h = 4000, 160, 120
    g = ss.GaussianParams(n, device=dev, scale_mult=3.0)
    torch.manual_seed(ss.SEED)
    net = ss.perturb_deformation_(DeformationFields(dict(ss.DYNERF_NET_KWARGS, net_width=64))).to(dev)
    net.deformation_net.set_aabb(list(ss.SCENE_MAX), list(ss.SCENE_MIN))
    assert net.deformation_net.pos_deform[1].weight.shape == (64, 64)
    assert net.deformation_net.shs_deform[3].weight.shape == (48, 64)
    deforms = DeformHolder(net)
    cams = ss.make_cameras(4, w, h, device=dev)
    bg = torch.ones(3, device=dev)
    g2 = ss.GaussianParams(n, seed=ss.SEED + 100, device=dev, requires_grad=False, scale_mult=3.0)
    with torch.no_grad():
        for cam in cams:
            cam.original_image = render(cam, g2, deforms, bg, stage="coarse")["render"].clamp(0, 1).clone()
    step = TrainStep(g, deforms, bg)
    assert not step.pipeline.usable("fine") and step.pipeline.usable("coarse")
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    losses = [float(step(cams[:2], iteration=1 + k)) for k in range(3)]
    torch.cuda.synchronize()
    assert all(l == l and l > 0 for l in losses)
    moved = {k: not torch.equal(v.detach(), before[k]) for k, v in net.named_parameters() if v.requires_grad}
    assert len(moved) == 34 and all(moved.values()), [k for k, m in moved.items() if not m]
    for k, v in net.named_parameters():
        assert torch.isfinite(v).all(), k
