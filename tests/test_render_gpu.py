"""GPU: the render() drop-in end to end, including the host-side colour and
covariance flags of src/gaussian_renderer/__init__.py:21-120."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(n=4000, w=160, h=120):
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    dev = torch.device("cuda:0")
    cam = ss.make_cameras(3, w, h, device=dev)[1]
    pc = ss.GaussianParams(n, device=dev, scale_mult=4.0)
    torch.manual_seed(0)
    from fdgs.train_step import DeformHolder
    net = ss.perturb_deformation_(DeformationFields(ss.DYNERF_NET_KWARGS)).to(dev)
    return cam, pc, DeformHolder(net), torch.ones(3, device=dev)


def _grads(pc):
    out = [p.grad.clone() for p in pc.parameters()]
    for p in pc.parameters():
        p.grad = None
    return out


def test_render_contract_and_flags():
    from fdgs.renderer import render
    cam, pc, deforms, bg = _scene()
    gen = torch.Generator().manual_seed(3)
    gout = torch.randn(3, cam.image_height, cam.image_width, generator=gen).cuda()

    for stage in ("coarse", "fine"):
        out = render(cam, pc, deforms, bg, stage=stage)
        assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii", "depth"}
        assert out["render"].shape == (3, cam.image_height, cam.image_width)
        assert out["depth"].shape == (1, cam.image_height, cam.image_width)
        assert out["radii"].dtype == torch.int32 and out["visibility_filter"].dtype == torch.bool
        (out["render"] * gout).sum().backward()
        assert out["viewspace_points"].grad is not None and out["viewspace_points"].grad.shape == pc.get_xyz.shape
        base = (out["render"].detach(), out["radii"].clone(), _grads(pc))

        # SH evaluated on the host: same colours up to rounding (direction from the undeformed centre in
        # both, so only exact for the coarse stage)
        if stage == "coarse":
            o2 = render(cam, pc, deforms, bg, stage=stage, convert_SHs_python=True)
            assert torch.equal(o2["radii"], base[1])
            assert float((o2["render"] - base[0]).abs().max()) <= 1e-5
            (o2["render"] * gout).sum().backward()
            g2 = _grads(pc)
            for a, b in zip(g2, base[2]):
                assert float((a - b).abs().max() / (b.abs().max() + 1e-30)) <= 1e-4
            # covariance built on the host
            o3 = render(cam, pc, deforms, bg, stage=stage, compute_cov3D_python=True)
            assert float((o3["render"] - base[0]).abs().max()) <= 1e-4
            assert int((o3["radii"] != base[1]).sum()) <= max(2, base[1].numel() // 500)   # ceil(3 sigma) ties
            (o3["render"] * gout).sum().backward()
            g3 = _grads(pc)
            for a, b in zip(g3, base[2]):
                assert float((a - b).abs().max() / (b.abs().max() + 1e-30)) <= 2e-3
        else:
            with pytest.raises(ValueError):
                render(cam, pc, deforms, bg, stage=stage, compute_cov3D_python=True)

        # override_color: constant colour -> image is colour * (1 - T) + bg * T
        col = torch.full((pc.get_xyz.shape[0], 3), 0.25, device="cuda")
        o4 = render(cam, pc, deforms, bg, stage=stage, override_color=col)
        assert float(o4["render"].min()) >= 0.25 - 1e-5 and float(o4["render"].max()) <= 1.0 + 1e-5


def test_activations_inside_the_rasterizer_match_torch():
    """render() leaves exp / normalize / sigmoid (src/gaussian_renderer/__init__.py:83-86) to the preprocess
    kernels when the model uses the reference's own callables; wrapping them hides their identity and sends
    them through PyTorch as the reference does.  Both must give the same image and parameter gradients,
    and with the deformation field's parameters too in the fine stage."""
    from fdgs.renderer import render
    cam, pc, deforms, bg = _scene()
    with torch.no_grad():
        pc._rotation[:7] *= 3.0             # far from unit length
        pc._rotation[7:9] = 0.0             # degenerate: normalize() divides by its eps
    gen = torch.Generator().manual_seed(5)
    gout = torch.randn(3, cam.image_height, cam.image_width, generator=gen).cuda()
    net_params = list(deforms.deformation_fields.parameters())
    for stage in ("coarse", "fine"):
        res = {}
        for mode in ("kernel", "torch"):
            acts = (pc.scaling_activation, pc.rotation_activation, pc.opacity_activation)
            if mode == "torch":
                pc.scaling_activation = lambda t: torch.exp(t)
                pc.rotation_activation = lambda t: torch.nn.functional.normalize(t)
                pc.opacity_activation = lambda t: torch.sigmoid(t)
            try:
                out = render(cam, pc, deforms, bg, stage=stage)
                (out["render"] * gout).sum().backward()
            finally:
                pc.scaling_activation, pc.rotation_activation, pc.opacity_activation = acts
            ng = [p.grad.clone() for p in net_params if p.grad is not None]
            for p in net_params:
                p.grad = None
            res[mode] = (out["render"].detach(), out["radii"].clone(), out["viewspace_points"].grad.clone(), _grads(pc), ng)
        k, t = res["kernel"], res["torch"]
        assert int((k[1] != t[1]).sum()) <= 2                       # ceil(3 sigma) ties under 1-ulp scale changes
        assert float((k[0] - t[0]).abs().max()) <= 1e-5
        assert float((k[2] - t[2]).abs().max() / (t[2].abs().max() + 1e-30)) <= 1e-4
        assert len(k[4]) == len(t[4])
        for a, b in zip(k[3] + k[4], t[3] + t[4]):
            assert float((a - b).abs().max() / (b.abs().max() + 1e-30)) <= 1e-4


@pytest.mark.parametrize("n", [0, 1])
def test_render_empty_and_single_gaussian_scene(n):
    """A scene without Gaussians renders the background and back-propagates empty gradients (every tile range
    empty, NULL list and geometry pointers at the C boundary); one Gaussian goes through every stage."""
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import DeformHolder
    dev = torch.device("cuda:0")
    g = ss.GaussianParams(1, device=dev)
    if n == 0:
        for a in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
            setattr(g, a, torch.nn.Parameter(getattr(g, a).detach()[:0].clone()))
    torch.manual_seed(1)
    net = ss.perturb_deformation_(DeformationFields(ss.DYNERF_NET_KWARGS)).to(dev)
    net.deformation_net.set_aabb(list(ss.SCENE_MAX), list(ss.SCENE_MIN))
    cam = ss.make_cameras(2, 160, 120, device=dev)[0]
    bg = torch.tensor([1.0, 0.5, 0.25], device=dev)
    for stage in ("coarse", "fine"):
        pkg = render(cam, g, DeformHolder(net), bg, stage=stage)
        (pkg["render"] - 0.5).abs().mean().backward()
        torch.cuda.synchronize()
        assert pkg["render"].shape == (3, 120, 160) and pkg["radii"].shape == (n,)
        assert g._xyz.grad.shape == (n, 3) and pkg["viewspace_points"].grad.shape == (n, 3)
        if n == 0:
            assert torch.equal(pkg["render"], bg.view(3, 1, 1).expand(3, 120, 160))
        else:
            assert torch.isfinite(g._xyz.grad).all()


def test_saved_scene_plays_back_through_the_hip_path(tmp_path):
    """The reference's playback (render.py:114-142): Scene.load reads point_cloud.ply and deformation.pth
    (src/scene/scene.py:35-41, gaussian_model.py:131-171, deformation_wrapper.py:31-41) into fresh objects and
    render() draws them.  Everything stored is float32, so a saved and re-loaded scene renders bit-identically
    (images, depth, radii) to the scene in memory, at several time stamps."""
    import gc
    from fdgs import io, scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import DeformHolder
    cam, pc, deforms, bg = _scene(n=20000, w=320, h=240)
    net = deforms.deformation_fields
    net.deformation_net.set_aabb(list(ss.SCENE_MAX), list(ss.SCENE_MIN))
    model_dir = str(tmp_path / "point_cloud" / "iteration_7")
    io.save_ply(model_dir + "/point_cloud.ply", pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling,
                pc._rotation)
    table = torch.ones(20000, dtype=torch.bool, device="cuda:0")
    io.save_deformation(model_dir, net, table, torch.zeros(20000, 3, device="cuda:0"))

    dev = torch.device("cuda:0")
    loaded = io.load_ply(model_dir + "/point_cloud.ply", max_sh_degree=3, device=dev)
    pc2 = ss.GaussianParams(1, device=dev, requires_grad=False)       # fresh container, overwritten below
    pc2._xyz, pc2._features_dc, pc2._features_rest = loaded["xyz"], loaded["features_dc"], loaded["features_rest"]
    pc2._opacity, pc2._scaling, pc2._rotation = loaded["opacity"], loaded["scaling"], loaded["rotation"]
    net2 = DeformationFields(ss.DYNERF_NET_KWARGS).to(dev)           # fresh random weights, then the file's
    table2, accum2 = io.load_deformation(model_dir, net2, 20000, device=dev)
    assert torch.equal(table2, table) and accum2.shape == (20000, 3)
    for a, b in zip(pc.parameters(), pc2.parameters()):
        assert torch.equal(a.detach(), b)
    with torch.no_grad():
        for t in (0.0, 0.37, 299 / 300):
            cam.time = t
            a = render(cam, pc, deforms, bg)
            b = render(cam, pc2, DeformHolder(net2), bg)
            assert torch.equal(a["radii"], b["radii"]) and int((a["radii"] > 0).sum()) > 1000
            assert torch.equal(a["render"], b["render"]) and torch.equal(a["depth"], b["depth"])
    # a grad-enabled render that is never backpropagated must not pin its buffers (no cycle through the context)
    gc.collect()
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for _ in range(3):
        render(cam, pc, deforms, bg)
    gc.disable()
    try:
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() <= base + (1 << 20)
    finally:
        gc.enable()


def test_frames_to8b_match_numpy():
    """fdgs.playback.FrameSink against the reference's host-side conversion (render.py:47,78), bit for bit, over more
    frames than it has staging buffers, with values outside [0, 1] and on the k / 255 steps."""
    import numpy as np
    from fdgs.playback import FrameSink
    dev = torch.device("cuda:0")
    H, W = 37, 53
    gen = torch.Generator().manual_seed(5)
    frames = []
    for i in range(5):
        x = torch.rand(3, H, W, generator=gen) * 1.4 - 0.2
        x[0, 0, :] = torch.arange(W, dtype=torch.float32) / 255.0           # exact steps
        x[1, 1, :] = (torch.arange(W, dtype=torch.float32) + 200) / 255.0
        x[2, 2, :4] = torch.tensor([0.0, 1.0, -0.0, 0.99999994])
        frames.append(x)
    sink = FrameSink(len(frames), H, W, device=dev)
    for x in frames:
        sink.push(x.to(dev))
    out = sink.frames()
    assert len(out) == len(frames)
    for x, got in zip(frames, out):
        ref = (255 * np.clip(x.numpy(), 0, 1)).astype(np.uint8).transpose(1, 2, 0)
        assert got.dtype == np.uint8 and got.shape == (H, W, 3)
        assert np.array_equal(got, ref)
    with pytest.raises(IndexError):
        sink.push(frames[0].to(dev))


@pytest.mark.parametrize("pipelined", [True, False])
def test_render_sequence_gives_the_frames_of_the_host_loop(pipelined):
    """render_sequence (render.py:74-79) against render() + the host conversion, frame by frame; `pipelined`: the next
    frame's deformation and preprocess are enqueued before the current frame's list length is waited for."""
    import numpy as np
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.playback import render_sequence
    from fdgs.renderer import render
    from fdgs.train_step import DeformHolder
    dev = torch.device("cuda:0")
    g = ss.GaussianParams(4000, device=dev, scale_mult=3.0, requires_grad=False)
    torch.manual_seed(3)
    net = DeformationFields(ss.DYNERF_NET_KWARGS)
    ss.perturb_deformation_(net)
    net = net.to(dev)
    deforms = DeformHolder(net)
    cams = ss.make_cameras(5, 96, 64, device=dev)
    bg = torch.ones(3, device=dev)
    images, renders = render_sequence(cams, g, deforms, bg, keep_renders=True, pipelined=pipelined)
    assert len(images) == len(cams) == len(renders)
    only_renders = render_sequence(cams[:2], g, deforms, bg, keep_renders=True, to8b=False, pipelined=pipelined)
    assert only_renders[0] == [] and torch.equal(only_renders[1][1], renders[1])
    with torch.no_grad():
        for cam, img, kept in zip(cams, images, renders):
            x = render(cam, g, deforms, bg)["render"]
            assert torch.equal(x, kept)
            ref = (255 * np.clip(x.cpu().numpy(), 0, 1)).astype(np.uint8).transpose(1, 2, 0)
            assert np.array_equal(img, ref)


def test_no_grad_render_without_the_sh_concatenation():
    """Under torch.no_grad() render() hands _features_dc / _features_rest to the deformation kernel as they are; the
    image is the one of the route through get_features, bit for bit."""
    from fdgs import deformation, scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import DeformHolder
    dev = torch.device("cuda:0")
    g = ss.GaussianParams(5000, device=dev, scale_mult=3.0)
    torch.manual_seed(3)
    net = DeformationFields(ss.DYNERF_NET_KWARGS)
    ss.perturb_deformation_(net)
    net = net.to(dev)
    deforms = DeformHolder(net)
    cam = ss.make_cameras(3, 128, 96, device=dev)[1]
    bg = torch.ones(3, device=dev)
    with torch.no_grad():
        a = render(cam, g, deforms, bg)
        n = g._xyz.shape[0]
        outs = net(g._xyz, g._scaling, g._rotation, g._opacity, deformation.SplitSH(g._features_dc, g._features_rest),
                   float(cam.time))
        ref = net(g._xyz, g._scaling, g._rotation, g._opacity, g.get_features, float(cam.time))
    for x, y in zip(outs, ref):
        assert x.shape == y.shape and torch.equal(x, y)
    b = render(cam, g, deforms, bg)          # with gradients: the concatenated tensor
    assert torch.equal(a["render"], b["render"].detach()) and torch.equal(a["radii"], b["radii"])
    with pytest.raises(ValueError):
        net(g._xyz, g._scaling, g._rotation, g._opacity, deformation.SplitSH(g._features_dc, g._features_rest),
            float(cam.time))
