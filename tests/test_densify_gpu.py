"""GPU parity: densification / pruning surgery through libfdgs.so against oracle/densify_ref.py (the
reference's clone -> split -> prune sequence restated step by step) on the same seeded state and the same
unit normal samples.  Copied rows, optimizer moments, the deformation table and all counts: bit-exact.
Positions and log-scales of the split children: 1e-6 relative (exp / log / divisions in another order)."""
import math
import time
import types

import numpy as np
import pytest
import torch

import densify_cases as dc
from oracle import densify_ref as dr

pytestmark = pytest.mark.gpu

CHILD_TOL = 1e-6
EXTENT, PERCENT_DENSE, MAX_GRAD = 5.0, 0.01, 2e-4


def _to_gpu(st, fused):
    """(gaussians, optimizer, deforms, surgery) holding a copy of the numpy state on the GPU."""
    from fdgs.adam import FusedAdam
    from fdgs.densify import GaussianSurgery, _ATTR
    dev = torch.device("cuda:0")
    g = types.SimpleNamespace(percent_dense=PERCENT_DENSE)
    groups = []
    for k in dc.NAMES:
        p = torch.nn.Parameter(torch.from_numpy(st["params"][k]).to(dev))
        setattr(g, _ATTR[k], p)
        groups.append({"params": [p], "lr": 1e-3, "name": k})
    extra = torch.nn.Parameter(torch.ones(4, 4, device=dev))           # a multi-tensor group like "deformation"
    groups.insert(1, {"params": [extra, torch.nn.Parameter(torch.ones(3, device=dev))], "lr": 1e-3, "name": "deformation"})
    opt = FusedAdam(groups, lr=0.0, eps=1e-15) if fused else torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    if st.get("exp_avg") is not None:
        for grp in groups:
            if grp["name"] in dc.NAMES:
                k = grp["name"]
                entry = {"exp_avg": torch.from_numpy(st["exp_avg"][k]).to(dev),
                         "exp_avg_sq": torch.from_numpy(st["exp_avg_sq"][k]).to(dev)}
                entry["step"] = 3 if fused else torch.tensor(3.0)
                opt.state[grp["params"][0]] = entry
    for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
        setattr(g, k, torch.from_numpy(st[k]).to(dev))
    d = types.SimpleNamespace(deformation_table=torch.from_numpy(st["deformation_table"]).to(dev),
                              deformation_accum=torch.from_numpy(st["deformation_accum"]).to(dev))
    return g, opt, d, GaussianSurgery(g, opt, deforms=d)


def _compare(st, g, opt, d, child_rows=None):
    from fdgs.densify import _ATTR
    n = st["params"]["xyz"].shape[0]
    groups = {grp["name"]: grp for grp in opt.param_groups}
    for k in dc.NAMES:
        p = getattr(g, _ATTR[k])
        assert p is groups[k]["params"][0] and isinstance(p, torch.nn.Parameter) and p.requires_grad
        got, want = p.detach().cpu().numpy(), st["params"][k]
        assert got.shape == want.shape, k
        if child_rows is not None and k in ("xyz", "scaling"):
            assert np.array_equal(got[:child_rows], want[:child_rows]), k
            err = np.abs(got[child_rows:] - want[child_rows:]).max() if n > child_rows else 0.0
            ref = max(np.abs(want[child_rows:]).max(), 1.0) if n > child_rows else 1.0
            assert err <= CHILD_TOL * ref, (k, err)
        else:
            assert np.array_equal(got, want), k
        if st.get("exp_avg") is not None:
            entry = opt.state[p]
            assert np.array_equal(entry["exp_avg"].cpu().numpy(), st["exp_avg"][k]), k
            assert np.array_equal(entry["exp_avg_sq"].cpu().numpy(), st["exp_avg_sq"][k]), k
    for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
        got = getattr(g, k).cpu().numpy()
        assert got.shape == st[k].shape and np.array_equal(got, st[k]), k
    assert np.array_equal(d.deformation_table.cpu().numpy(), st["deformation_table"])
    assert d.deformation_table.dtype is torch.bool
    assert np.array_equal(d.deformation_accum.cpu().numpy(), st["deformation_accum"])


@pytest.mark.parametrize("n,fused,moments", [(5000, True, True), (5000, False, True), (1031, True, False), (1, True, True),
                                             (2048, False, True)])
def test_densify_matches_reference_sequence(n, fused, moments):
    st = dc.make_state(n, seed=n, moments=moments)
    g, opt, d, surgery = _to_gpu(st, fused)
    rng = np.random.default_rng(n + 1)
    noise = rng.standard_normal((2 * n, 3)).astype(np.float32)
    used = {}

    def noise_fn(count):
        used["count"] = count
        return noise[:2 * count]
    clone_sel, split_sel = dr.densify(st, MAX_GRAD, EXTENT, PERCENT_DENSE, noise_fn)
    s = int(split_sel.sum())
    surgery.noise_fn = lambda count, device: torch.from_numpy(noise[:count]).to(device)
    out = surgery.densify(MAX_GRAD, EXTENT)
    assert out["clones"] == int(clone_sel.sum()) and out["splits"] == s
    assert out["rows"] == st["params"]["xyz"].shape[0] == n + out["clones"] + s
    if n >= 1000:
        assert out["clones"] > n // 20 and s > n // 20         # the case exercises both branches
    _compare(st, g, opt, d, child_rows=out["rows"] - 2 * s)
    # the optimizer still steps on the rebuilt tensors
    for grp in opt.param_groups:
        for p in grp["params"]:
            p.grad = torch.ones_like(p)
    opt.step()
    torch.cuda.synchronize()
    for k in dc.NAMES:
        p = [grp for grp in opt.param_groups if grp["name"] == k][0]["params"][0]
        assert opt.state[p]["exp_avg"].shape == p.shape


@pytest.mark.parametrize("max_screen_size", [None, 20])
@pytest.mark.parametrize("n", [4097, 3])
def test_prune_matches_reference(n, max_screen_size):
    st = dc.make_state(n, seed=11 + n)
    g, opt, d, surgery = _to_gpu(st, True)
    mask = dr.prune(st, 0.2, EXTENT, max_screen_size)
    out = surgery.prune(0.2, EXTENT, max_screen_size)
    assert out["rows"] == n - int(mask.sum()) and out["clones"] == 0 and out["splits"] == 0
    _compare(st, g, opt, d)


@pytest.mark.parametrize("which", ["none", "all", "every_third"])
def test_prune_points_edge_masks(which):
    n = 2500
    st = dc.make_state(n, seed=21)
    g, opt, d, surgery = _to_gpu(st, False)
    mask = {"none": np.zeros(n, bool), "all": np.ones(n, bool), "every_third": np.arange(n) % 3 == 0}[which]
    dr.prune_points(st, mask)
    out = surgery.prune_points(torch.from_numpy(mask).cuda())
    assert out["rows"] == n - int(mask.sum())
    _compare(st, g, opt, d)


def test_clone_only_and_split_only():
    n = 3000
    base = dc.make_state(n, seed=31)
    rng = np.random.default_rng(32)
    noise = rng.standard_normal((2 * n, 3)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        grads = base["xyz_gradient_accum"] / base["denom"]
    grads[np.isnan(grads)] = 0.0
    st = dc.copy_state(base)
    g, opt, d, surgery = _to_gpu(st, True)
    dr.densify_and_clone(st, grads, MAX_GRAD, EXTENT, PERCENT_DENSE)
    surgery.densify_and_clone(MAX_GRAD, EXTENT)
    _compare(st, g, opt, d)
    st = dc.copy_state(base)
    g, opt, d, surgery = _to_gpu(st, True)
    sel = np.logical_and(grads.reshape(-1) >= np.float32(MAX_GRAD),
                         np.exp(st["params"]["scaling"]).max(1) > np.float32(PERCENT_DENSE * EXTENT))
    s = int(sel.sum())
    dr.densify_and_split(st, grads, MAX_GRAD, EXTENT, PERCENT_DENSE, noise[:2 * s])
    out = surgery.densify_and_split(MAX_GRAD, EXTENT, noise=torch.from_numpy(noise[:2 * s]).cuda())
    assert out["splits"] == s and out["clones"] == 0
    _compare(st, g, opt, d, child_rows=out["rows"] - 2 * s)


def test_reset_opacity():
    st = dc.make_state(4000, seed=41)
    g, opt, d, surgery = _to_gpu(st, True)
    dr.reset_opacity(st)
    surgery.reset_opacity()
    got = g._opacity.detach().cpu().numpy()
    assert np.abs(got - st["params"]["opacity"]).max() <= 1e-5 * np.abs(st["params"]["opacity"]).max()
    entry = opt.state[g._opacity]
    assert not entry["exp_avg"].any() and not entry["exp_avg_sq"].any()
    assert opt.state[g._xyz]["exp_avg"].any()


def test_abi_rejects_bad_arguments():
    from fdgs import _lib
    a = _lib.SurgeryApplyArgs()
    a.rows_after, a.rows_carried, a.num_arrays = 4, 5, 0
    assert _lib.lib.fdgs_surgery_apply(a, None) == _lib.lib.fdgs_surgery_apply(a, None) != 0
    p = _lib.SurgeryPlanArgs()
    p.num_points = 10
    assert _lib.lib.fdgs_surgery_plan(p, None) != 0
    assert _lib.lib.fdgs_reset_opacity(None, 5, 0.01, None) != 0
    assert _lib.lib.fdgs_reset_opacity(None, 0, 0.01, None) == 0


def test_full_size_densify_then_train_step():
    """300,000 Gaussians at 1352 x 1014: a few training steps fill the statistics, one densification and one
    pruning pass rebuild every array, and training goes on with the new set.  Checked without the oracle:
    counts add up, survivors are untouched, clones equal their sources, moments of added rows are zero."""
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import DeformHolder, TrainStep
    dev = torch.device("cuda:0")
    n = 300000
    g = ss.GaussianParams(n, device=dev)
    torch.manual_seed(ss.SEED)
    net = DeformationFields(ss.DYNERF_NET_KWARGS)
    ss.perturb_deformation_(net)
    net = net.to(dev)
    with torch.no_grad():
        net.deformation_net.set_aabb(g._xyz.detach().max(0).values.tolist(), g._xyz.detach().min(0).values.tolist())
    deforms = DeformHolder(net)
    cams = ss.make_cameras(3, 1352, 1014, device=dev)
    bg = torch.ones(3, device=dev)
    g2 = ss.GaussianParams(n, seed=ss.SEED + 100, device=dev, requires_grad=False)
    with torch.no_grad():
        for cam in cams:
            cam.original_image = render(cam, g2, deforms, bg, stage="coarse")["render"].clamp(0, 1).clone()
    step = TrainStep(g, deforms, bg)
    for i in range(3):
        step([cams[i % len(cams)]])
    seen = step.denom.reshape(-1) > 0
    assert seen.any()
    grads = (step.xyz_gradient_accum / step.denom).nan_to_num(0.0).reshape(-1)
    thr = float(grads[seen].median())
    old = {k: getattr(g, a).detach().clone() for k, a in
           (("xyz", "_xyz"), ("f_rest", "_features_rest"), ("scaling", "_scaling"))}
    old_m = step.optimizer.state[g._features_rest]["exp_avg"].clone()
    big = old["scaling"].exp().max(1).values > 0.01 * 5.0
    want_clone = (grads >= thr) & ~big
    want_split = (grads >= thr) & big
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = step.densify(thr, 5.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("densify of %d -> %d rows: %.2f ms" % (n, out["rows"], dt * 1e3))
    c, s = int(want_clone.sum()), int(want_split.sum())
    assert (out["clones"], out["splits"]) == (c, s) and c > 0 and s > 0
    k = n - s
    assert out["survivors"] == k and out["rows"] == k + c + 2 * s == g._xyz.shape[0]
    assert torch.equal(g._xyz[:k], old["xyz"][~want_split])
    assert torch.equal(g._features_rest[k:k + c], old["f_rest"][want_clone])
    assert torch.equal(g._features_rest[k + c:k + c + s], old["f_rest"][want_split])
    assert torch.equal(g._features_rest[k + c + s:], old["f_rest"][want_split])
    m = step.optimizer.state[g._features_rest]["exp_avg"]
    assert torch.equal(m[:k], old_m[~want_split]) and not m[k:].any()
    # children: inside three standard deviations of the parent along its largest axis, scale / 1.6
    assert torch.allclose(g._scaling[k + c:k + c + s].exp() * 1.6, old["scaling"][want_split].exp(), rtol=1e-5)
    d = (g._xyz[k + c:k + c + s] - old["xyz"][want_split]).norm(dim=1)
    assert (d <= 6.0 * old["scaling"][want_split].exp().max(1).values * 3 ** 0.5).all() and d.max() > 0
    assert step.denom.shape == (out["rows"], 1) and not step.denom.any()
    for i in range(2):
        loss = step([cams[i]])
    assert torch.isfinite(loss)
    n1 = g._xyz.shape[0]
    out = step.prune(0.05, 5.0, 20)
    assert out["rows"] == g._xyz.shape[0] < n1 and step.max_radii2D.shape == (out["rows"],)
    step.reset_opacity()
    assert torch.sigmoid(g._opacity).max() <= 0.01 + 1e-6
    loss = step([cams[2]])
    torch.cuda.synchronize()
    assert torch.isfinite(loss)


def test_training_loop_with_surgery_converges():
    """A short run of the reference's loop shape (fdgs_trainer.py:112-205) on a small synthetic scene: steps,
    densification every 20 iterations, one pruning pass, one opacity reset; the L1 loss against the target
    renders must fall and the PSNR rise."""
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import DeformHolder, TrainStep
    dev = torch.device("cuda:0")
    n, w, h = 4000, 160, 120
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
    step = TrainStep(g, deforms, bg)
    losses, psnrs, rows = [], [], [n]
    for it in range(1, 91):
        losses.append(float(step([cams[it % len(cams)]])))
        psnrs.append(float(step.last_psnr))
        if it % 20 == 0 and it < 70:
            seen = step.denom.reshape(-1) > 0
            grads = (step.xyz_gradient_accum / step.denom).nan_to_num(0.0).reshape(-1)
            out = step.densify(float(grads[seen].quantile(0.9)), 5.0)
            rows.append(out["rows"])
        if it == 70:
            rows.append(step.prune(0.005, 5.0, None)["rows"])
        if it == 75:
            step.reset_opacity()
            op_reset = g._opacity.detach().clone()
    torch.cuda.synchronize()
    assert rows[1] > rows[0] and rows[2] > rows[1] and rows[-1] <= rows[-2]
    assert g._xyz.shape[0] == rows[-1] == step.denom.shape[0]
    # iterations 1-6 against 69-74 (one pass over the six views each; the opacity reset at 75 makes every Gaussian
    # nearly transparent, after which the loss must only stay finite and start falling again)
    first, last = sum(losses[:6]) / 6, sum(losses[68:74]) / 6
    assert all(math.isfinite(v) for v in losses) and last < 0.8 * first, (first, last, losses)
    assert sum(psnrs[68:74]) > sum(psnrs[:6])
    # after the reset the optimizer must be acting on the re-installed opacity tensor (with zeroed moments and the
    # shared step count Adam's next updates are 0.85 to 1.8 learning rates whatever the gradient's magnitude:
    # 15 steps of 0.05)
    groups = {grp["name"]: grp for grp in step.optimizer.param_groups}
    assert groups["opacity"]["params"][0] is g._opacity
    moved = (g._opacity.detach() - op_reset).abs()
    assert 0.3 < float(moved.max()) < 2.0
    print("loss %.4f -> %.4f, psnr %.2f -> %.2f, rows %s; after the reset %.4f -> %.4f" %
          (first, last, sum(psnrs[:6]) / 6, sum(psnrs[68:74]) / 6, rows, sum(losses[75:81]) / 6, sum(losses[84:90]) / 6))


@pytest.mark.parametrize("tag", ["densify", "prune", "prune_no_size", "prune_points"])
def test_against_reference_methods(tag):
    """libfdgs surgery against tests/golden/densify_small.npz: the outputs of the reference's own
    Gaussian4DTrainer._densify / _prune / _prune_points (run on the CPU by tests/golden/make_densify_golden.py)
    for the same seeded state and the same unit normal samples."""
    import test_oracle_densify as tod
    from fdgs.densify import _ATTR
    c = tod.GOLD_CASES[tag]
    st = dc.make_state(c["n"], seed=c["seed"])
    g, opt, d, surgery = _to_gpu(st, fused=(tag != "prune"))
    children = 0
    if tag == "densify":
        noise = torch.from_numpy(tod.GOLD["densify_unit_noise"]).cuda()
        out = surgery.densify(c["max_grad"], c["extent"], noise=noise)
        children = 2 * out["splits"]
        assert children == noise.shape[0]
    elif tag == "prune_points":
        out = surgery.prune_points((torch.arange(c["n"]) % 3 == 1).cuda())
    else:
        out = surgery.prune(c["min_opacity"], c["extent"], c["max_screen_size"])

    def get(kind, name):
        if kind == "param":
            return getattr(g, _ATTR[name]).detach().cpu().numpy()
        if kind == "stat":
            holder = d if name.startswith("deformation") else g
            return getattr(holder, name).cpu().numpy()
        return opt.state[getattr(g, _ATTR[name])][kind].cpu().numpy()
    tod.check_against_golden(tag, get, child_rows=out["rows"] - children if children else None, tol=CHILD_TOL)
