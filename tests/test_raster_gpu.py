"""GPU parity: the HIP rasterizer (through the C ABI) against oracle/raster_ref.py
on the same seeded inputs.  Integer stages bit-exact, image <= 1e-5 max-abs,
gradients <= 1e-4 relative (max-norm per tensor)."""
import pytest
import torch

import raster_cases as rc
from oracle import raster_ref as rr

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-5
GRAD_TOL = 1e-4


def _to_dev(S, dev):
    return S._replace(bg=S.bg.to(dev), viewmatrix=S.viewmatrix.to(dev), projmatrix=S.projmatrix.to(dev),
                      campos=S.campos.to(dev))


def _run_oracle(c, with_grad):
    ins = {k: c[k].clone().requires_grad_(with_grad) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    m2 = torch.zeros(c["means3D"].shape[0], 3, requires_grad=with_grad)
    color, radii, depth, aux = rr.rasterize(ins["means3D"], m2, ins["shs"], ins["opacities"], ins["scales"],
                                            ins["rotations"], c["S"], return_aux=True)
    return ins, m2, color, radii, depth, aux


@pytest.mark.parametrize("sort", ["tiles", "radix"])
@pytest.mark.parametrize("name", ["small", "ragged", "deg0", "medium", "n256", "n2049", "n4096"])
def test_forward_stages(name, sort, monkeypatch):
    """`sort`: the tile lists by binning + per-tile sort in LDS (the default) or by the radix sort over all keys."""
    from fdgs import rasterizer as fr
    monkeypatch.setenv("FDGS_SORT", sort)
    c = rc.make_case(name)
    _, _, color, radii, depth, aux = _run_oracle(c, False)
    dev = torch.device("cuda:0")
    S = _to_dev(c["S"], dev)
    st = fr.rasterize_forward(c["means3D"].to(dev), c["shs"].to(dev), c["opacities"].reshape(-1).to(dev).contiguous(),
                              c["scales"].to(dev), c["rotations"].to(dev), S, keep_keys=True)
    torch.cuda.synchronize()
    # stage 1: radii / tiles_touched / scan, bit-exact
    assert torch.equal(st.radii.cpu(), radii)
    assert torch.equal(st.tiles_touched.cpu(), aux["pre"]["tiles_touched"])
    assert torch.equal(st.point_offsets.cpu(), aux["offsets"])
    R = aux["keys"].numel()
    assert st.num_rendered == R
    # float per-Gaussian outputs of visible Gaussians, bit-exact (no contraction on either side)
    vis = radii > 0
    g = st.geom.cpu()
    pre = aux["pre"]
    for col, ref in ((0, pre["px"]), (1, pre["py"]), (2, pre["conA"]), (3, pre["conB"]), (4, pre["conC"]),
                     (9, pre["depth"])):
        assert torch.equal(g[vis, col], ref[vis]), f"geom column {col}"
    # colour is not an integer-feeding quantity: tolerance, not bit equality
    assert float((g[vis, 6:9] - pre["rgb"][vis]).abs().max()) <= 1e-6
    # stages 3-5: keys, sort order, ranges, bit-exact
    assert torch.equal(st.keys_sorted[:R].cpu(), aux["sorted_keys"])
    assert torch.equal(st.vals_sorted[:R].cpu(), aux["sorted_vals"])
    assert torch.equal(st.ranges.cpu(), aux["ranges"])
    # the launch order of the blend kernels: a permutation of the tiles, longest lists first (to one length class)
    order = st.tile_order.cpu().long()
    assert torch.equal(torch.sort(order)[0], torch.arange(order.numel()))
    lens = (aux["ranges"][:, 1] - aux["ranges"][:, 0]).long()[order]
    assert bool((lens[1:].float() <= lens[:-1].float() * 1.07 + 1).all())
    # stage 6
    assert torch.equal(st.n_contrib.cpu(), aux["n_contrib"])
    assert float((st.final_T.cpu() - aux["final_T"]).abs().max()) <= IMG_TOL
    assert float((st.color.cpu() - color).abs().max()) <= IMG_TOL
    dtol = IMG_TOL * max(1.0, float(depth.abs().max()))
    assert float((st.depth.cpu() - depth).abs().max()) <= dtol


@pytest.mark.parametrize("n,w,h,scale_mult,tied", [
    (10000, 64, 48, 8.0, 0),         # 12 tiles, lists of a few thousand entries: the large workgroups' LDS path
    (10000, 64, 48, 8.0, 2500),      # ... with 2500 Gaussians at one depth (ranked by id inside their depth field)
    (30000, 32, 32, 8.0, 0),         # 4 tiles, lists beyond every LDS segment: sorted where they lie in memory
    (30000, 32, 32, 8.0, 9000),
    (3000, 200, 120, 1.0, 700),      # short lists with ties (runs longer than the pairwise ranking takes: 64-bit network)
    (3000, 200, 120, 1.0, 20),       # ... with runs short enough to be ranked pairwise
    (3000, 3840, 2160, 12.0, 0),     # 32400 tiles: the walks' per-tile counters take 130 KB of LDS
    (3000, 2944, 2944, 12.0, 0),     # 33856 tiles, more than the counters hold: the call falls back to the radix sort
])
def test_tile_binned_sort_long_lists_and_equal_depths(n, w, h, scale_mult, tied, monkeypatch):
    """fdgs_raster_bin_sort by tile binning against the oracle's stable sort where the per-tile lists are long and
    where many Gaussians share one depth (equal keys keep the Gaussian order).  The call is made on its own, on the
    preprocess outputs of a forward pass that used the radix sort."""
    import math
    from fdgs import rasterizer as fr, scene_synth as ss, _lib
    from fdgs._lib import lib, ptr
    cam = ss.make_cameras(3, w, h)[1]
    g = ss.GaussianParams(n, scale_mult=scale_mult, requires_grad=False)
    means3D = g._xyz.clone()
    if tied:
        means3D[50:50 + tied] = means3D[50]
    S = rr.Settings(h, w, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.ones(3), 1.0,
                    cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center)
    ins = dict(means3D=means3D, shs=g.get_features.contiguous(), opacities=torch.sigmoid(g._opacity),
               scales=torch.exp(g._scaling), rotations=torch.nn.functional.normalize(g._rotation))
    pre = rr.preprocess(ins["means3D"], ins["scales"], ins["rotations"], ins["opacities"], ins["shs"], S)
    keys, vals, offsets = rr.build_keys(pre, S)
    skeys, svals = rr.sort_keys(keys, vals)
    gx, gy = rr.tile_grid(S)
    ranges = rr.tile_ranges(skeys, gx * gy)
    R = keys.numel()
    if tied:
        same = (skeys[1:] == skeys[:-1])
        assert int(same.sum()) >= tied // 2, "the case is meant to have equal keys"

    dev = torch.device("cuda:0")
    monkeypatch.setenv("FDGS_SORT", "radix")
    st = fr.rasterize_forward(ins["means3D"].to(dev), ins["shs"].to(dev), ins["opacities"].reshape(-1).to(dev),
                              ins["scales"].to(dev), ins["rotations"].to(dev), _to_dev(S, dev), keep_keys=True)
    torch.cuda.synchronize()
    assert st.num_rendered == R and torch.equal(st.keys_sorted[:R].cpu(), skeys)
    monkeypatch.setenv("FDGS_SORT", "tiles")
    ws_bytes = lib.fdgs_raster_sort_workspace_bytes(R)
    ka = torch.empty(R, device=dev, dtype=torch.int64)
    va = torch.empty(R, device=dev, dtype=torch.int32)
    ks = torch.zeros(R, device=dev, dtype=torch.int64)
    vs = torch.zeros(R, device=dev, dtype=torch.int32)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    rg = torch.empty_like(st.ranges)
    order = torch.empty_like(st.tile_order)
    _lib.call("fdgs_raster_bin_sort", _lib.BinSortArgs(n, w, h, R, ptr(st.geom), ptr(st.point_offsets), ptr(ka), ptr(va),
                                                       ptr(ks), ptr(vs), ptr(rg), ptr(ws), ws_bytes, 0, 0, ptr(order)))
    torch.cuda.synchronize()
    assert torch.equal(ks.cpu(), skeys)
    assert torch.equal(vs.cpu(), svals)
    assert torch.equal(rg.cpu(), ranges)
    assert torch.equal(torch.sort(order.cpu().long())[0], torch.arange(gx * gy))
    # the same call in its two parts (the counts before the list length is known, the rest after)
    ks2, vs2, rg2 = torch.zeros_like(ks), torch.zeros_like(vs), torch.empty_like(rg)
    ws2 = torch.empty_like(ws)
    first = _lib.BinSortArgs(n, w, h, 0, ptr(st.geom), ptr(st.point_offsets), None, None, None, None, None, ptr(ws2),
                             ws_bytes, 0, 0, None, _lib.BINSORT_COUNT)
    _lib.call("fdgs_raster_bin_sort", first)
    _lib.call("fdgs_raster_bin_sort", _lib.BinSortArgs(n, w, h, R, ptr(st.geom), ptr(st.point_offsets), ptr(ka), ptr(va),
                                                       ptr(ks2), ptr(vs2), ptr(rg2), ptr(ws2), ws_bytes, 0, 0,
                                                       ptr(order), _lib.BINSORT_FINISH))
    torch.cuda.synchronize()
    assert torch.equal(ks2, ks) and torch.equal(vs2, vs) and torch.equal(rg2, rg)


def test_tile_lists_of_random_scenes_match_the_radix_sort(monkeypatch):
    """Seeded random scenes (image sizes that are not multiples of the tile, 1 to 150,000 Gaussians, splat sizes from
    a fraction of a tile to most of the image, a third of the Gaussians at one position in some): the tile-binned
    lists, ranges and launch order against the radix sort's on the same preprocess outputs, bit for bit.  The radix
    sort itself is held to the oracle by the tests above; the binned call is made on its own and its outputs are only
    compared."""
    import math
    import random
    from fdgs import rasterizer as fr, scene_synth as ss, _lib
    from fdgs._lib import lib, ptr
    dev = torch.device("cuda:0")
    rng = random.Random(20240607)
    nonempty = 0
    for it in range(24):
        w = rng.choice([16, 17, 64, 100, 257, 640, 1352, 1920])
        h = rng.choice([16, 31, 48, 128, 360, 1014])
        n = rng.choice([1, 2, 63, 64, 65, 500, 1023, 1024, 1025, 5000, 40000, 150000])
        sm = rng.choice([0.3, 1.0, 3.0, 8.0, 20.0])
        if n * sm > 400000:
            sm = 1.0
        tied = rng.choice([0, 0, n // 3])
        cam = ss.make_cameras(5, w, h)[rng.randrange(5)]
        g = ss.GaussianParams(n, seed=it + 7, scale_mult=sm, requires_grad=False)
        xyz = g._xyz.clone()
        if tied > 1:
            xyz[:tied] = xyz[0]
        S = _to_dev(rr.Settings(h, w, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.ones(3), 1.0,
                                cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center), dev)
        monkeypatch.setenv("FDGS_SORT", "radix")
        st = fr.rasterize_forward(xyz.to(dev), g.get_features.contiguous().to(dev),
                                  torch.sigmoid(g._opacity).reshape(-1).to(dev), torch.exp(g._scaling).to(dev),
                                  torch.nn.functional.normalize(g._rotation).to(dev), S, keep_keys=True)
        torch.cuda.synchronize()
        R = st.num_rendered
        if R == 0:
            continue
        nonempty += 1
        monkeypatch.setenv("FDGS_SORT", "tiles")
        ws_bytes = lib.fdgs_raster_sort_workspace_bytes(R)
        ka = torch.empty(R, device=dev, dtype=torch.int64)
        va = torch.empty(R, device=dev, dtype=torch.int32)
        ks = torch.zeros(R, device=dev, dtype=torch.int64)
        vs = torch.zeros(R, device=dev, dtype=torch.int32)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        rg = torch.empty_like(st.ranges)
        order = torch.empty_like(st.tile_order)
        _lib.call("fdgs_raster_bin_sort", _lib.BinSortArgs(n, w, h, R, ptr(st.geom), ptr(st.point_offsets), ptr(ka),
                                                           ptr(va), ptr(ks), ptr(vs), ptr(rg), ptr(ws), ws_bytes, 0, 0,
                                                           ptr(order)))
        torch.cuda.synchronize()
        what = f"scene {it}: {w}x{h}, {n} Gaussians, scale x{sm}, {tied} at one position, {R} entries"
        assert torch.equal(ks, st.keys_sorted[:R]), what
        assert torch.equal(vs, st.vals_sorted[:R]), what
        assert torch.equal(rg, st.ranges), what
        assert torch.equal(torch.sort(order.long())[0], torch.arange(rg.shape[0], device=dev)), what
    assert nonempty >= 12


def test_scan_in_two_launches_gives_the_same_lists(monkeypatch):
    """Beyond 8192 groups of 256 Gaussians the scan of the tile counts takes a second launch (a one-workgroup scan of
    the group sums); forced here at a size the oracle handles."""
    from fdgs import rasterizer as fr
    c = rc.make_case("medium")
    _, _, _, radii, _, aux = _run_oracle(c, False)
    dev = torch.device("cuda:0")
    S = _to_dev(c["S"], dev)
    monkeypatch.setenv("FDGS_SCAN_FUSED_MAX_GROUPS", "0")
    st = fr.rasterize_forward(c["means3D"].to(dev), c["shs"].to(dev), c["opacities"].reshape(-1).to(dev).contiguous(),
                              c["scales"].to(dev), c["rotations"].to(dev), S, keep_keys=True)
    torch.cuda.synchronize()
    R = aux["keys"].numel()
    assert st.num_rendered == R
    assert torch.equal(st.point_offsets.cpu(), aux["offsets"])
    assert torch.equal(st.keys_sorted[:R].cpu(), aux["sorted_keys"])
    assert torch.equal(st.vals_sorted[:R].cpu(), aux["sorted_vals"])
    assert torch.equal(st.ranges.cpu(), aux["ranges"])


@pytest.mark.parametrize("name", ["small", "ragged", "deg0", "medium", "n2049"])
def test_backward(name):
    from fdgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    c = rc.make_case(name)
    ins, m2, color, radii, depth, aux = _run_oracle(c, True)
    gen = torch.Generator().manual_seed(1)
    gout = torch.randn(color.shape, generator=gen)
    (color * gout).sum().backward()

    dev = torch.device("cuda:0")
    S = c["S"]
    gs = GaussianRasterizationSettings(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.bg.to(dev),
                                       S.scale_modifier, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.sh_degree,
                                       S.campos.to(dev), False, False)
    d = {k: c[k].to(dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    m2d = torch.zeros(c["means3D"].shape[0], 3, device=dev, requires_grad=True)
    img, rad, dep = GaussianRasterizer(gs)(means3D=d["means3D"], means2D=m2d, shs=d["shs"], colors_precomp=None,
                                           opacities=d["opacities"], scales=d["scales"], rotations=d["rotations"],
                                           cov3D_precomp=None)
    assert img.shape == color.shape and rad.dtype == torch.int32 and dep.shape == (1, S.image_height, S.image_width)
    (img * gout.to(dev)).sum().backward()
    torch.cuda.synchronize()
    pairs = [("means2D", m2d.grad, m2.grad)] + [(k, d[k].grad, ins[k].grad) for k in d]
    for k, got, ref in pairs:
        got = got.cpu()
        assert got.shape == ref.shape, k
        assert torch.isfinite(got).all(), k
        rel = float((got - ref).abs().max() / (ref.abs().max() + 1e-30))
        assert rel <= GRAD_TOL, f"{k}: rel {rel:.3e}"


@pytest.mark.parametrize("name", ["ragged", "medium", "n4096"])
def test_backward_with_the_forward_pass_reach_masks(name):
    """The backward blend kernel either repeats the test "which 4x4 pixel blocks does this entry reach" or reads the
    forward kernel's verdicts (reach_masks): both ways give the same gradients (float atomics: to rounding), and the
    recorded bits are the ones the backward's own test finds."""
    from fdgs import rasterizer as fr
    c = rc.make_case(name)
    dev = torch.device("cuda:0")
    S = _to_dev(c["S"], dev)
    ins = [c["means3D"].to(dev), c["shs"].to(dev), c["opacities"].reshape(-1).to(dev).contiguous(),
           c["scales"].to(dev), c["rotations"].to(dev)]
    gout = torch.randn(3, S.image_height, S.image_width, generator=torch.Generator().manual_seed(3)).to(dev)
    grads = []
    for keep in (False, True):
        st = fr.rasterize_forward(*ins, S, keep_masks=keep)
        assert (st.reach_masks is not None) == keep
        grads.append(fr.rasterize_backward(st, gout, ins[0], ins[1], ins[3], ins[4], S))
    torch.cuda.synchronize()
    for k, ref in grads[0].items():
        got = grads[1][k]
        if ref is None:
            assert got is None
            continue
        rel = float((got - ref).abs().max() / (ref.abs().max() + 1e-30))
        assert rel <= 1e-5, f"{k}: rel {rel:.3e}"


def test_no_grad_and_empty():
    """Works under torch.no_grad (render.py:136) and with nothing visible."""
    from fdgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    c = rc.make_case("deg0")
    dev = torch.device("cuda:0")
    S = c["S"]
    gs = GaussianRasterizationSettings(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.bg.to(dev),
                                       S.scale_modifier, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.sh_degree,
                                       S.campos.to(dev), False, False)
    far = c["means3D"].clone()
    far[:, 2] = -50.0    # everything behind the camera
    with torch.no_grad():
        img, rad, dep = GaussianRasterizer(gs)(means3D=far.to(dev), means2D=torch.zeros_like(far).to(dev),
                                               shs=c["shs"].to(dev), colors_precomp=None,
                                               opacities=c["opacities"].to(dev), scales=c["scales"].to(dev),
                                               rotations=c["rotations"].to(dev), cov3D_precomp=None)
    torch.cuda.synchronize()
    assert int((rad > 0).sum()) == 0
    assert torch.allclose(img.cpu(), S.bg.reshape(3, 1, 1).expand(3, S.image_height, S.image_width))
    assert float(dep.abs().max()) == 0.0


def _cov6(scales, rotations):
    """Upper triangle of (R S)(R S)^T, float64 then rounded once, so both sides get identical inputs."""
    q = rotations.double()
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * scales.double()[:, None, :]
    c = L @ L.transpose(1, 2)
    return torch.stack([c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2]], 1).float().contiguous()


@pytest.mark.parametrize("name", ["small", "ragged"])
def test_precomputed_color_and_covariance(name):
    """colors_precomp / cov3D_precomp inputs (src/gaussian_renderer/__init__.py:66-101,104-113):
    same integer stages, image and gradients as the oracle given the same precomputed tensors."""
    from fdgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    c = rc.make_case(name)
    N = c["means3D"].shape[0]
    gen = torch.Generator().manual_seed(7)
    colors = torch.rand(N, 3, generator=gen) * 1.2 - 0.1        # some negative: used as is, no clamp
    cov = _cov6(c["scales"], c["rotations"])
    S = c["S"]

    ins = {"means3D": c["means3D"].clone().requires_grad_(True), "opacities": c["opacities"].clone().requires_grad_(True),
           "colors": colors.clone().requires_grad_(True), "cov": cov.clone().requires_grad_(True)}
    m2 = torch.zeros(N, 3, requires_grad=True)
    color, radii, depth, aux = rr.rasterize(ins["means3D"], m2, None, ins["opacities"], None, None, S, return_aux=True,
                                            colors_precomp=ins["colors"], cov3D_precomp=ins["cov"])
    gout = torch.randn(color.shape, generator=gen)
    (color * gout).sum().backward()

    dev = torch.device("cuda:0")
    gs = GaussianRasterizationSettings(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.bg.to(dev),
                                       S.scale_modifier, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.sh_degree,
                                       S.campos.to(dev), False, False)
    d = {"means3D": c["means3D"].to(dev).requires_grad_(True), "opacities": c["opacities"].to(dev).requires_grad_(True),
         "colors": colors.to(dev).requires_grad_(True), "cov": cov.to(dev).requires_grad_(True)}
    m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
    img, rad, dep = GaussianRasterizer(gs)(means3D=d["means3D"], means2D=m2d, shs=None, colors_precomp=d["colors"],
                                           opacities=d["opacities"], scales=None, rotations=None,
                                           cov3D_precomp=d["cov"])
    (img * gout.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(rad.cpu(), radii)
    assert float((img.detach().cpu() - color.detach()).abs().max()) <= IMG_TOL
    dtol = IMG_TOL * max(1.0, float(depth.abs().max()))
    assert float((dep.cpu() - depth).abs().max()) <= dtol
    for k, got, ref in [("means2D", m2d.grad, m2.grad)] + [(k, d[k].grad, ins[k].grad) for k in d]:
        got = got.cpu()
        assert got.shape == ref.shape, k
        assert torch.isfinite(got).all(), k
        rel = float((got - ref).abs().max() / (ref.abs().max() + 1e-30))
        assert rel <= GRAD_TOL, f"{k}: rel {rel:.3e}"


def test_rasterizer_argument_checks():
    """Same exceptions as the reference wrapper for ambiguous colour / covariance inputs."""
    from fdgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    c = rc.make_case("deg0")
    dev = torch.device("cuda:0")
    S = c["S"]
    gs = GaussianRasterizationSettings(S.image_height, S.image_width, S.tanfovx, S.tanfovy, S.bg.to(dev),
                                       S.scale_modifier, S.viewmatrix.to(dev), S.projmatrix.to(dev), S.sh_degree,
                                       S.campos.to(dev), False, False)
    r = GaussianRasterizer(gs)
    m = c["means3D"].to(dev)
    kw = dict(means3D=m, means2D=torch.zeros_like(m), opacities=c["opacities"].to(dev))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(shs=None, colors_precomp=None, scales=c["scales"].to(dev), rotations=c["rotations"].to(dev), **kw)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(shs=c["shs"].to(dev), colors_precomp=torch.zeros_like(m), scales=c["scales"].to(dev),
          rotations=c["rotations"].to(dev), **kw)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(shs=c["shs"].to(dev), scales=None, rotations=None, cov3D_precomp=None, **kw)


def test_fused_adam_matches_torch():
    """fdgs_adam_step against torch.optim.Adam with the reference's settings (eps 1e-15, per-group lr),
    odd sizes and an unaligned gradient view included."""
    from fdgs.adam import FusedAdam
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    shapes = [(1000, 3), (4097,), (128, 128), (3,), (16, 64, 64), (1,)]
    lrs = [1.6e-4, 2.5e-3, 5e-2, 1e-3, 1.6e-3, 5e-3]
    ref_p = [torch.randn(s, generator=gen).to(dev).requires_grad_(True) for s in shapes]
    our_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref_p, lrs)], lr=0.0, eps=1e-15)
    our = FusedAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(our_p, lrs))],
                    lr=0.0, eps=1e-15)
    for step in range(4):
        flat = torch.randn(sum(p.numel() for p in ref_p) + 1, generator=gen).to(dev)
        off = 1                                      # odd offset: gradient views not 16-byte aligned
        for a, b in zip(ref_p, our_p):
            g = flat[off:off + a.numel()].view(a.shape) * (10.0 ** (step - 2))
            a.grad, b.grad = g.clone(), g
            off += a.numel()
        ref.step()
        our.step()
    torch.cuda.synchronize()
    for a, b in zip(ref_p, our_p):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max()))
    for a, b in zip(ref_p, our_p):
        for key in ("exp_avg", "exp_avg_sq"):      # max-norm relative: single elements cancel to near zero
            x, y = ref.state[a][key], our.state[b][key]
            assert float((x - y).abs().max()) <= 2e-6 * float(x.abs().max()), key


def test_fused_adam_counts_steps_per_tensor_and_round_trips_its_state():
    """A tensor that gets its first gradient after k steps (the deformation field after the coarse stage,
    fdgs_trainer.py:114) starts its bias correction at step 1, as torch.optim.Adam's per-parameter
    state[p]["step"] does; a tensor without a gradient is not stepped.  state_dict() has torch's layout: it
    loads into torch.optim.Adam and back, and training continues identically."""
    from fdgs.adam import FusedAdam
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(12)
    shapes = [(500, 3), (128, 128), (48, 128)]
    lrs = [1.6e-4, 1.6e-4, 1.6e-3]
    ref_p = [torch.randn(s, generator=gen).to(dev).requires_grad_(True) for s in shapes]
    our_p = [p.detach().clone().requires_grad_(True) for p in ref_p]

    def make(cls, ps):
        return cls([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))],
                   lr=0.0, eps=1e-15)
    ref, our = make(torch.optim.Adam, ref_p), make(FusedAdam, our_p)

    def one_step(ro, oo, rp, op, step, late_from=5):
        for i, (a, b) in enumerate(zip(rp, op)):
            if i > 0 and step < late_from:          # tensors 1 and 2 join at step `late_from`
                a.grad = b.grad = None
                continue
            g = torch.randn(a.shape, generator=gen).to(dev)
            a.grad, b.grad = g.clone(), g
        ro.step()
        oo.step()

    for step in range(8):
        one_step(ref, our, ref_p, our_p, step)
    torch.cuda.synchronize()
    assert our.state[our_p[0]]["step"] == 8 and our.state[our_p[1]]["step"] == 3
    for a, b in zip(ref_p, our_p):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max()))
    # round trip through torch's own loader, in both directions
    ref2_p = [p.detach().clone().requires_grad_(True) for p in our_p]
    our2_p = [p.detach().clone().requires_grad_(True) for p in our_p]
    ref2, our2 = make(torch.optim.Adam, ref2_p), make(FusedAdam, our2_p)
    ref2.load_state_dict(our.state_dict())
    our2.load_state_dict(ref.state_dict())
    assert our2.state[our2_p[1]]["step"] == 3 and float(ref2.state[ref2_p[1]]["step"]) == 3.0
    for step in range(8, 11):
        one_step(ref2, our2, ref2_p, our2_p, step)
    for a, b in zip(ref2_p, our2_p):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))


def test_fused_adam_reads_column_slice_gradients_in_place():
    """The gradients of _features_dc [N,1,3] and _features_rest [N,15,3] are the column slices [:, :1] and [:, 1:]
    of one dL/dSH [N,16,3] array: FusedAdam reads them where they are (fdgs_adam_args.grad_row_len / _stride)
    and must give what torch.optim.Adam gives on contiguous copies.  N is odd-sized so that rows straddle the
    kernel's four-element groups and its last workgroup is partial."""
    from fdgs.adam import FusedAdam
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(13)
    n = 10007
    dc0, rest0 = torch.randn(n, 1, 3, generator=gen).to(dev), torch.randn(n, 15, 3, generator=gen).to(dev)
    ref_p = [dc0.clone().requires_grad_(True), rest0.clone().requires_grad_(True)]
    our_p = [dc0.clone().requires_grad_(True), rest0.clone().requires_grad_(True)]
    lrs = [2.5e-3, 2.5e-3 / 20]
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref_p, lrs)], lr=0.0, eps=1e-15)
    our = FusedAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(our_p, lrs))],
                    lr=0.0, eps=1e-15)
    for step in range(3):
        g_sh = torch.randn(n, 16, 3, generator=gen).to(dev)
        ref_p[0].grad, ref_p[1].grad = g_sh[:, :1, :].contiguous(), g_sh[:, 1:, :].contiguous()
        our_p[0].grad, our_p[1].grad = g_sh[:, :1, :], g_sh[:, 1:, :]
        assert not our_p[1].grad.is_contiguous()
        keep = g_sh.clone()
        ref.step()
        our.step()
        assert torch.equal(g_sh, keep)                      # read in place, never written
    torch.cuda.synchronize()
    for a, b in zip(ref_p, our_p):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max()))
        for key in ("exp_avg", "exp_avg_sq"):
            x, y = ref.state[a][key], our.state[b][key]
            assert float((x - y).abs().max()) <= 2e-6 * float(x.abs().max()), key


def test_l1_loss_and_densification_stats_match_torch():
    """fdgs.losses: scale * mean|a - b| with its gradient (src/model/losses.py:7-8) and the statistics update of
    src/runner/fdgs_trainer.py:181-182 against the PyTorch expressions; sizes that exercise the 16-byte path,
    the tail, and a non-contiguous input."""
    from fdgs import losses
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    for shape in ((3, 1014, 1352), (3, 37, 53), (1, 5)):
        a = torch.randn(shape, generator=gen).to(dev).requires_grad_(True)
        b = torch.randn(shape, generator=gen).to(dev)
        b.view(-1)[::7] = a.detach().view(-1)[::7]          # exact ties: sign 0
        a2 = a.detach().clone().requires_grad_(True)
        got = losses.l1_loss(a, b, 0.125) * 3.0
        ref = torch.abs(a2 - b).mean() * 0.125 * 3.0
        got.backward(); ref.backward()
        assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref))
        assert torch.equal(a.grad == 0, a2.grad == 0)
        assert float((a.grad - a2.grad).abs().max()) <= 1e-6 * float(a2.grad.abs().max())
    at = torch.randn(64, 48, generator=gen).to(dev).t()     # non-contiguous view
    assert abs(float(losses.l1_loss(at, torch.zeros_like(at))) - float(at.abs().mean())) <= 1e-6

    n = 10007
    radii = torch.randint(-1, 40, (n,), generator=gen).to(dev).to(torch.int32)
    vgrad = torch.randn(n, 3, generator=gen).to(dev)
    acc = torch.rand(n, 1, generator=gen).to(dev); den = torch.rand(n, 1, generator=gen).to(dev).round()
    mx = (torch.rand(n, generator=gen) * 30).to(dev)
    acc_r, den_r, mx_r = acc.clone(), den.clone(), mx.clone()
    vis = radii > 0
    mx_r[vis] = torch.max(mx_r[vis], radii[vis].float())
    acc_r[vis] += torch.norm(vgrad[vis, :2], dim=-1, keepdim=True)
    den_r[vis] += 1
    losses.add_densification_stats(radii, vgrad, acc, den, mx)
    assert torch.equal(mx, mx_r) and torch.equal(den, den_r)
    assert float((acc - acc_r).abs().max()) <= 1e-6


def test_distCUDA2_matches_all_pairs_search():
    """simple_knn._C.distCUDA2 (src/model/gaussian_model.py:57) against an all-pairs top-3 in float64; with
    duplicate points (distance 0 counts, the point itself does not) and a size that is no multiple of the tile."""
    from simple_knn._C import distCUDA2
    gen = torch.Generator().manual_seed(21)
    pts = torch.randn(3001, 3, generator=gen)
    pts[100] = pts[7]; pts[101] = pts[7]                  # a triple
    got = distCUDA2(pts.cuda()).cpu()
    d = torch.cdist(pts.double(), pts.double()) ** 2
    d.fill_diagonal_(float("inf"))
    ref = d.topk(3, dim=1, largest=False).values.mean(dim=1)
    assert got.shape == (3001,)
    assert float((got.double() - ref).abs().max() / ref.max()) <= 1e-6
    assert float(got[7]) <= 1e-6 * float(ref.max()) + float(ref[7]) * 1.001 and float(ref[7]) > 0   # two zeros + one real


def test_blend_results_do_not_depend_on_the_tile_order():
    """fdgs_raster_blend_fwd with the tiles taken longest list first, in index order (NULL) and in reverse: the same
    image, depth, transmittance and contributor counts bit for bit (a tile's pixels are computed by one workgroup
    whichever it is); an array that is not a permutation is refused tile by tile instead of written through."""
    from fdgs import _lib, rasterizer as fr
    from fdgs._lib import ptr
    c = rc.make_case("medium")
    dev = torch.device("cuda:0")
    S = _to_dev(c["S"], dev)
    st = fr.rasterize_forward(c["means3D"].to(dev), c["shs"].to(dev), c["opacities"].reshape(-1).to(dev).contiguous(),
                              c["scales"].to(dev), c["rotations"].to(dev), S)
    H, W = int(S.image_height), int(S.image_width)
    bg = st.cam[3]
    tiles = st.ranges.shape[0]
    ref = (st.color.clone(), st.depth.clone(), st.final_T.clone(), st.n_contrib.clone())
    reverse = torch.arange(tiles - 1, -1, -1, device=dev, dtype=torch.int32)
    bogus = torch.full((tiles,), tiles + 5, device=dev, dtype=torch.int32)
    for order in (None, reverse, bogus):
        out = [torch.full_like(t, 7) for t in ref]
        _lib.call("fdgs_raster_blend_fwd", _lib.BlendFwdArgs(W, H, ptr(st.ranges), ptr(st.vals_sorted), ptr(st.geom),
                                                             ptr(bg), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]),
                                                             ptr(order)))
        torch.cuda.synchronize()
        for got, want in zip(out, ref):
            if order is bogus:
                assert bool((got == 7).all())           # nothing written
            else:
                assert torch.equal(got, want)
