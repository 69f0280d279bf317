"""GPU parity of the HIP deformation field (through the C ABI) against
oracle/deform_ref.py and against the golden vectors captured from the
reference's own modules.  Outputs <= 2e-5 abs, gradients <= 1e-4 relative."""
import os

import numpy as np
import pytest
import torch

import deform_cases as dc
from oracle import deform_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
OUT_TOL = 2e-5
GRAD_TOL = 1e-4


def _module(case, dev):
    from fdgs.deformation import DeformationFields
    net = DeformationFields(dc.net_kwargs(case))
    sd = dc.make_state(case)
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())
    net.load_state_dict(sd)
    return net.to(dev), sd


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(params=["split", "f32"])
def matmul(request, monkeypatch):
    """Both ways the library forms the fp32 matrix products (FDGS_MATMUL_* in include/fdgs.h)."""
    from fdgs import _lib, deformation
    monkeypatch.setattr(deformation, "MATMUL", {"f32": _lib.MATMUL_F32, "split": _lib.MATMUL_SPLIT_BF16}[request.param])
    return request.param


@pytest.mark.parametrize("case", list(dc.CASES))
def test_forward_backward_vs_golden_and_oracle(case, matmul):
    dev = torch.device("cuda:0")
    net, sd = _module(case, dev)
    gold = np.load(os.path.join(GOLD, f"deform_{case}.npz"))
    ins = [t.to(dev).requires_grad_(i < 5) for i, t in enumerate(dc.make_inputs(case))]
    outs = net(*ins)
    names = ("pts", "scales", "rotations", "opacity", "shs")
    for name, t in zip(names, outs):
        ref = torch.from_numpy(gold["out_" + name])
        assert t.shape == ref.shape
        assert float((t.detach().cpu() - ref).abs().max()) <= OUT_TOL, name
    gouts = [g.to(dev) for g in dc.make_grad_outputs(case)]
    params = dict(net.named_parameters())
    tkeys = dc.trainable_keys(sd)
    grads = torch.autograd.grad(outs, ins[:5] + [params[k] for k in tkeys], gouts)
    torch.cuda.synchronize()
    for name, g in zip(("xyz", "scales", "rotations", "opacity", "shs"), grads[:5]):
        assert _rel(g.cpu(), torch.from_numpy(gold["gin_" + name])) <= GRAD_TOL, name
    for k, g in zip(tkeys, grads[5:]):
        assert _rel(g.cpu(), torch.from_numpy(gold["gp_" + k])) <= GRAD_TOL, k


def test_scalar_time_single_view_and_no_grad(matmul):
    """render() passes one time per view repeated N times (gaussian_renderer/__init__.py:52);
    also runs under torch.no_grad and with N not a multiple of the tile."""
    dev = torch.device("cuda:0")
    net, sd = _module("tiny", dev)
    xyz, sc, ro, op, sh, _ = dc.make_inputs("tiny")
    n = 333
    t = torch.full((n, 1), 0.37)
    ref = deform_ref.deform_forward(sd, xyz[:n], sc[:n], ro[:n], op[:n], sh[:n], t)
    with torch.no_grad():
        got = net(xyz[:n].to(dev), sc[:n].to(dev), ro[:n].to(dev), op[:n].to(dev), sh[:n].to(dev), t.to(dev))
    for a, b in zip(got, ref):
        assert float((a.cpu() - b).abs().max()) <= OUT_TOL
    one = net(xyz[:1].to(dev), sc[:1].to(dev), ro[:1].to(dev), op[:1].to(dev), sh[:1].to(dev), t[:1].to(dev))
    assert [tuple(o.shape) for o in one] == [(1, 3), (1, 3), (1, 4), (1, 1), (1, 16, 3)]


def test_split_products_are_fp32_accurate():
    """The split-bf16 matrix products (six exact bf16 x bf16 terms, fp32 accumulation) must be as close
    to a float64 evaluation as the fp32-MFMA path is: the forward has no discontinuity inside the box,
    so the two error levels are directly comparable."""
    from fdgs import _lib, deformation
    dev = torch.device("cuda:0")
    net, sd = _module("dynerf", dev)
    xyz, sc, ro, op, sh, tm = dc.make_inputs("dynerf")
    sd64 = {k: v.double() for k, v in sd.items()}
    ref = deform_ref.deform_forward(sd64, xyz.double(), sc.double(), ro.double(), op.double(), sh.double(), tm.double())
    errs = {}
    old = deformation.MATMUL
    try:
        for name, mode in (("f32", _lib.MATMUL_F32), ("split", _lib.MATMUL_SPLIT_BF16)):
            deformation.MATMUL = mode
            with torch.no_grad():
                got = net(xyz.to(dev), sc.to(dev), ro.to(dev), op.to(dev), sh.to(dev), tm.to(dev))
            errs[name] = max(float((a.cpu().double() - b).abs().max()) for a, b in zip(got, ref))
    finally:
        deformation.MATMUL = old
    assert errs["f32"] <= 2e-6 and errs["split"] <= 2e-6, errs
    assert errs["split"] <= 2.0 * errs["f32"] + 2e-7, errs


@pytest.mark.parametrize("case,n", [("tiny", 333), ("dynerf", None)])
def test_hexplane_features_alone(case, n):
    """HexPlaneField.forward / get_density (src/model/hexplane.py:182-204): the [N,32] plane features,
    per-point times and one scalar time, N not a multiple of the kernel's point group."""
    dev = torch.device("cuda:0")
    net, sd = _module(case, dev)
    xyz, _, _, _, _, tm = dc.make_inputs(case)
    if n is not None:
        xyz, tm = xyz[:n], tm[:n]
    grid = net.deformation_net.grid
    ref = deform_ref.hexplane_features(sd, xyz, tm)
    got = grid(xyz.to(dev), tm.to(dev))
    assert got.shape == ref.shape == (xyz.shape[0], 32)
    assert float((got.cpu() - ref).abs().max()) <= 1e-6
    ref1 = deform_ref.hexplane_features(sd, xyz, torch.full_like(tm, 0.25))
    got1 = grid.get_density(xyz.to(dev), 0.25)
    assert float((got1.cpu() - ref1).abs().max()) <= 1e-6
    assert grid(xyz[:0].to(dev), tm[:0].to(dev)).shape == (0, 32)


@pytest.mark.parametrize("n", [700, 1, 33])
@pytest.mark.parametrize("hex_bwd", ["bricks", "flat", "3"])
def test_scalar_time_backward_matches_oracle(hex_bwd, n, monkeypatch):
    """One time stamp for the whole call (how render() calls the field): the HexPlane backward then keeps the
    time-plane rows in LDS.  Checked against autograd through the oracle, for the brick kernel (default and the
    finest brick grid) and for the one-atomic-per-corner kernel, with some points outside the box (clipped to its faces)."""
    if hex_bwd != "bricks":
        monkeypatch.setenv("FDGS_HEX_BWD", hex_bwd)
    dev = torch.device("cuda:0")
    net, sd = _module("tiny", dev)
    xyz, sc, ro, op, sh, _ = dc.make_inputs("tiny")
    xyz, sc, ro, op, sh = (t[:n].clone() for t in (xyz, sc, ro, op, sh))
    aabb = sd["deformation_net.grid.aabb"]
    if n >= 48:
        xyz[16:32] = aabb[0] + (aabb[0] - aabb[1]) * 0.25  # outside: clipped to the border
        xyz[32:48] = aabb[1] - (aabb[0] - aabb[1]) * 0.25
    tval = 0.37
    gouts = [g[:n] for g in dc.make_grad_outputs("tiny")]
    # oracle: autograd through the CPU restatement
    sdr = {k: v.clone().requires_grad_(v.is_floating_point() and "aabb" not in k) for k, v in sd.items()}
    xr = xyz.clone().requires_grad_(True)
    ref = deform_ref.deform_forward(sdr, xr, sc, ro, op, sh, torch.full((n, 1), tval))
    tkeys = dc.trainable_keys(sd)
    ref_g = torch.autograd.grad(ref, [xr] + [sdr[k] for k in tkeys], gouts, allow_unused=True)
    xg = xyz.to(dev).requires_grad_(True)
    outs = net(xg, sc.to(dev), ro.to(dev), op.to(dev), sh.to(dev), tval)
    params = dict(net.named_parameters())
    got = torch.autograd.grad(outs, [xg] + [params[k] for k in tkeys], [g.to(dev) for g in gouts])
    torch.cuda.synchronize()
    for name, g, r in zip(["xyz"] + tkeys, got, ref_g):
        r = torch.zeros_like(g.cpu()) if r is None else r
        assert _rel(g.cpu(), r) <= GRAD_TOL, (hex_bwd, name)


@pytest.mark.parametrize("case", ["tiny", "dynerf"])      # the cases tests/golden/reg_*.npz hold
def test_plane_regulation_vs_reference_golden(case):
    """fdgs.losses.compute_regulation (autograd) and add_regulation_grad_ (in place after the backward) against
    the reference's DeformationWrapper.compute_regulation on the golden parameters."""
    from fdgs import losses
    from test_oracle_deform import check_regulation_against_golden, REG_WEIGHTS
    dev = torch.device("cuda:0")
    net, sd = _module(case, dev)
    planes = [p for gp in net.deformation_net.grid.grids for p in gp]
    loss = losses.compute_regulation(net, *REG_WEIGHTS) * 2.0            # a non-unit upstream gradient
    grads = torch.autograd.grad(loss, planes)
    check_regulation_against_golden(case, float(loss) / 2.0, [g.cpu() / 2.0 for g in grads])
    # in-place route: .grad holds something already, the regulariser's gradient is added to it
    for p in planes:
        p.grad = torch.zeros_like(p)
    val = losses.add_regulation_grad_(net, *REG_WEIGHTS)
    once = [p.grad.clone() for p in planes]
    check_regulation_against_golden(case, float(val), [g.cpu() for g in once])
    losses.add_regulation_grad_(net, *REG_WEIGHTS, want_loss=False)          # accumulates: x + x is exact
    assert all(torch.equal(p.grad, 2.0 * g) for p, g in zip(planes, once))


def test_two_backward_calls_on_two_streams_overlap_safely():
    """fdgs_deform_bwd keeps no state of its own: the helper stream and events it forks onto are lent by the caller,
    one triple per calling stream.  Two forward + backward passes enqueued on two streams (they overlap on the
    device: each runs for more than a millisecond, the host enqueues the second within that time) must give the
    results of the same passes run one after the other."""
    import copy
    from fdgs import _lib
    dev = torch.device("cuda:0")
    net_a, _ = _module("dynerf", dev)
    net_b = copy.deepcopy(net_a)
    n = 200000
    gen = torch.Generator().manual_seed(21)

    def inputs(scale):
        lo, hi = -1.5 * scale, 1.5 * scale
        xyz = lo + (hi - lo) * torch.rand(n, 3, generator=gen)
        return [xyz.to(dev).requires_grad_(True)] + [torch.randn(s, generator=gen).to(dev).requires_grad_(True)
                                                     for s in ((n, 3), (n, 4), (n, 1), (n, 16, 3))]
    ins_a, ins_b = inputs(1.0), inputs(0.7)
    cots = [torch.randn(s, generator=gen).to(dev) for s in ((n, 3), (n, 3), (n, 4), (n, 1), (n, 16, 3))]

    def run(net, ins, t):
        outs = net(*ins, t)
        params = [p for p in net.parameters() if p.requires_grad]
        return [o.detach() for o in outs] + list(torch.autograd.grad(outs, ins + params, cots))

    ref_a, ref_b = run(net_a, ins_a, 0.25), run(net_b, ins_b, 0.75)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    before = len(_lib._SIDE)
    with torch.cuda.stream(s1):
        got_a = run(net_a, ins_a, 0.25)
    with torch.cuda.stream(s2):
        got_b = run(net_b, ins_b, 0.75)
    torch.cuda.synchronize()
    assert len(_lib._SIDE) == before + 2          # each calling stream got its own helper stream and events
    for ref, got in ((ref_a, got_a), (ref_b, got_b)):
        for r, g in zip(ref, got):
            assert float((r - g).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-12     # atomics: order noise only


@pytest.mark.parametrize("n", [1, 333, 4097])
def test_forward_reads_the_two_sh_tensors_without_a_concatenation(matmul, n):
    """fdgs_deform_fwd_args.shs_rest: the SH coefficients as GaussianModel keeps them, _features_dc [N,1,3] and
    _features_rest [N,15,3], instead of their concatenation: all five outputs bit-identical."""
    import ctypes as C
    from fdgs import _lib, deformation
    from fdgs._lib import lib, ptr
    dev = torch.device("cuda:0")
    net, _ = _module("dynerf", dev)
    gen = torch.Generator().manual_seed(41)
    xyz = (3.0 * torch.rand(n, 3, generator=gen) - 1.5).to(dev)
    sc, ro, op = (torch.randn(n, k, generator=gen).to(dev) for k in (3, 4, 1))
    f_dc, f_rest = torch.randn(n, 1, 3, generator=gen).to(dev), torch.randn(n, 15, 3, generator=gen).to(dev)
    shs = torch.cat((f_dc, f_rest), dim=1).contiguous()
    P = deformation._fill_params(net.deformation_net)
    ws_bytes = lib.fdgs_deform_fwd_workspace_bytes(C.byref(P), n)
    outs = []
    for dc_arg, rest_arg in ((shs, None), (f_dc, f_rest)):
        ws = torch.empty(max(ws_bytes, 16), device=dev, dtype=torch.uint8)
        o = [torch.empty(n, k, device=dev) for k in (3, 3, 4, 1, 48)]
        a = _lib.DeformFwdArgs(n, P, ptr(xyz), ptr(sc), ptr(ro), ptr(op), ptr(dc_arg), None, 0.37,
                               ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(o[3]), ptr(o[4]), None, None, None, None, None,
                               ptr(ws), ws_bytes, ptr(rest_arg))
        _lib.call("fdgs_deform_fwd", a)
        outs.append(o)
    torch.cuda.synchronize()
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    assert float((outs[0][4] - shs.reshape(n, 48)).abs().max()) > 0       # the head's residual was added
