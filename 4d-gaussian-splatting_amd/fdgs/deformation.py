"""Drop-in for the reference's `DeformationFields` (src/model/deformation.py:180-250).

Same constructor (`DeformationFields(kwargs)` with the keys of
configs/train/dynerf.yaml:85-111), same `forward(point, scales, rotations,
opacity, shs, time)` -> 5-tuple, same `state_dict()` keys and shapes (so a
reference-trained `deformation.pth` loads), same helpers the trainer and the
scene use: `get_mlp_parameters()` / `get_grid_parameters()` (split on "grid" in
the parameter name, deformation.py:165-177), `deformation_net.set_aabb`,
`deformation_net.grid.grids`, `get_aabb`.

The module only holds parameters; forward and backward run in libfdgs.so
(fdgs_deform_fwd / fdgs_deform_bwd).  The positional encodings of
deformation.py:235-237 are not evaluated: every consumer slices them back to
their identity part (deformation.py:86,130,137,147), so outputs and gradients are
unchanged.
"""
import ctypes as C
import itertools

import os

import torch
import torch.nn as nn

from . import _lib, adam
from ._lib import lib, ptr, check, stream_ptr

PLANE_AXES = list(itertools.combinations(range(4), 2))
HEADS = ("pos", "scales", "rotations", "opacity", "shs")
HEAD_DIMS = (3, 3, 4, 1, 48)


class HexPlaneField(nn.Module):
    """Parameter container with the layout of src/model/hexplane.py:128-180."""

    def __init__(self, bounds, planeconfig, multires):
        super().__init__()
        aabb = torch.tensor([[bounds, bounds, bounds], [-bounds, -bounds, -bounds]], dtype=torch.float32)
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        self.grid_config = [planeconfig]
        self.multiscale_res_multipliers = list(multires)
        self.concat_features = True
        self.grids = nn.ModuleList()
        self.feat_dim = 0
        self.resolutions = []
        out_dim = planeconfig["output_coordinate_dim"]
        for m in self.multiscale_res_multipliers:
            reso = [r * m for r in planeconfig["resolution"][:3]] + list(planeconfig["resolution"][3:])
            self.resolutions.append(reso)
            gp = nn.ParameterList()
            for (a, b) in PLANE_AXES:
                g = nn.Parameter(torch.empty(1, out_dim, reso[b], reso[a]))
                if b == 3:                       # planes with a time axis start at one (hexplane.py:80-81)
                    nn.init.ones_(g)
                else:
                    nn.init.uniform_(g, a=0.1, b=0.5)
                gp.append(g)
            self.grids.append(gp)
            self.feat_dim += out_dim

    @property
    def get_aabb(self):
        return self.aabb[0], self.aabb[1]

    def set_aabb(self, xyz_max, xyz_min):
        """Row 0 = xyz_max, row 1 = xyz_min (src/model/hexplane.py:174-180)."""
        dev = self.aabb.device
        aabb = torch.tensor([list(map(float, xyz_max)), list(map(float, xyz_min))], dtype=torch.float32, device=dev)
        self.aabb = nn.Parameter(aabb, requires_grad=False)

    @torch.no_grad()
    def get_density(self, pts, timestamps=None):
        """Multi-level plane features [N, 32] at (pts, timestamps) (src/model/hexplane.py:182-196).
        Forward only: gradients flow through DeformationFields.forward, which fuses this step."""
        pts = pts.reshape(-1, pts.shape[-1]).detach().to(torch.float32).contiguous()
        n = pts.shape[0]
        _lib.require_gpu(pts, self.aabb)
        if isinstance(timestamps, (float, int)):
            time_scalar, time_c = float(timestamps), None
        else:
            time_scalar, time_c = 0.0, timestamps.detach().to(torch.float32).reshape(-1).contiguous()
            if time_c.numel() != n:
                raise ValueError("timestamps must have one entry per point")
        P = _lib.DeformParams()
        P.num_levels = len(self.grids)
        for l, reso in enumerate(self.resolutions):
            for k in range(4):
                P.res[l][k] = int(reso[k])
        P.aabb = ptr(self.aabb.detach())
        for l, gp in enumerate(self.grids):
            for ci in range(6):
                P.planes[l][ci] = ptr(gp[ci])
        out = torch.empty(n, self.feat_dim, device=pts.device, dtype=torch.float32)
        ws_bytes = lib.fdgs_hexplane_features_workspace_bytes(C.byref(P), n)
        ws = torch.empty(max(ws_bytes, 16), device=pts.device, dtype=torch.uint8)
        a = _lib.HexFeaturesArgs(n, P, ptr(pts), ptr(time_c), time_scalar, ptr(out), ptr(ws), ws_bytes)
        _lib.call("fdgs_hexplane_features", a)
        return out

    def forward(self, pts, timestamps=None):
        return self.get_density(pts, timestamps)


def _head(width, out_dim):
    return nn.Sequential(nn.ReLU(), nn.Linear(width, width), nn.ReLU(), nn.Linear(width, out_dim))


class DeformationNet(nn.Module):
    def __init__(self, W, grid_pe, bounds, kplanes_config, multires):
        super().__init__()
        self.W = W
        self.grid_pe = grid_pe
        self.grid = HexPlaneField(bounds, kplanes_config, multires)
        self.feature_out = nn.Sequential(nn.Linear(self.grid.feat_dim, W))     # defor_depth - 1 == -1 extra layers
        self.pos_deform = _head(W, 3)
        self.scales_deform = _head(W, 3)
        self.rotations_deform = _head(W, 4)
        self.opacity_deform = _head(W, 1)
        self.shs_deform = _head(W, 48)

    @property
    def get_aabb(self):
        return self.grid.get_aabb

    def set_aabb(self, xyz_max, xyz_min):
        self.grid.set_aabb(xyz_max, xyz_min)

    def get_mlp_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" not in n]

    def get_grid_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" in n]


# how the library forms the fp32 matrix products: "f32" (fp32 MFMA) or "split" (exact three-way bf16 split of
# both operands, six bf16 MFMAs per product, fp32 accumulation); see FDGS_MATMUL_* in include/fdgs.h
MATMUL = {"f32": _lib.MATMUL_F32, "split": _lib.MATMUL_SPLIT_BF16}[os.environ.get("FDGS_DEFORM_MATMUL", "split")]


def _trainable(net):
    """The 34 trainable tensors in the order of the C structs: 12 planes, w0, b0, then (w1, b1, w2, b2) per
    head.  The list is built once per module (Parameter objects are stable; their storage may move)."""
    ts = net.__dict__.get("_fdgs_trainable")
    if ts is None:
        ts = []
        for gp in net.grid.grids:
            ts += list(gp)
        ts += [net.feature_out[0].weight, net.feature_out[0].bias]
        for name in HEADS:
            seq = getattr(net, name + "_deform")
            ts += [seq[1].weight, seq[1].bias, seq[3].weight, seq[3].bias]
        net.__dict__["_fdgs_trainable"] = ts
        # flat gradient buffer: every tensor whose size is a multiple of 16 floats first (each then starts on a
        # 64-byte boundary, so consumers can use 16-byte accesses), the five small last-layer biases at the end
        order = sorted(range(len(ts)), key=lambda i: (ts[i].numel() % 16 != 0, i))
        net.__dict__["_fdgs_grad_order"] = order
        net.__dict__["_fdgs_grad_sizes"] = [ts[i].numel() for i in order]
        net.__dict__["_fdgs_grad_shapes"] = [tuple(ts[i].shape) for i in order]
    return ts


def _set_pointers(P, tensors, n_levels):
    """Writes 34 device pointers (order of _trainable) into a DeformParams / DeformParamGrads struct."""
    k = 0
    for l in range(n_levels):
        row = P.planes[l]
        for ci in range(6):
            row[ci] = ptr(tensors[k]); k += 1
    P.w0 = ptr(tensors[k]); P.b0 = ptr(tensors[k + 1]); k += 2
    for h in range(5):
        P.w1[h] = ptr(tensors[k]); P.b1[h] = ptr(tensors[k + 1])
        P.w2[h] = ptr(tensors[k + 2]); P.b2[h] = ptr(tensors[k + 3])
        k += 4


def _fill_params(net, grads=None, tensors=None):
    """fdgs_deform_params from the module's tensors (or from `tensors`, the same 34 in the same order: the
    zero-padded set of a narrower network), or the gradient twin from `grads` (same order)."""
    grid = net.grid
    n_levels = len(grid.resolutions)
    if grads is not None:
        G = _lib.DeformParamGrads()
        _set_pointers(G, grads, n_levels)
        return G
    P = _lib.DeformParams()
    P.num_levels = n_levels
    for l, reso in enumerate(grid.resolutions):
        for k in range(4):
            P.res[l][k] = int(reso[k])
    P.aabb = ptr(grid.aabb.detach())
    P.matmul = MATMUL
    _set_pointers(P, _trainable(net) if tensors is None else tensors, n_levels)
    return P


KERNEL_WIDTH = 128          # hidden width the kernels are built for


def _kernel_parameters(net):
    """The 34 tensors as the kernels take them.  A network narrower than 128 (configs/train/dnerf.yaml:83 has
    net_width 64; create_net, src/model/deformation.py:62-81, builds any width) runs as the 128-wide network whose
    extra hidden units have zero weights and zero biases: those units are exactly zero before and after every ReLU,
    the products they enter are exact zeros, and every sum only gains zeros - the outputs are those of the narrow
    network, and so are the gradients of its own entries (the padding's gradients are dropped by the slicing that
    autograd does for F.pad).  Costs the flops of the wide network."""
    ts = _trainable(net)
    pad = KERNEL_WIDTH - net.W
    if pad == 0:
        return ts
    F = torch.nn.functional
    out = list(ts[:12])
    out += [F.pad(ts[12], (0, 0, 0, pad)), F.pad(ts[13], (0, pad))]                 # W0 [W,32], b0 [W]
    for h in range(5):
        w1, b1, w2, b2 = ts[14 + 4 * h:18 + 4 * h]
        out += [F.pad(w1, (0, pad, 0, pad)), F.pad(b1, (0, pad)), F.pad(w2, (0, pad)), b2]
    return out


# Data-parallel hook: the gradient of the SH coefficients leaves the deformation field unchanged (shs_final =
# shs + delta), so it is final before this module's backward kernels run.  When set, the callable receives that
# tensor (a copy: the kernels still need the rank's own) right before the kernels are enqueued and may start an
# in-place all-reduce on it; the handle it returns
# (or None) is waited for (stream order only) after they are enqueued: 71 % of the gradient bytes then cross
# the links beside the deformation backward instead of after it.
RESIDUAL_GRAD_HOOK = None


class SplitSH:
    """The SH coefficients as the two tensors GaussianModel keeps (src/model/gaussian_model.py:64-65, _features_dc
    [N,1,3] and _features_rest [N,15,3]) in place of their concatenation get_features (:86-89): accepted as `shs` by
    DeformationFields.forward under torch.no_grad(), where the [N,16,3] copy (58 MB at 300k Gaussians, per frame)
    would only be read once by the deformation kernel."""
    __slots__ = ("dc", "rest")

    def __init__(self, dc, rest):
        self.dc, self.rest = dc, rest


class _DeformFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, grad_mode, xyz, scales, rotations, opacity, shs, time, *params):
        sh_rest = None
        if isinstance(shs, SplitSH):
            if grad_mode:
                raise ValueError("SplitSH is an inference input: pass get_features where gradients are needed")
            shs, sh_rest = shs.dc, shs.rest
        _lib.require_gpu(xyz, scales, rotations, opacity, shs, sh_rest, *params)
        N = xyz.shape[0]
        dev = xyz.device
        def c(t):
            t = t.detach()
            return t if (t.dtype is torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()
        xyz_c, sc_c, ro_c, op_c, sh_c = c(xyz), c(scales), c(rotations), c(opacity), c(shs)
        rest_c = c(sh_rest) if sh_rest is not None else None
        time_scalar = 0.0
        if isinstance(time, (float, int)):
            time_scalar, time_c = float(time), None
        else:
            time_c = c(time).reshape(-1)
            if time_c.numel() != N:
                raise ValueError("time must have one entry per point")
        outs = [torch.empty_like(t) for t in (xyz_c, sc_c, ro_c, op_c)]
        outs.append(torch.empty_like(sh_c) if rest_c is None else
                    torch.empty(N, 1 + rest_c.shape[1], 3, device=dev, dtype=torch.float32))
        # needs_input_grad is also set under torch.no_grad(); the caller's grad mode decides whether
        # the activations are saved (0.92 GB at 300k points)
        need_grad = grad_mode and any(ctx.needs_input_grad)
        split_saves = need_grad and MATMUL == _lib.MATMUL_SPLIT_BF16
        # fp32-MFMA mode keeps hidden [N,128]; split mode keeps relu(hidden) twice as operand pieces (768 B/point
        # each: lane = point for the dW2 recompute, points-as-K for the dW1 products) and the ReLU bit masks
        # (96 B/point); relu(h1) is recomputed
        hidden = torch.empty(N, KERNEL_WIDTH, device=dev, dtype=torch.float32) if (need_grad and not split_saves) else None
        act = torch.empty((N + 31) // 32 * 32, 192, device=dev, dtype=torch.float32) if split_saves else None
        yhid = torch.empty((N + 15) // 16 * 16 + 16, 192, device=dev, dtype=torch.float32) if split_saves else None
        mask = torch.empty(N, 12, device=dev, dtype=torch.int64) if split_saves else None
        yfeat = torch.empty((N + 15) // 16 * 16 + 16, 48, device=dev, dtype=torch.float32) if split_saves else None
        kparams = [p.detach() if p.is_contiguous() else p.detach().contiguous() for p in params]
        P = _fill_params(net, tensors=kparams)
        ws_bytes = lib.fdgs_deform_fwd_workspace_bytes(C.byref(P), N)
        ws = torch.empty(max(ws_bytes, 16), device=dev, dtype=torch.uint8)
        a = _lib.DeformFwdArgs(N, P, ptr(xyz_c), ptr(sc_c), ptr(ro_c), ptr(op_c), ptr(sh_c), ptr(time_c), time_scalar,
                               ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), ptr(outs[4]), ptr(hidden),
                               ptr(act), ptr(mask), ptr(yhid), ptr(yfeat), ptr(ws), ws_bytes, ptr(rest_c))
        _lib.call("fdgs_deform_fwd", a)
        ctx.net = net
        ctx.time_scalar = time_scalar
        ctx.param_versions = [p._version for p in _trainable(net)]
        ctx.param_generation = adam.GENERATION[0]       # FusedAdam writes through raw pointers: no version bump
        ctx.matmul = MATMUL
        ctx.save_for_backward(xyz_c, time_c, hidden, ws, act, mask, yhid, yfeat, *kparams)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_xyz, g_sc, g_ro, g_op, g_sh):
        net = ctx.net
        xyz_c, time_c, hidden, fwd_ws, act, mask, yhid, yfeat = ctx.saved_tensors[:8]
        kparams = list(ctx.saved_tensors[8:])            # the tensors the forward ran on (padded for a narrow network)
        N = xyz_c.shape[0]
        dev = xyz_c.device
        def c(t, shape):
            if t is None:
                return torch.zeros(shape, device=dev, dtype=torch.float32)
            return t if (t.dtype is torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()
        g_xyz = c(g_xyz, (N, 3)); g_sc = c(g_sc, (N, 3)); g_ro = c(g_ro, (N, 4))
        g_op = c(g_op, (N, 1)); g_sh = c(g_sh, (N, 16, 3))
        params = _trainable(net)
        # one zero-filled flat buffer, viewed per parameter (the library accumulates into it); every tensor whose size
        # is a multiple of 16 floats first, so that each starts on a 64-byte boundary
        order = sorted(range(len(kparams)), key=lambda i: (kparams[i].numel() % 16 != 0, i))
        sizes = [kparams[i].numel() for i in order]
        flat = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
        grads = [None] * len(kparams)
        for i, piece in zip(order, flat.split_with_sizes(sizes)):
            grads[i] = piece.view(kparams[i].shape)
        P = _fill_params(net, tensors=kparams)
        P.matmul = ctx.matmul
        G = _fill_params(net, grads)
        d_xyz = torch.empty_like(xyz_c)
        # the forward's channel-last plane copies are still valid if no parameter was written since
        # (the twelve planes are always the module's own tensors, also for a padded network)
        unchanged = (ctx.param_generation == adam.GENERATION[0]
                     and all(p._version == v for p, v in zip(params[:12], ctx.param_versions[:12])))
        planes_cl = ptr(fwd_ws) if unchanged else None
        ws_bytes = lib.fdgs_deform_bwd_workspace_bytes(C.byref(P), N)
        ws = torch.empty(max(ws_bytes, 16), device=dev, dtype=torch.uint8)
        a = _lib.DeformBwdArgs(N, P, ptr(xyz_c), ptr(time_c), ctx.time_scalar, ptr(hidden), ptr(act), ptr(mask), ptr(yhid), ptr(yfeat), planes_cl, ptr(g_xyz), ptr(g_sc), ptr(g_ro),
                               ptr(g_op), ptr(g_sh), ptr(d_xyz), G, ptr(ws), ws_bytes, *_lib.side_resources())
        g_sh_out, pending = g_sh, None
        if RESIDUAL_GRAD_HOOK is not None:
            g_sh_out = g_sh.clone()          # the kernels below still read this rank's own cotangent
            pending = RESIDUAL_GRAD_HOOK(g_sh_out)
        _lib.call("fdgs_deform_bwd", a)
        if pending is not None:
            pending.wait()
        # net, grad_mode, xyz, scales, rotations, opacity, shs, time, *params
        return (None, None, d_xyz, g_sc, g_ro, g_op, g_sh_out, None) + tuple(grads)


class DeformationFields(nn.Module):
    def __init__(self, kwargs):
        super().__init__()
        self.net_width = kwargs["net_width"]
        self.timebase_pe = kwargs["timebase_pe"]
        self.defor_depth = kwargs["defor_depth"]
        self.posebase_pe = kwargs["posebase_pe"]
        self.scale_rotation_pe = kwargs["scale_rotation_pe"]
        self.opacity_pe = kwargs["opacity_pe"]
        self.grid_pe = kwargs["grid_pe"]
        self.no_grid = kwargs["no_grid"]
        self.bounds = kwargs["bounds"]
        self.kplanes_config = kwargs["kplanes_config"]
        self.multires = kwargs["multires"]
        self.empty_voxel = kwargs["empty_voxel"]
        self.static_mlp = kwargs["static_mlp"]
        if (not (1 <= self.net_width <= KERNEL_WIDTH) or self.defor_depth > 1 or self.grid_pe != 0 or self.empty_voxel
                or self.static_mlp or len(self.multires) != 2 or self.kplanes_config["output_coordinate_dim"] != 16
                or self.kplanes_config["grid_dimensions"] != 2 or self.kplanes_config["input_coordinate_dim"] != 4):
            raise NotImplementedError("libfdgs implements the shipped deformation configurations "
                                      "(configs/train/dynerf.yaml:85-111, dnerf.yaml:83-107): width up to 128, "
                                      "defor_depth 0/1, two levels, 16 channels, no grid PE / static MLP / empty voxel")
        self.deformation_net = DeformationNet(self.net_width, self.grid_pe, self.bounds, self.kplanes_config,
                                              self.multires)
        self.register_buffer("time_poc", torch.FloatTensor([(2 ** i) for i in range(self.timebase_pe)]))
        self.register_buffer("pos_poc", torch.FloatTensor([(2 ** i) for i in range(self.posebase_pe)]))
        self.register_buffer("rotation_scaling_poc",
                             torch.FloatTensor([(2 ** i) for i in range(self.scale_rotation_pe)]))
        self.register_buffer("opacity_poc", torch.FloatTensor([(2 ** i) for i in range(self.opacity_pe)]))
        for m in self.modules():                  # deformation.py:252-258: Xavier weights, default bias
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=1)

    @property
    def get_aabb(self):
        return self.deformation_net.get_aabb

    def forward(self, point, scales=None, rotations=None, opacity=None, shs=None, time=None):
        return self.forward_dynamic(point, scales, rotations, opacity, shs, time)

    def forward_dynamic(self, point, scales=None, rotations=None, opacity=None, shs=None, time=None):
        net = self.deformation_net
        return _DeformFunction.apply(net, torch.is_grad_enabled(), point, scales, rotations, opacity, shs, time,
                                     *_kernel_parameters(net))

    def get_mlp_parameters(self):
        return self.deformation_net.get_mlp_parameters()

    def get_grid_parameters(self):
        return self.deformation_net.get_grid_parameters()
