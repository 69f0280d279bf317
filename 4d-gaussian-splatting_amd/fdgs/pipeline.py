"""One training view as a straight sequence of C-ABI calls on persistent buffers.

`render()` (fdgs/renderer.py) keeps the reference's interface: autograd Functions, fresh tensors per call,
a Python frame per stage.  On a 3 ms iteration that interface costs a tenth of the time in host work while the
GPU waits.  The training loop of the reference (src/runner/fdgs_trainer.py:239-248, :127-140) always runs the same
chain - render, L1 loss, backward - so `ViewPipeline.run` issues exactly that chain: deformation forward,
rasterizer forward, L1 loss (+ PSNR), L1 backward, rasterizer backward, deformation backward, with every buffer
allocated once per (N, image size) and every argument struct built once.  Same kernels, same arithmetic and the
same results as the autograd route (tests/test_pipeline_gpu.py compares them); gradients are returned as tensors
for the caller to attach to the parameters.
"""
import ctypes as C
import math
import os

import torch

from . import _lib, deformation as _deformation, rasterizer as _rasterizer
from ._lib import lib, ptr


def arena_layout(N, deform_numel):
    """The flat gradient array of one set: (regions, marks, total floats).  Order
        deformation field | xyz | means2D | scaling | rotation | opacity | shs        (each padded to 256 bytes)
    marks[name] is a region's first float, marks["end"] the total."""
    def pad(n):
        return (n + 63) // 64 * 64
    regions = (("dflat", (deform_numel,)), ("xyz", (N, 3)), ("means2D", (N, 3)), ("scaling", (N, 3)),
               ("rotation", (N, 4)), ("opacity", (N, 1)), ("shs", (N, 16, 3)))
    marks, off = {}, 0
    for name, shape in regions:
        marks[name] = off
        off += pad(math.prod(shape))
    marks["end"] = off
    return regions, marks, off


def exchange_bounds(marks, fine, early_done):
    """[first, last) of the arena a data-parallel exchange still has to sum over the ranks: everything (fine stage),
    everything but the deformation field (coarse stage: it has no gradient), and only up to the early range
    (means2D .. shs) once that range has travelled by itself."""
    first = marks["dflat"] if fine else marks["xyz"]
    last = marks["means2D"] if early_done else marks["end"]
    return first, last


class ViewGrads:
    """Gradients of one view: per-Gaussian tensors (raw-parameter space) and the deformation field's flat buffer."""
    __slots__ = ("xyz", "scaling", "rotation", "opacity", "shs", "means2D", "radii", "loss", "psnr", "early_exchanged")


class ViewPipeline:
    def __init__(self, gaussians, deformation_fields, background):
        self.g = gaussians
        self.net = deformation_fields            # DeformationFields, or None (coarse-only use)
        self.bg = background.detach().to(torch.float32).contiguous()
        self.N = self.W = self.H = -1
        self.capacity = 0
        self._one = None
        self.deform_numel = 0                    # floats reserved in the gradient arena for the deformation field
        # FDGS_ASIDE=0: the view-independent work runs on the view's own stream while the host waits for the
        # tile-list length, as it does in the coarse stage
        self.aside = os.environ.get("FDGS_ASIDE", "1") != "0"
        self._aside = None                       # (stream, event before the view, event after the work)

    # ------------------------------------------------------------------------------------------------------
    def usable(self, stage):
        g = self.g
        return (g.scaling_activation is torch.exp and g.opacity_activation is torch.sigmoid
                and g.rotation_activation is torch.nn.functional.normalize
                and ("coarse" in stage or (self.net is not None
                                           and self.net.deformation_net.W == _deformation.KERNEL_WIDTH)))

    def ensure(self, N, W, H, dev, deform_numel=None):
        """Allocates (or re-allocates, when a size changed) the persistent buffers; run() does it by itself, a
        caller that wants the gradient arena before the first run() calls it with the deformation field's size."""
        if deform_numel is None:
            deform_numel = self.deform_numel
        if (N, W, H, deform_numel) != (self.N, self.W, self.H, self.deform_numel) or self._one is None \
                or self._one.device != dev:
            self.deform_numel = deform_numel
            self._alloc(N, W, H, dev)

    def _alloc(self, N, W, H, dev):
        f32, i32 = torch.float32, torch.int32
        e = lambda *s, dtype=f32: torch.empty(*s, device=dev, dtype=dtype)
        self.N, self.W, self.H = N, W, H
        gx, gy = (W + _lib.TILE - 1) // _lib.TILE, (H + _lib.TILE - 1) // _lib.TILE
        # deformation outputs and what its backward needs
        self.d_out = [e(N, 3), e(N, 3), e(N, 4), e(N, 1), e(N, 16, 3)]
        self.shs_in = e(N, 16, 3)
        self.act = e((N + 31) // 32 * 32, 192)
        self.yhid = e((N + 15) // 16 * 16 + 16, 192)
        self.mask = e(N, 12, dtype=torch.int64)
        self.yfeat = e((N + 15) // 16 * 16 + 16, 48)
        self.hidden = None
        # rasterizer
        self.geom = e(N, _lib.GEOM_FLOATS)
        self.radii = e(N, dtype=i32)
        self.tiles_touched = e(N, dtype=i32)
        self.point_offsets = e(N, dtype=i32)
        self.scan_ws = e(max(1, lib.fdgs_raster_scan_workspace_bytes(N) // 4), dtype=i32)
        self.color = e(3, H, W)
        self.depth = e(1, H, W)
        self.final_T = e(H, W)
        self.n_contrib = e(H, W, dtype=i32)
        self.ranges = e(gx * gy, 2, dtype=i32)
        # tiles by decreasing list length: the blend launches' order (FDGS_TILE_ORDER=0: image order)
        self.tile_order = e(gx * gy, dtype=i32) if os.environ.get("FDGS_TILE_ORDER", "1") != "0" else None
        n_img = 3 * H * W
        # loss and PSNR scalars: two of each, used in turn, so that the values of a view stay readable while the
        # next view (or the next step's) is computed
        self._scalars = [(e(()), e(())), (e(()), e(()))]
        self._scalar_turn = 0
        self.l1_partial = e(2 * lib.fdgs_l1_loss_partials(n_img))
        self.grad_image = e(3, H, W)
        # backward; two sets of per-Gaussian gradient buffers so that a second view can be added to the first
        self.dgeom = torch.zeros(N, _lib.GEOM_FLOATS, device=dev, dtype=f32)
        # Every gradient of a set that leaves this object lives in ONE flat array ("arena"), in the order
        #   deformation field | xyz | means2D | scaling | rotation | opacity | shs        (each padded to 256 bytes)
        # so that a data-parallel exchange is an in-place all-reduce of a contiguous range of it, with no packing:
        # all of it; all but the deformation field (coarse stage); or in two parts - "early": means2D .. shs, final
        # once the rasterizer backward has run (these reach their parameters through the residual adds only), and
        # "late": deformation field + xyz, which the deformation backward produces.
        regions, marks, total = arena_layout(N, self.deform_numel)
        self.arena_regions, self.arena_marks = regions, marks
        self.sets, self.arenas = [], []
        for _ in range(2):
            arena = torch.zeros(total, device=dev, dtype=f32)
            views = dict(means3D=e(N, 3))
            for name, shape in regions:
                views[name] = arena[marks[name]:marks[name] + math.prod(shape)].view(shape)
            self.sets.append(views)
            self.arenas.append(arena)
        self._one = torch.ones(1, device=dev, dtype=f32)
        self.r_host = torch.empty(3, dtype=i32).pin_memory()
        self._deform_ws = None
        self._R_last = 0

    def exchange_range(self, fine, early_done):
        """The part of set 0's arena a data-parallel exchange still has to sum over the ranks: a 1-D view."""
        first, last = exchange_bounds(self.arena_marks, fine, early_done)
        return self.arenas[0][first:last]

    def early_views(self, buf):
        """The named gradients inside `buf`, a 1-D tensor laid out like an arena's early range (means2D .. shs)."""
        m = self.arena_marks
        out = {}
        for name, shape in self.arena_regions[2:]:
            off = m[name] - m["means2D"]
            out[name] = buf[off:off + math.prod(shape)].view(shape)
        return out

    def _list_buffers(self, R, dev):
        if R <= self.capacity:
            return
        step = 1 << max(int(R).bit_length() - 3, 10)
        cap = (R + step - 1) // step * step
        self.capacity = cap
        ws_bytes = lib.fdgs_raster_sort_workspace_bytes(cap)
        self.keys_a = torch.empty(cap, device=dev, dtype=torch.int64)
        self.vals_a = torch.empty(cap, device=dev, dtype=torch.int32)
        self.keys_b = torch.empty(cap, device=dev, dtype=torch.int64)
        self.vals_b = torch.empty(cap, device=dev, dtype=torch.int32)
        self.sort_ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        self.sort_ws_bytes = ws_bytes

    def _deform_buffers(self, P, N, dev):
        fb = lib.fdgs_deform_fwd_workspace_bytes(C.byref(P), N)
        bb = lib.fdgs_deform_bwd_workspace_bytes(C.byref(P), N)
        if self._deform_ws is None or self._deform_ws[0].numel() < fb or self._deform_ws[1].numel() < bb:
            self._deform_ws = (torch.empty(max(fb, 16), device=dev, dtype=torch.uint8),
                               torch.empty(max(bb, 16), device=dev, dtype=torch.uint8))
        return self._deform_ws[0], fb, self._deform_ws[1], bb

    # ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def run(self, cam, gt, scale, stage, grad_set=0, deform_grads=None, want_psnr=True, early_hook=None,
            while_waiting=None):
        """Forward, loss and backward of one view.  gt: [3,H,W] float32 on the device.  scale: the view's share of
        the batch loss (1 / views in the batch).  deform_grads: the 34 gradient tensors of the deformation field in
        the order of fdgs.deformation._trainable, accumulated into (fine stage).  Returns a ViewGrads whose tensors
        are this object's buffers of set `grad_set` (valid until that set is used again).  while_waiting: callable
        run after the tile-list length has been requested from the device and before the host waits for it - work
        enqueued there (anything that does not depend on this view) runs while that 4-byte read is in flight; it may
        return a device scalar (float32) that is added to this view's loss value.
        early_hook: callable(buf, radii) -> handle with wait(), or None.  Called (fine stage) as soon as the
        gradients that do not pass through the deformation backward are final - view-space gradient, scaling,
        rotation, opacity, SH: the arena's early range - and before that backward is enqueued, with a flat tensor
        it may overwrite (a data-parallel caller starts an all-reduce on it) and this view's radii; the returned
        ViewGrads' means2D / scaling / rotation / opacity / shs are then views of that tensor.  With grad_set 0 the
        tensor is a copy of the range (the backward still reads the view's own blocks as cotangents); with grad_set
        1 the view's range is added to set 0's - the running sum of the rank's earlier views - and that is handed
        over."""
        g = self.g
        xyz = g._xyz.detach()
        dev = xyz.device
        N = xyz.shape[0]
        W, H = int(cam.image_width), int(cam.image_height)
        self.ensure(N, W, H, dev)
        fine = "fine" in stage
        S = self.sets[grad_set]
        sc, ro, op = g._scaling.detach(), g._rotation.detach(), g._opacity.detach()
        f_dc, f_rest = g._features_dc.detach(), g._features_rest.detach()
        if not fine:        # the rasterizer reads one [N,16,3] array; the deformation field reads the two tensors
            torch.cat((f_dc, f_rest), dim=1, out=self.shs_in)
        tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        vm = cam.world_view_transform
        pm = cam.full_proj_transform
        cp = cam.camera_center
        acts = _lib.ACT_SCALE_EXP | _lib.ACT_ROTATION_NORMALIZE | _lib.ACT_OPACITY_SIGMOID
        side_term = side_pending = None

        if fine:
            net = self.net.deformation_net
            P = _deformation._fill_params(net)
            split = P.matmul == _lib.MATMUL_SPLIT_BF16
            if not split and self.hidden is None:
                self.hidden = torch.empty(N, net.W, device=dev, dtype=torch.float32)
            fws, fws_bytes, bws, bws_bytes = self._deform_buffers(P, N, dev)
            o = self.d_out
            a = _lib.DeformFwdArgs(N, P, ptr(xyz), ptr(sc), ptr(ro), ptr(op), ptr(f_dc), None, float(cam.time),
                                   ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(o[3]), ptr(o[4]),
                                   None if split else ptr(self.hidden), ptr(self.act) if split else None,
                                   ptr(self.mask) if split else None, ptr(self.yhid) if split else None,
                                   ptr(self.yfeat) if split else None, ptr(fws), fws_bytes, ptr(f_rest))
            # work that depends on no view (the caller's regulariser and gradient clears) goes to a second stream,
            # ordered after everything enqueued before this view and submitted behind the deformation forward: its
            # workgroups fill the compute units that kernel's last, partly empty round leaves idle (300,000 points
            # are 4.58 rounds of one-wave-per-SIMD tiles) instead of standing between the preprocess kernel and the
            # tile lists
            if while_waiting is not None and self.aside:
                cur = torch.cuda.current_stream()
                if self._aside is None or self._aside[0].device != cur.device:
                    with torch.cuda.device(cur.device):
                        self._aside = (torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event())
                side, before, after = self._aside
                before.record(cur)
            _lib.call("fdgs_deform_fwd", a)
            if while_waiting is not None and self.aside:
                side.wait_event(before)
                with torch.cuda.stream(side):
                    side_term = while_waiting()
                    after.record(side)
                while_waiting, side_pending = None, after
            m3, scales, rots, opac, shs = o[0], o[1], o[2], o[3].reshape(-1), o[4]
        else:
            m3, scales, rots, opac, shs = xyz, sc, ro, op.reshape(-1), self.shs_in

        sh_degree = int(g.active_sh_degree)
        pa = _lib.PreprocessFwdArgs(N, sh_degree, W, H, tanx, tany, 1.0, ptr(vm), ptr(pm), ptr(cp), ptr(m3),
                                    ptr(scales), ptr(rots), ptr(opac), ptr(shs), None, None, ptr(self.geom),
                                    ptr(self.radii), ptr(self.tiles_touched), ptr(self.point_offsets),
                                    ptr(self.scan_ws), acts)
        _lib.call("fdgs_raster_preprocess_fwd", pa)
        # the data-dependent list length R: one 4-byte device->host read (the step's only synchronisation)
        self.r_host.copy_(self.scan_ws[:3], non_blocking=True)     # depth range (two words) and R
        ready = torch.cuda.Event()
        ready.record()
        loss_term = while_waiting() if while_waiting is not None else side_term

        def binsort_args():
            return _lib.BinSortArgs(N, W, H, 0, ptr(self.geom), ptr(self.point_offsets), ptr(self.keys_a),
                                    ptr(self.vals_a), ptr(self.keys_b), ptr(self.vals_b), ptr(self.ranges),
                                    ptr(self.sort_ws), self.sort_ws_bytes, 0, 0, ptr(self.tile_order))
        # everything the host can do before it knows R is done before it waits - the argument blocks of all five calls
        # that follow included (a block is a dozen pointer look-ups in Python): the GPU idles from the end of the scan
        # until the first launch after the wait, and between short kernels while the host is still catching up
        n_img = 3 * H * W
        self._scalar_turn ^= 1
        loss_buf, psnr_buf = self._scalars[self._scalar_turn]
        d_m3 = S["means3D"] if fine else S["xyz"]

        def list_args():
            plist = ptr(self.vals_b)
            # which pixel blocks an entry reaches: found by the forward blend kernel, read by the backward one; four
            # bytes per entry in the buffer of the unsorted keys, which is free once the lists are built
            masks = ptr(self.keys_a)
            return (_lib.BlendFwdArgs(W, H, ptr(self.ranges), plist, ptr(self.geom), ptr(self.bg), ptr(self.color),
                                      ptr(self.depth), ptr(self.final_T), ptr(self.n_contrib), ptr(self.tile_order),
                                      masks),
                    # dgeom: zero from its allocation, re-zeroed by preprocess_bwd below
                    _lib.BlendBwdArgs(W, H, ptr(self.ranges), plist, ptr(self.geom), ptr(self.bg), ptr(self.final_T),
                                      ptr(self.n_contrib), ptr(self.grad_image), ptr(self.dgeom),
                                      ptr(self.tile_order), masks))
        bs = binsort_args() if self.capacity > 0 else None
        counted = False
        if bs is not None:              # the counting part of the tile lists runs while the list length travels
            bs.phase = _lib.BINSORT_COUNT
            _lib.call("fdgs_raster_bin_sort", bs)
            counted = True
        bf, bb = list_args() if self.capacity > 0 else (None, None)
        # the view's loss is this term alone: its gradient comes out of the same pass over the two images
        l1 = _lib.L1FwdArgs(n_img, ptr(self.color), ptr(gt), float(scale), ptr(loss_buf), ptr(self.l1_partial),
                            ptr(psnr_buf) if want_psnr else None, ptr(self.grad_image), ptr(loss_term))
        pb = _lib.PreprocessBwdArgs(N, sh_degree, W, H, tanx, tany, 1.0, ptr(vm), ptr(pm), ptr(cp), ptr(m3),
                                    ptr(scales), ptr(rots), ptr(shs), None, None, ptr(self.geom), ptr(self.dgeom),
                                    ptr(d_m3), ptr(S["means2D"]), ptr(S["scaling"]), ptr(S["rotation"]),
                                    ptr(S["opacity"]), ptr(S["shs"]), None, None, acts, 1)
        ready.synchronize()
        dmin, dmax, R = (int(v) & 0xffffffff for v in self.r_host.tolist())
        _rasterizer.LAST_NUM_RENDERED = R
        if bs is None or R > self.capacity:
            self._list_buffers(max(R, 1), dev)
            bs = binsort_args()
            bf, bb = list_args()
            counted = False             # another workspace: the counts are taken again
        bs.num_rendered, bs.depth_bits_min, bs.depth_bits_max = R, dmin, dmax
        bs.phase = _lib.BINSORT_FINISH if counted else _lib.BINSORT_ALL
        if R == 0:
            bf.point_list = bb.point_list = None
        if side_pending is not None:
            torch.cuda.current_stream().wait_event(side_pending)
        _lib.call("fdgs_raster_bin_sort", bs)
        _lib.call("fdgs_raster_blend_fwd", bf)
        _lib.call("fdgs_l1_loss_fwd", l1)
        _lib.call("fdgs_raster_blend_bwd", bb)
        _lib.call("fdgs_raster_preprocess_bwd", pb)
        out = ViewGrads()
        early = S
        pending = None
        if fine:
            if early_hook is not None:       # data parallel: these gradients are final here and can start travelling
                m = self.arena_marks
                own = self.arenas[grad_set][m["means2D"]:m["end"]]
                if grad_set != 0:            # the running sum of the rank's earlier views: this view's range joins it
                    buf = self.arenas[0][m["means2D"]:m["end"]]
                    buf.add_(own)
                else:
                    buf = own.clone()        # the kernels below still read this view's own blocks as cotangents
                pending = early_hook(buf, self.radii)
                # With grad_set 0 and no collective started (the hook returned None) the copy is dropped: the view's
                # own blocks stay the gradients and travel with the arena afterwards.  With grad_set 1 the range has
                # been added to set 0's in any case, and that sum is what the caller continues with.
                if pending is not None or grad_set != 0:
                    early = self.early_views(buf)
            G = _deformation._fill_params(net, deform_grads)
            b = _lib.DeformBwdArgs(N, P, ptr(xyz), None, float(cam.time), None if split else ptr(self.hidden),
                                   ptr(self.act) if split else None, ptr(self.mask) if split else None,
                                   ptr(self.yhid) if split else None, ptr(self.yfeat) if split else None, ptr(fws),
                                   ptr(S["means3D"]), ptr(S["scaling"]), ptr(S["rotation"]), ptr(S["opacity"]),
                                   ptr(S["shs"]), ptr(S["xyz"]), G, ptr(bws), bws_bytes, *_lib.side_resources())
            _lib.call("fdgs_deform_bwd", b)
            if pending is not None:
                pending.wait()
        out.xyz, out.scaling, out.rotation, out.opacity = S["xyz"], early["scaling"], early["rotation"], early["opacity"]
        out.shs, out.means2D, out.radii = early["shs"], early["means2D"], self.radii
        out.early_exchanged = early is not S
        out.loss, out.psnr = loss_buf, (psnr_buf if want_psnr else None)
        return out
