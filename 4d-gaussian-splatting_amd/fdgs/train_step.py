"""One training iteration over a batch of views, mirroring the reference's loop
body (src/runner/fdgs_trainer.py:112-205): per-view render, L1 loss over the
batch, backward, view-space gradient / radii / visibility statistics, Adam step
with the reference's eight parameter groups (:262-274, eps 1e-15).  With
torch.distributed initialised the views are sharded over ranks and gradients
and statistics are all-reduced (fdgs.dp)."""
import os

import torch

from . import _lib, dp, losses, schedule
from .renderer import render


class DeformHolder:
    """What render() needs from the reference's DeformationWrapper
    (src/model/deformation_wrapper.py:9-24): a `deformation_fields` attribute."""

    def __init__(self, deformation_fields):
        self.deformation_fields = deformation_fields


class TrainStep:
    # time_smoothness_weight, l1_time_planes, plane_tv_weight of configs/train/dynerf.yaml:53-55
    DYNERF_REGULATION = (0.001, 0.0001, 0.0002)

    def __init__(self, gaussians, deforms, background, lr_scale=1.0, stage="fine", fused_adam=True,
                 regulation=DYNERF_REGULATION, lambda_dssim=0.0):
        self.g, self.deforms, self.bg, self.stage = gaussians, deforms, background, stage
        self.lambda_dssim = float(lambda_dssim)     # configs/train/dynerf.yaml:36 ships 0
        self.last_psnr = None                       # fdgs_trainer.py:131: mean PSNR of the local views, on device
        # fdgs_trainer.py:133-135: the grid regulariser joins the loss in the fine stage when its weight is set;
        # the stage is looked at per call, so one object (one optimizer, one set of moments) spans coarse -> fine
        self.regulation_weights = tuple(regulation) if (regulation and regulation[0] != 0) else None
        df = deforms.deformation_fields
        groups = [   # configs/train/dynerf.yaml:23-35, fdgs_trainer.py:262-272
            {"params": [gaussians._xyz], "lr": 0.00016 * lr_scale, "name": "xyz"},
            {"params": list(df.get_mlp_parameters()), "lr": 0.00016 * lr_scale, "name": "deformation"},
            {"params": list(df.get_grid_parameters()), "lr": 0.0016 * lr_scale, "name": "grid"},
            {"params": [gaussians._features_dc], "lr": 0.0025, "name": "f_dc"},
            {"params": [gaussians._features_rest], "lr": 0.0025 / 20.0, "name": "f_rest"},
            {"params": [gaussians._opacity], "lr": 0.05, "name": "opacity"},
            {"params": [gaussians._scaling], "lr": 0.005, "name": "scaling"},
            {"params": [gaussians._rotation], "lr": 0.001, "name": "rotation"},
        ]
        _lib.require_gpu(gaussians._xyz)    # every kernel of the step lives in libfdgs.so: there is no CPU path
        if fused_adam:                      # one launch for all 40 tensors
            from .adam import FusedAdam
            self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        self.learning_rates = schedule.LearningRates(spatial_lr_scale=lr_scale)   # fdgs_trainer.py:275-286
        self._bind_parameters()
        self.early_exchange = os.environ.get("FDGS_DP_EARLY", "1") != "0"
        # the straight C-ABI chain on persistent buffers (fdgs/pipeline.py) instead of render() + autograd;
        # FDGS_PIPELINE=0 keeps the autograd route (same kernels, more host work)
        from .pipeline import ViewPipeline
        self.pipeline = ViewPipeline(gaussians, getattr(deforms, "deformation_fields", None), background)
        self.use_pipeline = os.environ.get("FDGS_PIPELINE", "1") != "0"
        self._dflat = None
        dp.broadcast_parameters(self.params)      # replicas start identical whatever the ranks' RNG state was
        n = gaussians._xyz.shape[0]
        dev = gaussians._xyz.device
        self.xyz_gradient_accum = torch.zeros(n, 1, device=dev)
        self.denom = torch.zeros(n, 1, device=dev)
        self.max_radii2D = torch.zeros(n, device=dev)

    def _bind_parameters(self):
        """The flat parameter list and the gradient buckets of the tensors the optimizer holds now."""
        self.params = [p for grp in self.optimizer.param_groups for p in grp["params"]]
        self.bucket = dp.GradBucket(self.params)
        # the SH gradients can be exchanged early (fdgs.deformation.RESIDUAL_GRAD_HOOK); the rest then goes here
        sh_ids = {id(self.g._features_dc), id(self.g._features_rest)}
        self.bucket_without_sh = dp.GradBucket([p for p in self.params if id(p) not in sh_ids])

    # -- densification / pruning between steps (fdgs_trainer.py:191-200) --------------------------------
    def surgery(self, percent_dense=0.01):
        from .densify import GaussianSurgery

        def noise(count, device):       # every replica must place the split children identically
            t = torch.randn(count, 3, device=device, dtype=torch.float32)
            dp.broadcast_parameters([t])
            return t
        deforms = self.deforms if hasattr(self.deforms, "deformation_table") else None
        return GaussianSurgery(self.g, self.optimizer, deforms=deforms, stats=self, percent_dense=percent_dense,
                               noise_fn=noise)

    def densify(self, max_grad, extent, noise=None, percent_dense=0.01):
        out = self.surgery(percent_dense).densify(max_grad, extent, noise=noise)
        self._bind_parameters()
        return out

    def prune(self, min_opacity, extent, max_screen_size, percent_dense=0.01):
        out = self.surgery(percent_dense).prune(min_opacity, extent, max_screen_size)
        self._bind_parameters()
        return out

    def reset_opacity(self):
        self.surgery().reset_opacity()
        self._bind_parameters()

    # -- the per-view work, two routes to the same gradients ---------------------------------------------------
    def _views_direct(self, views, scale, early_hook, while_waiting=None, like=None):
        """fdgs/pipeline.py: the chain of C-ABI calls on persistent buffers; gradients attached to .grad here.
        Every gradient (and the view-space gradient) is a view of the pipeline's flat gradient arena.  `like`: a
        camera of the global batch, for a rank that has no view of its own this step and only needs the arena (of
        zeros) to take part in the exchange."""
        g = self.g
        fine = "fine" in self.stage
        dev = g._xyz.device
        n = g._xyz.shape[0]
        cam0 = views[0] if views else like
        dgrads = None
        deform_numel = None
        if fine:
            from . import deformation
            net = self.deforms.deformation_fields.deformation_net
            params = deformation._trainable(net)
            order, sizes, shapes = net._fdgs_grad_order, net._fdgs_grad_sizes, net._fdgs_grad_shapes
            deform_numel = sum(sizes)
        self.pipeline.ensure(n, int(cam0.image_width), int(cam0.image_height), dev, deform_numel)
        acc = self.pipeline.sets[0]
        if fine:
            flat = acc["dflat"]
            if self._dflat is None or self._dflat[0].data_ptr() != flat.data_ptr():
                dgrads = [None] * len(params)
                for i, piece, shape in zip(order, flat.split_with_sizes(sizes), shapes):
                    dgrads[i] = piece.view(shape)
                self._dflat = (flat, dgrads)
            flat, dgrads = self._dflat
            for p, gr in zip(params, dgrads):     # attached now: the regulariser adds to them before the first
                p.grad = gr                        # view's backward does; zeroed where the regulariser runs
        loss = psnr_sum = radii = None
        final = acc          # where the rank's summed means2D / scaling / rotation / opacity / shs end up
        m = self.pipeline.arena_marks
        if not views:
            self.pipeline.arenas[0].zero_()
            radii = torch.zeros(n, dtype=torch.int32, device=dev)
        for i, cam in enumerate(views):
            # The gradients that do not pass through the deformation backward are complete for the rank after its
            # LAST view's rasterizer backward: they are handed to the hook there (one view: a copy of the view's
            # range; several: the running sum in set 0, this view's range added), with the rank's radii so far.
            last = i == len(views) - 1
            hook_now = None
            if early_hook is not None and fine and last:
                def hook_now(buf, view_radii, prev=radii):
                    return early_hook(buf, view_radii if prev is None else torch.maximum(prev, view_radii))
            vg = self.pipeline.run(cam, cam.original_image, scale, self.stage, grad_set=min(i, 1), deform_grads=dgrads,
                                   early_hook=hook_now, while_waiting=while_waiting if i == 0 else None)
            # a view's scalars live in one of the pipeline's two buffers, valid until the view after the next
            loss = vg.loss if loss is None else loss + vg.loss
            psnr_sum = vg.psnr if psnr_sum is None else psnr_sum + vg.psnr
            if i == 0:
                radii = vg.radii if len(views) == 1 else vg.radii.clone()
                if vg.early_exchanged:          # one view: its early range travels as a copy
                    final = dict(means2D=vg.means2D, scaling=vg.scaling, rotation=vg.rotation, opacity=vg.opacity,
                                 shs=vg.shs)
            else:
                own = self.pipeline.arenas[1]
                if vg.early_exchanged:          # the early range was added for the exchange already: xyz is left
                    acc["xyz"].add_(vg.xyz)
                else:                           # one add over xyz .. shs (the deformation field accumulates in place)
                    self.pipeline.arenas[0][m["xyz"]:m["end"]].add_(own[m["xyz"]:m["end"]])
                radii = torch.maximum(radii, vg.radii)
        if views:
            self.last_psnr = psnr_sum / len(views) if len(views) > 1 else psnr_sum
        g._xyz.grad = acc["xyz"]
        g._scaling.grad, g._rotation.grad, g._opacity.grad = final["scaling"], final["rotation"], final["opacity"]
        g._features_dc.grad, g._features_rest.grad = final["shs"][:, :1, :], final["shs"][:, 1:, :]
        return loss, radii, None, final["means2D"]      # visibility = radii > 0 where a caller needs it

    def _views_autograd(self, views, scale, hook):
        """render() + autograd: the reference's interface (any activation callables, SSIM term)."""
        radii_list, vis_list, vsp_list = [], [], []
        loss = None
        psnr_sum = None
        for cam in views:
            pkg = render(cam, self.g, self.deforms, self.bg, stage=self.stage, cam_type="dynerf")
            # loss and the iteration's PSNR (fdgs_trainer.py:130-131) from one pass over the two images
            l1, view_psnr = losses.l1_loss_psnr(pkg["render"], cam.original_image, scale)
            psnr_sum = view_psnr if psnr_sum is None else psnr_sum + view_psnr
            loss = l1 if loss is None else loss + l1
            if self.lambda_dssim != 0:      # fdgs_trainer.py:136-138; the mean over the batch is the mean of the views'
                loss = loss + (self.lambda_dssim * scale) * (1.0 - losses.ssim(pkg["render"], cam.original_image))
            radii_list.append(pkg["radii"])
            vis_list.append(pkg["visibility_filter"])
            vsp_list.append(pkg["viewspace_points"])
        self.last_psnr = psnr_sum / len(views) if (psnr_sum is not None and len(views) > 1) else psnr_sum
        if loss is None:
            return None, None, None, None
        from . import deformation
        # with several views autograd sums their gradients itself: nothing is final before the last backward ends
        deformation.RESIDUAL_GRAD_HOOK = hook if len(views) == 1 else None
        try:
            loss.backward()
        finally:
            deformation.RESIDUAL_GRAD_HOOK = None
        if len(views) == 1:
            return loss, radii_list[0], vis_list[0], vsp_list[0].grad
        radii = torch.stack(radii_list, 0).max(dim=0).values
        vis = torch.stack(vis_list, 0).any(dim=0)
        vgrad = torch.zeros_like(vsp_list[0])
        for v in vsp_list:
            vgrad = vgrad + v.grad
        return loss, radii, vis, vgrad

    @property
    def regulation(self):
        return self.regulation_weights if "fine" in self.stage else None

    def __call__(self, global_views, iteration=None, stage=None, before_optimizer_step=None):
        """global_views: the whole batch (identical on every rank); returns the local loss value tensor (with one
        local view on the direct route it is one of two device scalars used in turn: read it, or copy it, before
        the step after the next; `last_psnr` likewise).
        iteration: the trainer's counter; when given, the scheduled learning rates are set first
        (fdgs_trainer.py:216, 288-302).
        stage: "coarse" / "fine" for this and the following calls (fdgs_trainer.py:114); None keeps the last one.
        before_optimizer_step: callable(self) run between the statistics update and the optimizer step - where the
        reference densifies, prunes and resets opacities (fdgs_trainer.py:191-200, then :203-205).  Tensors the
        surgery replaces have no gradient afterwards and are not stepped in that iteration, as in the reference."""
        if stage is not None:
            self.stage = stage
        if iteration is not None:
            self.learning_rates.update(self.optimizer, iteration)
        views = dp.shard_views(global_views)
        scale = dp.local_loss_scale(len(views), len(global_views))
        n = self.g._xyz.shape[0]
        dev = self.g._xyz.device
        # fdgs_trainer.py:133-135.  A view-independent term: every rank takes 1/world of it and the gradient
        # all-reduce (a sum) completes it.
        reg = None
        reg_w = None
        if self.regulation is not None:
            share = 1.0 / dp.rank_world()[1]
            reg_w = tuple(w * share for w in self.regulation)
        early = []          # what has been exchanged while the last deformation backward ran: ("sh", tensor) on the
        hook = None         # autograd route; ("range", flat tensor, radii, pending MAX of the radii) on the direct one
        early_hook = None
        # the same number of views on every rank (so all ranks take this branch together): after a rank's last
        # view's rasterizer backward everything but the deformation field's and the positions' gradients is complete
        # and is exchanged while that view's deformation backward runs
        if views and dp.is_dist() and len(global_views) % dp.rank_world()[1] == 0 and self.early_exchange:
            def hook(g_sh):             # autograd route (one view per rank): the SH gradient
                try:
                    work = dp.all_reduce_async(g_sh)
                except Exception:       # the collective could not be started (the same on every rank): the
                    self.early_exchange = False     # gradient then travels with the bucket, now and later
                    return None
                early.append(("sh", g_sh))
                return work

            def early_hook(buf, rank_radii):
                try:
                    work = dp.all_reduce_async(buf)
                    r = rank_radii.clone()
                    work_r = dp.all_reduce_async(r, torch.distributed.ReduceOp.MAX)
                except Exception:
                    self.early_exchange = False
                    return None
                early.append(("range", buf, r, work_r))
                return work
        direct = (self.use_pipeline and self.lambda_dssim == 0 and self.pipeline.usable(self.stage)
                  and all(getattr(c, "original_image", None) is not None for c in views))
        if direct:
            # the regulariser depends on no view: it is enqueued inside the first view's forward, where the host
            # waits for the tile-list length and the GPU would otherwise idle
            box = {}

            def regulariser():      # runs inside the first view's forward: returns the term to add to its loss value
                if self._dflat is not None and "fine" in self.stage:
                    self._dflat[0].zero_()      # the deformation gradients of this step start here
                if reg_w is not None:
                    box["reg"] = losses.add_regulation_grad_(self.deforms.deformation_fields, *reg_w)
                    return box["reg"]
                return None
            loss, radii, vis, vgrad = self._views_direct(views, scale, early_hook, regulariser if views else None,
                                                         like=global_views[0] if global_views else None)
            if "reg" in box:        # already inside `loss` (added on the device by the first view's loss kernel)
                reg, reg_w = None, None
        else:
            loss, radii, vis, vgrad = self._views_autograd(views, scale, hook)
        if reg_w is not None:
            # same gradients as adding the term to the loss, written into the planes' .grad in one launch
            reg = losses.add_regulation_grad_(self.deforms.deformation_fields, *reg_w)
        if not views and not direct:
            # a rank without views this step still takes part in the collectives, with zeros for exactly the
            # tensors that get a gradient in this stage on the other ranks (the deformation field: fine stage only)
            radii = torch.zeros(n, dtype=torch.int32, device=dev)
            vis = torch.zeros(n, dtype=torch.bool, device=dev)
            vgrad = torch.zeros(n, 3, device=dev)
            if dp.is_dist():
                gaussian_ids = {id(t) for t in self.g.parameters()} if hasattr(self.g, "parameters") else set()
                for p in self.params:
                    if p.requires_grad and p.grad is None and ("fine" in self.stage or id(p) in gaussian_ids):
                        p.grad = torch.zeros_like(p)
        if dp.is_dist():
            # `early` is the same on every rank: the hooks are armed only when every rank has the same number of
            # views, and fire once, in the last view's backward
            if direct:
                # every gradient and the view-space gradient are views of one flat array: one in-place SUM of the
                # range that has not travelled yet (deformation field + positions after an early exchange), nothing
                # packed; the radii went with the early exchange or take a MAX of their own
                went_early = len(early) == 1 and early[0][0] == "range"
                dp.all_reduce_(self.pipeline.exchange_range("fine" in self.stage, went_early),
                               torch.distributed.ReduceOp.SUM)
                if went_early:
                    if early[0][3] is not None:
                        early[0][3].wait()
                    radii = early[0][2]
                else:
                    radii, _, _ = dp.reduce_densification_stats(radii, None, None)
            else:
                (self.bucket_without_sh if len(early) == 1 else self.bucket).allreduce_()
                radii, _, vgrad = dp.reduce_densification_stats(radii, None, vgrad)
            vis = radii > 0
        # fdgs_trainer.py:181-182, gaussian_model.py:182-184 in one launch
        losses.add_densification_stats(radii, vgrad, self.xyz_gradient_accum, self.denom, self.max_radii2D)
        if before_optimizer_step is not None:
            before_optimizer_step(self)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        out = loss.detach() if loss is not None else torch.zeros((), device=dev)
        return out if reg is None else out + reg
