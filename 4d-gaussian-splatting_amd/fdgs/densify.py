"""Densification and pruning of the Gaussians with their optimizer state, through libfdgs.so.

The reference does this in its trainer (src/runner/fdgs_trainer.py:304-492) as boolean-mask gathers and
torch.cat calls per tensor - six parameters, twelve Adam moments, five per-Gaussian statistics - and three
passes (clone, split, prune of the split parents).  Here one densification is: classify (one launch), plan
(three launches), one host read of four counts, apply (one launch for all arrays), split children (one
launch).  The result has the row order of the reference's sequence.

`GaussianSurgery` carries the reference's method names without the leading underscore: densify, prune,
prune_points, densify_and_clone, densify_and_split, reset_opacity.
"""
import torch
from torch import nn

from . import _lib
from ._lib import lib, ptr

PARAM_NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
         "scaling": "_scaling", "rotation": "_rotation"}


def _f32c(t):
    t = t.detach()
    return t if (t.dtype is torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


@torch.no_grad()
def densify_classify(xyz_gradient_accum, denom, scaling, grad_threshold, size_limit):
    """uint8 class per Gaussian (KEEP / CLONE / SPLIT): fdgs_trainer.py:487-492, :426-430, :392-397."""
    _lib.require_gpu(xyz_gradient_accum, denom, scaling)
    n = scaling.shape[0]
    if xyz_gradient_accum.numel() != n or denom.numel() != n or scaling.shape != (n, 3):
        raise ValueError("densify_classify: accumulators with N entries, scaling [N,3]")
    cls = torch.empty(n, device=scaling.device, dtype=torch.uint8)
    acc, den, sc = _f32c(xyz_gradient_accum), _f32c(denom), _f32c(scaling)
    _lib.call("fdgs_densify_classify", _lib.DensifyClassifyArgs(n, ptr(acc), ptr(den), ptr(sc), float(grad_threshold),
                                                                float(size_limit), ptr(cls)))
    return cls


@torch.no_grad()
def prune_classify(opacity, max_radii2D, scaling, min_opacity, max_screen_size, world_limit):
    """uint8 class per Gaussian (KEEP / DROP): fdgs_trainer.py:477-484.  max_screen_size None or 0: off."""
    _lib.require_gpu(opacity, max_radii2D, scaling)
    n = opacity.shape[0]
    cls = torch.empty(n, device=opacity.device, dtype=torch.uint8)
    mss = float(max_screen_size) if max_screen_size else 0.0
    op = _f32c(opacity)
    rad = _f32c(max_radii2D) if mss > 0 else None
    sc = _f32c(scaling) if mss > 0 else None
    _lib.call("fdgs_prune_classify", _lib.PruneClassifyArgs(n, ptr(op), ptr(rad), ptr(sc), float(min_opacity), mss,
                                                            float(world_limit), ptr(cls)))
    return cls


@torch.no_grad()
def surgery_plan(cls):
    """(src, rows_after, survivors, clones, splits): src[r] = old row of new row r (int32, length 2 N, the
    first rows_after entries are meaningful).  One host synchronisation (the four counts)."""
    _lib.require_gpu(cls)
    if cls.dtype is not torch.uint8 or not cls.is_contiguous():
        raise ValueError("classes must be a contiguous uint8 tensor")
    n = cls.numel()
    dev = cls.device
    src = torch.empty(max(2 * n, 1), device=dev, dtype=torch.int32)
    counts = torch.empty(4, device=dev, dtype=torch.int32)
    ws_bytes = lib.fdgs_surgery_workspace_bytes(n)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    _lib.call("fdgs_surgery_plan", _lib.SurgeryPlanArgs(n, ptr(cls), ptr(src), ptr(counts), ptr(ws), ws_bytes))
    rows_after, survivors, clones, splits = counts.tolist()
    return src, rows_after, survivors, clones, splits


@torch.no_grad()
def surgery_apply(src, rows_after, rows_carried, arrays):
    """arrays: list of (tensor [N, ...], zero_new).  Returns the moved tensors [rows_after, ...]: row r is
    old row src[r]; with zero_new the rows r >= rows_carried are zeros."""
    outs = []
    batch = []
    for t, zero_new in arrays:
        _lib.require_gpu(t)
        t = t.detach()
        if not t.is_contiguous():
            t = t.contiguous()
        out = torch.empty((rows_after,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        outs.append(out)
        row_bytes = t.element_size()
        for d in t.shape[1:]:
            row_bytes *= d
        if rows_after and row_bytes:
            batch.append((t, out, row_bytes, _lib.SURGERY_ZERO_NEW if zero_new else 0))
    for i in range(0, len(batch), _lib.SURGERY_MAX_ARRAYS):
        part = batch[i:i + _lib.SURGERY_MAX_ARRAYS]
        a = _lib.SurgeryApplyArgs()
        a.rows_after, a.rows_carried, a.src, a.num_arrays = rows_after, rows_carried, ptr(src), len(part)
        for k, (t, out, rb, fl) in enumerate(part):
            a.in_[k] = t.data_ptr() if t.numel() else None
            a.out[k] = out.data_ptr()
            a.row_bytes[k] = rb
            a.flags[k] = fl
        _lib.call("fdgs_surgery_apply", a)
    return outs


class GaussianSurgery:
    """Binds the objects the reference's trainer edits: `gaussians` (attributes _xyz, _features_dc,
    _features_rest, _opacity, _scaling, _rotation, percent_dense), the optimizer (parameter groups named
    xyz / f_dc / f_rest / opacity / scaling / rotation with one tensor each, torch.optim.Adam or
    fdgs.adam.FusedAdam), `deforms` (deformation_table, deformation_accum; optional) and the holder of the
    statistics xyz_gradient_accum / denom / max_radii2D (the reference keeps them on `gaussians`)."""

    def __init__(self, gaussians, optimizer, deforms=None, stats=None, percent_dense=None, noise_fn=None):
        self.gaussians = gaussians
        self.optimizer = optimizer
        self.deforms = deforms
        self.stats = stats if stats is not None else gaussians
        # configs/train/dynerf.yaml: percent_dense 0.01 (the reference keeps it on the Gaussian model)
        self.percent_dense = percent_dense if percent_dense is not None else getattr(gaussians, "percent_dense", 0.01)
        # noise_fn(count, device) -> [count, 3] unit normal samples; data-parallel replicas must all get the same
        self.noise_fn = noise_fn or (lambda count, device: torch.randn(count, 3, device=device, dtype=torch.float32))

    # -- optimizer plumbing (fdgs_trainer.py:304-366) ---------------------------------------------------
    def _groups(self):
        found = {}
        for group in self.optimizer.param_groups:
            if len(group["params"]) == 1 and group.get("name") in PARAM_NAMES:
                found[group["name"]] = group
        missing = [n for n in PARAM_NAMES if n not in found]
        if missing:
            raise ValueError("optimizer has no single-tensor parameter group named " + ", ".join(missing))
        return found

    def _install(self, name, group, tensor, exp_avg, exp_avg_sq):
        old = group["params"][0]
        st = self.optimizer.state.pop(old, None) if old in self.optimizer.state else None
        new = nn.Parameter(tensor.requires_grad_(True))
        group["params"][0] = new
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = exp_avg, exp_avg_sq
            self.optimizer.state[new] = st
        setattr(self.gaussians, _ATTR[name], new)
        return new

    def _moments(self, group):
        p = group["params"][0]
        st = self.optimizer.state.get(p) if p in self.optimizer.state else None
        if st is None or "exp_avg" not in st:
            return None
        return st["exp_avg"], st["exp_avg_sq"]

    # -- one pass through a class vector ----------------------------------------------------------------
    @torch.no_grad()
    def apply_classes(self, cls, noise=None, reset_stats=False):
        """Rebuilds every per-Gaussian array for the classes `cls`.  noise: [2 S, 3] unit normal samples for the
        split children (drawn here when None).  reset_stats: the statistics restart from zero at the new size
        (densification, fdgs_trainer.py:468-475); otherwise they are carried row by row (pruning, :381-386)."""
        groups = self._groups()
        src, rows_after, survivors, clones, splits = surgery_plan(cls)
        old = {n: groups[n]["params"][0].detach() for n in PARAM_NAMES}
        arrays, slots = [], []
        for n in PARAM_NAMES:
            arrays.append((old[n], False)); slots.append(("p", n))
            mom = self._moments(groups[n])
            if mom is not None:
                arrays.append((mom[0], True)); slots.append(("m", n))
                arrays.append((mom[1], True)); slots.append(("v", n))
        stat_names = ("xyz_gradient_accum", "denom", "max_radii2D")
        if not reset_stats:
            for k in stat_names:
                arrays.append((getattr(self.stats, k), False)); slots.append(("s", k))
        if self.deforms is not None:
            arrays.append((self.deforms.deformation_table, False)); slots.append(("d", "deformation_table"))
            if not reset_stats:
                arrays.append((self.deforms.deformation_accum, False)); slots.append(("d", "deformation_accum"))
        outs = surgery_apply(src, rows_after, survivors, arrays)
        new = dict(zip(slots, outs))
        if splits:
            dev = old["xyz"].device
            if noise is None:
                noise = self.noise_fn(2 * splits, dev)
            noise = _f32c(noise)
            if noise.shape != (2 * splits, 3):
                raise ValueError("noise must be [2 * splits, 3] = [%d, 3]" % (2 * splits))
            _lib.require_gpu(noise)
            _lib.call("fdgs_split_children", _lib.SplitChildrenArgs(
                2 * splits, survivors + clones, ptr(src), ptr(_f32c(old["xyz"])), ptr(_f32c(old["scaling"])),
                ptr(_f32c(old["rotation"])), ptr(noise), ptr(new[("p", "xyz")]), ptr(new[("p", "scaling")])))
        for n in PARAM_NAMES:
            self._install(n, groups[n], new[("p", n)], new.get(("m", n)), new.get(("v", n)))
        dev = new[("p", "xyz")].device
        if reset_stats:
            self.stats.xyz_gradient_accum = torch.zeros(rows_after, 1, device=dev)
            self.stats.denom = torch.zeros(rows_after, 1, device=dev)
            self.stats.max_radii2D = torch.zeros(rows_after, device=dev)
        else:
            for k in stat_names:
                setattr(self.stats, k, new[("s", k)])
        if self.deforms is not None:
            self.deforms.deformation_table = new[("d", "deformation_table")]
            self.deforms.deformation_accum = (torch.zeros(rows_after, 3, device=dev) if reset_stats
                                              else new[("d", "deformation_accum")])
        return {"rows": rows_after, "survivors": survivors, "clones": clones, "splits": splits}

    # -- the reference's drivers -------------------------------------------------------------------------
    def _classify_densify(self, max_grad, extent):
        return densify_classify(self.stats.xyz_gradient_accum, self.stats.denom, self.gaussians._scaling,
                                max_grad, self.percent_dense * extent)

    def densify(self, max_grad, extent, noise=None):
        """fdgs_trainer.py:487-492: clone the small, split the large Gaussians whose mean view-space gradient
        reaches max_grad - one pass instead of clone, split and prune-the-parents."""
        return self.apply_classes(self._classify_densify(max_grad, extent), noise=noise, reset_stats=True)

    def densify_and_clone(self, max_grad, extent):
        """fdgs_trainer.py:426-449 alone."""
        cls = self._classify_densify(max_grad, extent)
        cls[cls == _lib.CLS_SPLIT] = _lib.CLS_KEEP
        return self.apply_classes(cls, reset_stats=True)

    def densify_and_split(self, max_grad, extent, noise=None):
        """fdgs_trainer.py:388-424 alone."""
        cls = self._classify_densify(max_grad, extent)
        cls[cls == _lib.CLS_CLONE] = _lib.CLS_KEEP
        return self.apply_classes(cls, noise=noise, reset_stats=True)

    def prune_points(self, mask):
        """fdgs_trainer.py:368-386: removes the Gaussians where mask is True."""
        _lib.require_gpu(mask)
        cls = mask.reshape(-1).to(torch.uint8) * _lib.CLS_DROP
        return self.apply_classes(cls.contiguous(), reset_stats=False)

    def prune(self, min_opacity, extent, max_screen_size):
        """fdgs_trainer.py:477-484."""
        cls = prune_classify(self.gaussians._opacity.reshape(-1), self.stats.max_radii2D, self.gaussians._scaling,
                             min_opacity, max_screen_size, 0.1 * extent)
        return self.apply_classes(cls, reset_stats=False)

    @torch.no_grad()
    def reset_opacity(self, cap=0.01):
        """src/model/gaussian_model.py:173-176: opacities capped at `cap`, the Adam moments of the opacity tensor
        zeroed (fdgs_trainer.py:304-317; the reference calls a method GaussianModel does not have and would
        raise there - this does what that call names)."""
        group = self._groups()["opacity"]
        p = group["params"][0]
        _lib.require_gpu(p)
        new = _f32c(p).clone()
        rc = lib.fdgs_reset_opacity(new.data_ptr(), new.numel(), float(cap), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "fdgs_reset_opacity")
        mom = self._moments(group)
        zeros = (torch.zeros_like(new), torch.zeros_like(new)) if mom is not None else (None, None)
        return self._install("opacity", group, new, *zeros)
