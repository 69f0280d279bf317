"""Multi-tensor Adam in one launch, with the surface of `torch.optim.Adam` that the
reference's training loop touches (src/runner/fdgs_trainer.py:262-274: parameter
groups with a `name` and an `lr`, eps 1e-15; `step()` at :204, `zero_grad()`;
`param_groups[i]["lr"]` rewritten by the learning-rate schedule at :276-302;
`state[p]["exp_avg"]`, `["exp_avg_sq"]` edited by the densification surgery at
:304-366).  No weight decay, no amsgrad, one shared step count."""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, ptr


class FusedAdam:
    def __init__(self, params, lr=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.param_groups = []
        for grp in params:
            g = dict(grp)
            g["params"] = list(g["params"])
            g.setdefault("lr", lr)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            self.param_groups.append(g)
        if len({(g["betas"], g["eps"]) for g in self.param_groups}) != 1:
            raise ValueError("betas and eps must be the same in every group (one launch)")
        self.state = {}
        self._step = 0

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self):
        entries = []
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state.get(p)
                if st is None or st["exp_avg"].shape != p.shape:
                    st = {"exp_avg": torch.zeros_like(p, memory_format=torch.contiguous_format),
                          "exp_avg_sq": torch.zeros_like(p, memory_format=torch.contiguous_format)}
                    self.state[p] = st
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not (p.is_cuda and grad.is_cuda):
                    raise RuntimeError("libfdgs kernels run on the GPU only")
                if not p.is_contiguous() or p.dtype is not torch.float32 or grad.dtype is not torch.float32:
                    raise ValueError("FusedAdam needs contiguous float32 parameters and gradients")
                entries.append((p, grad, st, float(g["lr"])))
        if not entries:
            return
        self._step += 1
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        for st in self.state.values():
            st["step"] = self._step
        for i in range(0, len(entries), _lib.ADAM_MAX_TENSORS):
            part = entries[i:i + _lib.ADAM_MAX_TENSORS]
            a = _lib.AdamArgs()
            a.num_tensors, a.step, a.beta1, a.beta2, a.eps = len(part), self._step, b1, b2, eps
            lrs, numels, pp, gp, mp, vp = a.lr, a.numel, a.param, a.grad, a.exp_avg, a.exp_avg_sq
            for t, (p, grad, st, lr) in enumerate(part):
                lrs[t] = lr; numels[t] = p.numel()
                pp[t] = p.data_ptr(); gp[t] = grad.data_ptr()
                mp[t] = st["exp_avg"].data_ptr(); vp[t] = st["exp_avg_sq"].data_ptr()
            _lib.call("fdgs_adam_step", a)
