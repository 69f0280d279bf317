"""Multi-tensor Adam in one launch, with the surface of `torch.optim.Adam` that the
reference's training loop touches (src/runner/fdgs_trainer.py:262-274: parameter
groups with a `name` and an `lr`, eps 1e-15; `step()` at :204, `zero_grad()`;
`param_groups[i]["lr"]` rewritten by the learning-rate schedule at :276-302;
`state[p]["exp_avg"]`, `["exp_avg_sq"]` edited by the densification surgery at
:304-366; `state_dict()` stored by the checkpoint tuple at :517-531).  No weight
decay, no amsgrad.

As in torch.optim.Adam, a tensor without a gradient is skipped and every tensor
counts its own steps from its first gradient (`state[p]["step"]`): the deformation
field, which gets no gradient during the coarse stage, starts its bias correction
at step 1 when the fine stage begins."""
import torch

from . import _lib

# Bumped by every FusedAdam.step(): the kernel writes the parameters through raw pointers, which does not touch
# the tensors' autograd version counters; code that caches something derived from parameter values (the
# deformation field's channel-last plane copies) compares this beside Tensor._version.
GENERATION = [0]


def _row_slice(t):
    """(row length, row stride) when `t` [rows, ...] is a column slice of a wider row-major array - every row a
    contiguous run of elements, rows a constant stride apart - else None."""
    if t.dim() < 2 or t.shape[0] == 0:
        return None
    inner = t[0]
    if not inner.is_contiguous():
        return None
    length = inner.numel()
    stride = t.stride(0)
    return (length, stride) if stride >= length > 0 else None


class FusedAdam:
    def __init__(self, params, lr=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.param_groups = []
        for grp in params:
            g = dict(grp)
            g["params"] = list(g["params"])
            g.setdefault("lr", lr)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            # the remaining keys of torch.optim.Adam's groups, at the values this optimizer implements, so that
            # state_dict() loads into torch.optim.Adam and back
            for k, v in (("weight_decay", 0), ("amsgrad", False), ("maximize", False), ("foreach", None),
                         ("capturable", False), ("differentiable", False), ("fused", None),
                         ("decoupled_weight_decay", False)):
                g.setdefault(k, v)
            if g["weight_decay"] != 0 or g["amsgrad"] or g["maximize"]:
                raise ValueError("FusedAdam implements plain Adam: no weight decay, amsgrad or maximize")
            self.param_groups.append(g)
        if len({(tuple(g["betas"]), g["eps"]) for g in self.param_groups}) != 1:
            raise ValueError("betas and eps must be the same in every group (one launch)")
        self.state = {}

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
                if st is None or "exp_avg" not in st or st["exp_avg"].shape != p.shape:
                    st = {"step": 0,
                          "exp_avg": torch.zeros_like(p, memory_format=torch.contiguous_format),
                          "exp_avg_sq": torch.zeros_like(p, memory_format=torch.contiguous_format)}
                    self.state[p] = st
                grad, row = p.grad, None
                if not grad.is_contiguous():
                    row = _row_slice(grad)          # a column slice of a wider array is read in place
                    if row is None:
                        grad = grad.contiguous()
                if not (p.is_cuda and grad.is_cuda):
                    raise RuntimeError("libfdgs kernels run on the GPU only")
                if not p.is_contiguous() or p.dtype is not torch.float32 or grad.dtype is not torch.float32:
                    raise ValueError("FusedAdam needs contiguous float32 parameters and gradients")
                st["step"] = int(st.get("step", 0)) + 1
                entries.append((p, grad, st, float(g["lr"]), row))
        if not entries:
            return
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        for i in range(0, len(entries), _lib.ADAM_MAX_TENSORS):
            part = entries[i:i + _lib.ADAM_MAX_TENSORS]
            a = _lib.AdamArgs()
            a.num_tensors, a.beta1, a.beta2, a.eps = len(part), b1, b2, eps
            steps, lrs, numels, pp, gp, mp, vp = a.step, a.lr, a.numel, a.param, a.grad, a.exp_avg, a.exp_avg_sq
            for t, (p, grad, st, lr, row) in enumerate(part):
                steps[t] = st["step"]; lrs[t] = lr; numels[t] = p.numel()
                if row is not None:
                    a.grad_row_len[t], a.grad_row_stride[t] = row
                pp[t] = p.data_ptr(); gp[t] = grad.data_ptr()
                mp[t] = st["exp_avg"].data_ptr(); vp[t] = st["exp_avg_sq"].data_ptr()
            _lib.call("fdgs_adam_step", a)
        GENERATION[0] += 1

    # -- torch.optim.Optimizer's serialisation layout (fdgs_trainer.py:517-531 stores optimizer.state_dict()) --------
    def state_dict(self):
        index, packed_groups = {}, []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                index.setdefault(id(p), len(index))
                ids.append(index[id(p)])
            packed = {k: v for k, v in g.items() if k != "params"}
            packed["params"] = ids
            packed_groups.append(packed)
        state = {}
        for p, st in self.state.items():
            if id(p) in index:
                state[index[id(p)]] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"],
                                       "exp_avg_sq": st["exp_avg_sq"]}
        return {"state": state, "param_groups": packed_groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(a["params"]) != len(b["params"])
                                                        for a, b in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has different parameter groups")
        by_index = {}
        for saved, g in zip(groups, self.param_groups):
            for k, v in saved.items():
                if k != "params":
                    g[k] = v
            for i, p in zip(saved["params"], g["params"]):
                by_index[i] = p
        self.state = {}
        for i, st in sd["state"].items():
            p = by_index[int(i)]
            self.state[p] = {"step": int(round(float(st["step"]))),
                             "exp_avg": st["exp_avg"].to(device=p.device, dtype=torch.float32).contiguous().clone(),
                             "exp_avg_sq": st["exp_avg_sq"].to(device=p.device, dtype=torch.float32).contiguous().clone()}
        GENERATION[0] += 1
