"""Generate tests/golden/reg_<case>.npz: the reference's grid regulariser on the golden state dicts.

Run in the authoring container only (it needs /root/reference):
    python tests/golden/make_reg_golden.py

`DeformationWrapper.compute_regulation` (src/model/deformation_wrapper.py:52-92, with
`compute_plane_smoothness` of src/utils/plane_utils.py:3-9) is called as the reference wrote it; only the
object it runs on is built by hand, because the wrapper's constructor moves the field to "cuda"
(:22) and this container has no GPU: a bare instance gets the CPU DeformationFields as its
`deformation_fields`, which is all these methods read.  Stored: the loss for the weights of
configs/train/dynerf.yaml:53-55 and the gradients of the twelve planes (all of them for the small
case, every 997th element plus sum and absolute sum for the DyNeRF-sized one).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import deform_cases as dc  # noqa: E402
from make_deform_golden import import_reference  # noqa: E402

WEIGHTS = dict(time_smoothness_weight=0.001, l1_time_planes_weight=0.0001, plane_tv_weight=0.0002)


def main():
    deformation, _ = import_reference()
    from src.model.deformation_wrapper import DeformationWrapper
    for case in dc.CASES:
        net = deformation.DeformationFields(dc.net_kwargs(case))
        net.load_state_dict(dc.make_state(case))
        wrap = DeformationWrapper.__new__(DeformationWrapper)
        torch.nn.Module.__init__(wrap)
        wrap.deformation_fields = net
        loss = wrap.compute_regulation(**WEIGHTS)
        planes = [p for gp in net.deformation_net.grid.grids for p in gp]
        grads = torch.autograd.grad(loss, planes)
        out = {"loss": np.float64(loss.item()),
               "parts": np.array([float(wrap.plane_regulation()), float(wrap.time_regulation()),
                                  float(wrap.l1_regulation())], dtype=np.float64)}
        for i, g in enumerate(grads):
            g = g.detach().reshape(-1)
            if case == "tiny":
                out[f"grad_{i}"] = g.numpy()
            else:
                out[f"grad_{i}_sample"] = g[::997].numpy()
                out[f"grad_{i}_sums"] = np.array([g.double().sum().item(), g.double().abs().sum().item()])
        path = os.path.join(HERE, f"reg_{case}.npz")
        np.savez_compressed(path, **out)
        print(case, "loss", float(loss), "->", path)


if __name__ == "__main__":
    main()
