"""Generate tests/golden/deform_<case>.npz from the reference's own modules.

Run in the authoring container only (it needs /root/reference):
    python tests/golden/make_deform_golden.py [case ...]

The reference modules are imported as they are; three third-party modules that
its import chain names but this path never calls (tkinter, plyfile,
simple_knn._C) are absent here and are registered as empty placeholders so
that `import src.model.deformation` resolves.  Inputs and parameters come from
tests/deform_cases.py (exact-integer hash), so the fixture stores only what the
reference computed: the five outputs, the input gradients and the parameter
gradients for a fixed set of output cotangents.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import deform_cases as dc  # noqa: E402


def import_reference():
    tk = types.ModuleType("tkinter"); tk.W = "w"
    ply = types.ModuleType("plyfile"); ply.PlyData = object; ply.PlyElement = object
    knn = types.ModuleType("simple_knn"); knn_c = types.ModuleType("simple_knn._C")
    knn_c.distCUDA2 = None; knn._C = knn_c
    for name, mod in (("tkinter", tk), ("plyfile", ply), ("simple_knn", knn), ("simple_knn._C", knn_c)):
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, "/root/reference")
    import src.model.deformation as deformation
    import src.model.hexplane as hexplane
    # device default "cuda" -> "cpu" (deformation.py:22): there is no GPU in this container
    d = list(deformation.DeformationNet.__init__.__defaults__)
    d[-1] = "cpu"
    deformation.DeformationNet.__init__.__defaults__ = tuple(d)
    return deformation, hexplane


def main():
    deformation, _ = import_reference()
    torch.set_num_threads(8)
    only = sys.argv[1:]                      # case names; none = all
    for case in dc.CASES:
        if only and case not in only:
            continue
        net = deformation.DeformationFields(dc.net_kwargs(case))
        sd = dc.make_state(case)
        ref_keys = list(net.state_dict().keys())
        assert sorted(ref_keys) == sorted(sd.keys()), (sorted(ref_keys), sorted(sd.keys()))
        for k, v in net.state_dict().items():
            assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
        net.load_state_dict(sd)
        net = net.double() if False else net
        inputs = [t.clone().requires_grad_(i < 5) for i, t in enumerate(dc.make_inputs(case))]
        outs = net(*inputs)
        params = dict(net.named_parameters())
        tkeys = dc.trainable_keys(sd)
        gouts = dc.make_grad_outputs(case)
        grads = torch.autograd.grad(outs, inputs[:5] + [params[k] for k in tkeys], gouts)
        blob = {"state_keys": np.array(ref_keys)}
        for name, t in zip(("pts", "scales", "rotations", "opacity", "shs"), outs):
            blob["out_" + name] = t.detach().numpy()
        for name, t in zip(("xyz", "scales", "rotations", "opacity", "shs"), grads[:5]):
            blob["gin_" + name] = t.numpy()
        for k, t in zip(tkeys, grads[5:]):
            blob["gp_" + k] = t.numpy()
        path = os.path.join(HERE, f"deform_{case}.npz")
        np.savez_compressed(path, **blob)
        print(case, "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
