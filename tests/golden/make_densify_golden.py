"""Generates tests/golden/densify_small.npz by running the reference's own surgery methods
(src/runner/fdgs_trainer.py:304-492: _densify, _prune, _prune_points with the optimizer plumbing they call) on
the CPU.  Run in the authoring container only: python tests/golden/make_densify_golden.py

What is arranged around the reference code, which is imported as it is:
  * third-party modules its import chain names but this code never calls (tkinter, plyfile, simple_knn,
    diff_gaussian_rasterization, tensorboard, torchvision, cv2) are absent here and registered as empty
    placeholders;
  * the methods spell device="cuda"; the name `torch` inside the two reference modules involved is bound to a
    proxy that forwards everything to torch and rewrites that one keyword to "cpu";
  * torch.normal(mean, std) (fdgs_trainer.py:404) is forwarded as mean + std * u with u = torch.randn(...), which
    is what torch.normal computes, so that the unit samples u can be stored in the fixture.
The trainer object is created without its constructor (which opens a scene); only the attributes the surgery
reads are set.  Inputs regenerate from tests/densify_cases.py; the fixture holds the outputs and the samples."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import densify_cases as dc  # noqa: E402

CASES = {"densify": dict(n=600, seed=101, max_grad=2e-4, extent=5.0),
         "prune": dict(n=500, seed=102, min_opacity=0.2, extent=5.0, max_screen_size=20),
         "prune_no_size": dict(n=500, seed=103, min_opacity=0.2, extent=5.0, max_screen_size=None),
         "prune_points": dict(n=300, seed=104)}
PERCENT_DENSE = 0.01
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "scaling": "_scaling", "rotation": "_rotation"}


class CpuTorch:
    """torch with device="cuda" rewritten to "cpu" and normal() expressed through recorded unit samples."""

    def __init__(self):
        self.unit_samples = []

    def normal(self, mean, std):
        u = torch.randn(std.shape)
        self.unit_samples.append(u.clone())
        return mean + std * u

    def __getattr__(self, name):
        obj = getattr(torch, name)
        if not callable(obj) or isinstance(obj, type):
            return obj

        def call(*a, **k):
            if k.get("device") == "cuda":
                k["device"] = "cpu"
            return obj(*a, **k)
        return call


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules.setdefault(name, m)
        return m
    stub("tkinter", W="w")
    stub("plyfile", PlyData=object, PlyElement=object)
    stub("simple_knn", _C=stub("simple_knn._C", distCUDA2=None))
    stub("diff_gaussian_rasterization", GaussianRasterizationSettings=object, GaussianRasterizer=object)
    stub("torch.utils.tensorboard", SummaryWriter=object)
    stub("torchvision"); stub("torchvision.transforms"); stub("cv2")
    sys.path.insert(0, "/root/reference")
    import src.runner.fdgs_trainer as ft
    import src.utils.graphics_utils as gu
    from src.model.gaussian_model import GaussianModel
    proxy = CpuTorch()
    ft.torch = proxy
    gu.torch = proxy           # build_rotation (imported by the trainer from here) allocates on "cuda" too
    return ft, GaussianModel, proxy


def make_trainer(ft, GaussianModel, st):
    tr = object.__new__(ft.Gaussian4DTrainer)
    g = GaussianModel(sh_degree=3, percent_dense=PERCENT_DENSE)
    groups = []
    for k in dc.NAMES:
        p = torch.nn.Parameter(torch.from_numpy(st["params"][k].copy()))
        setattr(g, ATTR[k], p)
        groups.append({"params": [p], "lr": 1e-3, "name": k})
    # a multi-tensor group like the reference's "deformation" group: the surgery must skip it
    groups.insert(1, {"params": [torch.nn.Parameter(torch.ones(2, 2)), torch.nn.Parameter(torch.ones(3))],
                      "lr": 1e-3, "name": "deformation"})
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    for grp in groups:
        if grp["name"] in dc.NAMES:
            k = grp["name"]
            opt.state[grp["params"][0]] = {"step": torch.tensor(3.0),
                                           "exp_avg": torch.from_numpy(st["exp_avg"][k].copy()),
                                           "exp_avg_sq": torch.from_numpy(st["exp_avg_sq"][k].copy())}
    g.xyz_gradient_accum = torch.from_numpy(st["xyz_gradient_accum"].copy())
    g.denom = torch.from_numpy(st["denom"].copy())
    g.max_radii2D = torch.from_numpy(st["max_radii2D"].copy())
    tr.gaussians = g
    tr.deforms = types.SimpleNamespace(deformation_table=torch.from_numpy(st["deformation_table"].copy()),
                                       deformation_accum=torch.from_numpy(st["deformation_accum"].copy()))
    tr.optimizer = opt
    return tr


def dump(out, tag, tr):
    g = tr.gaussians
    for k in dc.NAMES:
        p = getattr(g, ATTR[k])
        out["%s_%s" % (tag, k)] = p.detach().numpy()
        entry = tr.optimizer.state[p]
        out["%s_%s_exp_avg" % (tag, k)] = entry["exp_avg"].numpy()
        out["%s_%s_exp_avg_sq" % (tag, k)] = entry["exp_avg_sq"].numpy()
    out[tag + "_xyz_gradient_accum"] = g.xyz_gradient_accum.numpy()
    out[tag + "_denom"] = g.denom.numpy()
    out[tag + "_max_radii2D"] = g.max_radii2D.numpy()
    out[tag + "_deformation_table"] = tr.deforms.deformation_table.numpy()
    out[tag + "_deformation_accum"] = tr.deforms.deformation_accum.numpy()


def main():
    ft, GaussianModel, proxy = import_reference()
    torch.manual_seed(6666)
    out = {}
    c = CASES["densify"]
    tr = make_trainer(ft, GaussianModel, dc.make_state(c["n"], seed=c["seed"]))
    tr._densify(c["max_grad"], c["extent"])
    assert len(proxy.unit_samples) == 1
    out["densify_unit_noise"] = proxy.unit_samples[0].numpy()
    dump(out, "densify", tr)
    for tag in ("prune", "prune_no_size"):
        c = CASES[tag]
        tr = make_trainer(ft, GaussianModel, dc.make_state(c["n"], seed=c["seed"]))
        tr._prune(c["min_opacity"], c["extent"], c["max_screen_size"])
        dump(out, tag, tr)
    c = CASES["prune_points"]
    tr = make_trainer(ft, GaussianModel, dc.make_state(c["n"], seed=c["seed"]))
    mask = torch.arange(c["n"]) % 3 == 1
    tr._prune_points(mask)
    dump(out, "prune_points", tr)
    path = os.path.join(HERE, "densify_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; rows after densify:", out["densify_xyz"].shape[0],
          "children:", out["densify_unit_noise"].shape[0], "after prune:", out["prune_xyz"].shape[0],
          out["prune_no_size_xyz"].shape[0])


if __name__ == "__main__":
    main()
