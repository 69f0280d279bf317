"""Generates tests/golden/schedule.json with the reference's own get_expon_lr_func (src/utils/general_utils.py)
and Gaussian4DTrainer._update_learning_rate (src/runner/fdgs_trainer.py:275-302), configs/train/dynerf.yaml values.
Run in the authoring container only: python tests/golden/make_schedule_golden.py
Absent third-party modules named by the trainer's import chain are empty placeholders (never called here)."""
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ITERS = [0, 1, 2, 10, 100, 999, 3000, 10000, 14000, 20000, 25000]
SCALE = 4.5


def main():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules.setdefault(name, m)
        return m
    stub("tkinter", W="w"); stub("plyfile", PlyData=object, PlyElement=object)
    stub("simple_knn", _C=stub("simple_knn._C", distCUDA2=None))
    stub("diff_gaussian_rasterization", GaussianRasterizationSettings=object, GaussianRasterizer=object)
    stub("torch.utils.tensorboard", SummaryWriter=object)
    stub("torchvision"); stub("torchvision.transforms"); stub("cv2")
    sys.path.insert(0, "/root/reference")
    import src.runner.fdgs_trainer as ft
    import yaml
    cfg = yaml.safe_load(open("/root/reference/configs/train/dynerf.yaml"))

    def find(d, key):
        if isinstance(d, dict):
            if key in d:
                return d[key]
            for v in d.values():
                r = find(v, key)
                if r is not None:
                    return r
        return None
    opt = find(cfg, "optimizer_kwargs")
    tr = object.__new__(ft.Gaussian4DTrainer)
    tr.spatial_lr_scale = SCALE
    for k in ("position_lr_init", "position_lr_final", "position_lr_delay_mult", "position_lr_max_steps",
              "deformation_lr_init", "deformation_lr_final", "deformation_lr_delay_mult", "grid_lr_init",
              "grid_lr_final"):
        setattr(tr, k, opt[k])
    names = ["xyz", "deformation", "grid", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    groups = [{"params": [torch.nn.Parameter(torch.zeros(1))], "lr": 0.123, "name": n} for n in names]
    tr.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    # the three schedule objects exactly as _training_setup builds them (fdgs_trainer.py:275-286)
    gl = ft.get_expon_lr_func
    tr.xyz_scheduler_args = gl(lr_init=tr.position_lr_init * SCALE, lr_final=tr.position_lr_final * SCALE,
                               lr_delay_mult=tr.position_lr_delay_mult, max_steps=tr.position_lr_max_steps)
    tr.deformation_scheduler_args = gl(lr_init=tr.deformation_lr_init * SCALE, lr_final=tr.deformation_lr_final * SCALE,
                                       lr_delay_mult=tr.deformation_lr_delay_mult, max_steps=tr.position_lr_max_steps)
    tr.grid_scheduler_args = gl(lr_init=tr.grid_lr_init * SCALE, lr_final=tr.grid_lr_final * SCALE,
                                lr_delay_mult=tr.deformation_lr_delay_mult, max_steps=tr.position_lr_max_steps)
    out = {"iters": ITERS, "spatial_lr_scale": SCALE, "lr": {}}
    for it in ITERS:
        tr._update_learning_rate(it)
        out["lr"][str(it)] = {g["name"]: float(g["lr"]) for g in tr.optimizer.param_groups}
    # a delayed schedule (lr_delay_steps > 0), not used by the shipped configs but part of the function
    f = gl(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.05, max_steps=3000)
    out["delayed"] = {str(it): float(f(it)) for it in [-1, 0, 1, 100, 499, 500, 1500, 3000, 4000]}
    out["zero"] = float(gl(lr_init=0.0, lr_final=0.0)(10))
    path = os.path.join(HERE, "schedule.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
