"""Per-iteration schedules of the reference's trainer: learning rates of the xyz / deformation / grid parameter
groups (src/runner/fdgs_trainer.py:275-302 with get_expon_lr_func, src/utils/general_utils.py:52-84) and the
densification / pruning thresholds of the fine stage (fdgs_trainer.py:184-189).  Host arithmetic in double, as
the reference's Python does it; nothing here touches the GPU."""
import math


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation from lr_init (step 0) to lr_final (step max_steps), optionally eased in over
    lr_delay_steps by a sine ramp that starts at lr_delay_mult (general_utils.py:52-84)."""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(
                0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        else:
            delay_rate = 1.0
        t = min(max(step / max_steps, 0.0), 1.0)
        return delay_rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return helper


# configs/train/dynerf.yaml:23-52
DYNERF = dict(position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
              position_lr_max_steps=20000, deformation_lr_init=0.00016, deformation_lr_final=0.000016,
              deformation_lr_delay_mult=0.01, grid_lr_init=0.0016, grid_lr_final=0.00016)


class LearningRates:
    """update(optimizer, iteration) rewrites param_group["lr"] by group name exactly as
    Gaussian4DTrainer._update_learning_rate (fdgs_trainer.py:288-302): "xyz", any name containing "grid",
    "deformation"; the other groups keep their constant rates."""

    def __init__(self, spatial_lr_scale=1.0, **kw):
        c = dict(DYNERF, **kw)
        s = spatial_lr_scale
        steps = c["position_lr_max_steps"]
        self.xyz = get_expon_lr_func(c["position_lr_init"] * s, c["position_lr_final"] * s,
                                     lr_delay_mult=c["position_lr_delay_mult"], max_steps=steps)
        self.deformation = get_expon_lr_func(c["deformation_lr_init"] * s, c["deformation_lr_final"] * s,
                                             lr_delay_mult=c["deformation_lr_delay_mult"], max_steps=steps)
        self.grid = get_expon_lr_func(c["grid_lr_init"] * s, c["grid_lr_final"] * s,
                                      lr_delay_mult=c["deformation_lr_delay_mult"], max_steps=steps)

    def update(self, optimizer, iteration):
        for group in optimizer.param_groups:
            name = group.get("name", "")
            if name == "xyz":
                group["lr"] = self.xyz(iteration)
            if "grid" in name:
                group["lr"] = self.grid(iteration)
            elif name == "deformation":
                group["lr"] = self.deformation(iteration)


def fine_thresholds(iteration, densify_until_iter, opacity_threshold_fine_init, opacity_threshold_fine_after,
                    densify_grad_threshold_fine_init, densify_grad_threshold_after):
    """(opacity_threshold, densify_threshold) of the fine stage at `iteration` (fdgs_trainer.py:188-189): both
    move linearly from their initial to their final value over densify_until_iter iterations."""
    o = opacity_threshold_fine_init - iteration * (opacity_threshold_fine_init - opacity_threshold_fine_after) / \
        densify_until_iter
    d = densify_grad_threshold_fine_init - iteration * (densify_grad_threshold_fine_init -
                                                        densify_grad_threshold_after) / densify_until_iter
    return o, d
