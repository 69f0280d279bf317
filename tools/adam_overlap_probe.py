"""Can the HexPlane feature gather of the NEXT view hide behind the optimizer step?  Times, at 300k Gaussians:
  (a) one Adam launch over every tensor, then the feature gather, on one stream;
  (b) Adam over the tensors the gather reads (planes, positions), then the gather on a second stream beside Adam
      over the rest (SH coefficients, scales, rotations, opacities: 85 % of the bytes) on the first.
    python tools/adam_overlap_probe.py
"""
import sys

import torch

sys.path.insert(0, "4d-gaussian-splatting_amd")
from fdgs import scene_synth as ss                       # noqa: E402
from fdgs.adam import FusedAdam                          # noqa: E402
from fdgs.deformation import DeformationFields           # noqa: E402

dev = torch.device("cuda:0")
N = 300000


def main():
    g = ss.GaussianParams(N, device=dev)
    net = ss.perturb_deformation_(DeformationFields(ss.DYNERF_NET_KWARGS)).to(dev)
    net.deformation_net.set_aabb(list(ss.SCENE_MAX), list(ss.SCENE_MIN))
    grid = net.deformation_net.grid
    first = [g._xyz] + list(net.get_grid_parameters()) + list(net.get_mlp_parameters())
    rest = [g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation]
    for p in first + rest:
        p.grad = torch.randn_like(p) * 1e-3
    opt_all = FusedAdam([{"params": first + rest, "lr": 1e-6}], eps=1e-15)
    opt_first = FusedAdam([{"params": first, "lr": 1e-6}], eps=1e-15)
    opt_rest = FusedAdam([{"params": rest, "lr": 1e-6}], eps=1e-15)
    side = torch.cuda.Stream()
    ev, back = torch.cuda.Event(), torch.cuda.Event()
    xyz = g._xyz.detach()

    def serial():
        opt_all.step()
        grid.get_density(xyz, 0.37)

    def adam_only():
        opt_all.step()

    def features_only():
        grid.get_density(xyz, 0.37)

    def overlapped(features_first):
        cur = torch.cuda.current_stream()
        opt_first.step()
        ev.record(cur)
        if not features_first:
            opt_rest.step()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            grid.get_density(xyz, 0.37)
            back.record(side)
        if features_first:
            opt_rest.step()
        cur.wait_event(back)

    def timed(fn, n=40):
        for _ in range(60):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    for name, fn in (("Adam, all tensors", adam_only), ("feature gather", features_only),
                     ("Adam then gather, one stream", serial),
                     ("gather submitted before the rest of Adam", lambda: overlapped(True)),
                     ("gather submitted after the rest of Adam", lambda: overlapped(False)),
                     ("Adam then gather, one stream", serial)):
        print("%-44s %.4f ms" % (name, timed(fn)), flush=True)


if __name__ == "__main__":
    main()
