"""Deterministic deformation-field test cases shared by the golden generator
and the parity tests (CPU oracle and GPU)."""
import itertools
import math

import torch

from hashrand import uniform, triangular

PLANE_AXES = list(itertools.combinations(range(4), 2))

CASES = {
    # name: (resolution[x,y,z,t], multires, n_points)
    "tiny": ([8, 6, 5, 10], [1, 2], 777),
    "dynerf": ([64, 64, 64, 150], [1, 2], 1536),   # configs/train/dynerf.yaml:95-101
    "dnerf64": ([12, 9, 7, 25], [1, 2], 900),      # net_width 64 as configs/train/dnerf.yaml:83 (small planes)
    "narrow40": ([8, 6, 5, 10], [1, 2], 333),      # a width that is no multiple of 32, defor_depth 1 (= 0: create_net
                                                   # adds layers from depth 2 on, src/model/deformation.py:72-75)
}
WIDTHS = {"dnerf64": 64, "narrow40": 40}           # net_width of a case; 128 (configs/train/dynerf.yaml:85) otherwise
DEPTHS = {"narrow40": 1}
FEAT = 16
HEAD_DIMS = {"pos": 3, "scales": 3, "rotations": 4, "opacity": 1, "shs": 48}
AABB_MAX = [2.5, 2.0, 1.0]      # configs/train/dynerf.yaml:69-70
AABB_MIN = [-2.5, -2.0, -1.0]


def net_kwargs(case):
    reso, multires, _ = CASES[case]
    return dict(net_width=WIDTHS.get(case, 128), timebase_pe=4, defor_depth=DEPTHS.get(case, 0), posebase_pe=10,
                scale_rotation_pe=2, opacity_pe=2, grid_pe=0, no_grid=False, bounds=1.6,
                kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4,
                                "output_coordinate_dim": FEAT, "resolution": list(reso)},
                multires=list(multires), empty_voxel=False, static_mlp=False)


def make_state(case):
    """state_dict with the reference's 39 keys, filled deterministically."""
    reso, multires, _ = CASES[case]
    WIDTH = WIDTHS.get(case, 128)
    sd = {}
    sd["time_poc"] = torch.tensor([2.0 ** i for i in range(4)])
    sd["pos_poc"] = torch.tensor([2.0 ** i for i in range(10)])
    sd["rotation_scaling_poc"] = torch.tensor([2.0 ** i for i in range(2)])
    sd["opacity_poc"] = torch.tensor([2.0 ** i for i in range(2)])
    g = "deformation_net.grid."
    sd[g + "aabb"] = torch.tensor([AABB_MAX, AABB_MIN], dtype=torch.float32)
    for li, m in enumerate(multires):
        r = [reso[0] * m, reso[1] * m, reso[2] * m, reso[3]]
        for ci, (a, b) in enumerate(PLANE_AXES):
            shape = (1, FEAT, r[b], r[a])
            key = g + f"grids.{li}.{ci}"
            if b == 3:   # time planes: near one (init is exactly one; perturbed so grads are non-trivial)
                sd[key] = 1.0 + triangular(f"{case}/{key}", shape, 0.05)
            else:
                sd[key] = uniform(f"{case}/{key}", shape, 0.1, 0.5)

    def lin(prefix, out_f, in_f):
        bound = math.sqrt(6.0 / (in_f + out_f))
        sd[prefix + ".weight"] = uniform(f"{case}/{prefix}.w", (out_f, in_f), -bound, bound)
        b = 1.0 / math.sqrt(in_f)
        sd[prefix + ".bias"] = uniform(f"{case}/{prefix}.b", (out_f,), -b, b)

    lin("deformation_net.feature_out.0", WIDTH, FEAT * len(multires))
    for h, k in HEAD_DIMS.items():
        lin(f"deformation_net.{h}_deform.1", WIDTH, WIDTH)
        lin(f"deformation_net.{h}_deform.3", k, WIDTH)
    return sd


def make_inputs(case):
    n = CASES[case][2]
    xyz = uniform(f"{case}/xyz", (n, 3), -1.0, 1.0) * torch.tensor([3.0, 2.4, 1.2])  # some outside the AABB
    # exact corners / borders of the box, to hit the clip branches
    xyz[0] = torch.tensor(AABB_MAX)
    xyz[1] = torch.tensor(AABB_MIN)
    xyz[2] = torch.tensor([0.0, 0.0, 0.0])
    scales = triangular(f"{case}/scales", (n, 3), 0.5) - 4.6
    rot = triangular(f"{case}/rot", (n, 4), 1.0)
    opacity = triangular(f"{case}/opacity", (n, 1), 1.5)
    shs = triangular(f"{case}/shs", (n, 16, 3), 0.3)
    tsel = torch.arange(n) % 4
    time = torch.where(tsel == 0, torch.tensor(0.0),
                       torch.where(tsel == 1, torch.tensor(0.37),
                                   torch.where(tsel == 2, torch.tensor(299.0 / 300.0),
                                               uniform(f"{case}/time", (n,), 0.0, 1.0))))
    return xyz, scales, rot, opacity, shs, time.reshape(n, 1).contiguous()


def make_grad_outputs(case):
    n = CASES[case][2]
    return (triangular(f"{case}/g_pts", (n, 3), 1.0), triangular(f"{case}/g_sc", (n, 3), 1.0),
            triangular(f"{case}/g_rot", (n, 4), 1.0), triangular(f"{case}/g_op", (n, 1), 1.0),
            triangular(f"{case}/g_sh", (n, 16, 3), 1.0))


def trainable_keys(sd):
    return [k for k in sd if k.startswith("deformation_net.") and not k.endswith("aabb")]
