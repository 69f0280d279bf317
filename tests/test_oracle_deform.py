"""The CPU restatement oracle/deform_ref.py against golden vectors captured
from the reference's src/model/deformation.py + hexplane.py."""
import os

import numpy as np
import pytest
import torch

import deform_cases as dc
from oracle import deform_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("case", list(dc.CASES))
def test_oracle_matches_reference_golden(case):
    gold = np.load(os.path.join(GOLD, f"deform_{case}.npz"))
    sd = dc.make_state(case)
    assert sorted(gold["state_keys"].tolist()) == sorted(sd.keys())
    tkeys = dc.trainable_keys(sd)
    for k in tkeys:
        sd[k].requires_grad_(True)
    inputs = [t.clone().requires_grad_(i < 5) for i, t in enumerate(dc.make_inputs(case))]
    outs = deform_ref.deform_forward(sd, *inputs)
    for name, t in zip(("pts", "scales", "rotations", "opacity", "shs"), outs):
        ref = torch.from_numpy(gold["out_" + name])
        assert t.shape == ref.shape
        assert float((t - ref).abs().max()) < 2e-5, name
    grads = torch.autograd.grad(outs, inputs[:5] + [sd[k] for k in tkeys], dc.make_grad_outputs(case))
    for name, g in zip(("xyz", "scales", "rotations", "opacity", "shs"), grads[:5]):
        assert _rel(g, torch.from_numpy(gold["gin_" + name])) < 1e-4, name
    for k, g in zip(tkeys, grads[5:]):
        assert _rel(g, torch.from_numpy(gold["gp_" + k])) < 1e-4, k


def test_single_point_shape():
    """N == 1 survives (the reference's squeeze() at hexplane.py:49 still yields [1,32])."""
    sd = dc.make_state("tiny")
    ins = [t[:1].clone() for t in dc.make_inputs("tiny")]
    outs = deform_ref.deform_forward(sd, *ins)
    assert [tuple(o.shape) for o in outs] == [(1, 3), (1, 3), (1, 4), (1, 1), (1, 16, 3)]


REG_WEIGHTS = (0.001, 0.0001, 0.0002)     # time_smoothness, l1_time_planes, plane_tv (configs/train/dynerf.yaml:53-55)


def check_regulation_against_golden(case, loss, grads):
    """loss: float; grads: the twelve plane gradients in level-major order (CPU tensors)."""
    gold = np.load(os.path.join(GOLD, f"reg_{case}.npz"))
    assert abs(loss - float(gold["loss"])) <= 2e-6 * abs(float(gold["loss"]))
    for i, g in enumerate(grads):
        g = g.reshape(-1)
        if case == "tiny":
            ref = torch.from_numpy(gold[f"grad_{i}"])
            assert float((g - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        else:
            ref = torch.from_numpy(gold[f"grad_{i}_sample"])
            assert float((g[::997] - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
            s, sa = gold[f"grad_{i}_sums"]
            assert abs(float(g.double().abs().sum()) - sa) <= 1e-5 * sa
            assert abs(float(g.double().sum()) - s) <= 1e-5 * sa


@pytest.mark.parametrize("case", ["tiny", "dynerf"])      # the cases tests/golden/reg_*.npz hold
def test_regulation_oracle_matches_reference_golden(case):
    """oracle compute_regulation against the values the reference's DeformationWrapper.compute_regulation
    produced (tests/golden/make_reg_golden.py)."""
    sd = {k: (v.clone().requires_grad_(True) if ".grids." in k else v) for k, v in dc.make_state(case).items()}
    loss = deform_ref.compute_regulation(sd, *REG_WEIGHTS)
    keys = [f"deformation_net.grid.grids.{l}.{ci}" for l in range(2) for ci in range(6)]
    grads = torch.autograd.grad(loss, [sd[k] for k in keys])
    check_regulation_against_golden(case, float(loss), [g.detach() for g in grads])
