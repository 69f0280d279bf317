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
