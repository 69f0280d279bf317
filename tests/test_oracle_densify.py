"""CPU checks of oracle/densify_ref.py, the restatement of the reference's densification / pruning surgery
(src/runner/fdgs_trainer.py:304-492): hand-derived cases (the row order the clone -> split -> prune sequence
must produce, the optimizer moments of added rows, the statistics reset) and, at the end of the file, the outputs
of the reference's own methods."""
import numpy as np

import densify_cases as dc
from oracle import densify_ref as dr


def _tiny():
    """Six Gaussians: rows 1 and 4 small with a large gradient (clone), rows 2 and 5 large with a large
    gradient (split), row 0 quiet, row 3 with denom 0 (gradient NaN -> 0)."""
    st = dc.make_state(6, seed=3, zero_denom_every=0)
    st["denom"] = np.array([[4], [2], [5], [0], [1], [3]], np.float32)
    st["xyz_gradient_accum"] = np.array([[1e-5], [1.0], [4.0], [0.0], [3.0], [4.0]], np.float32)
    for k in dc.NAMES:      # row id in the first component of every array
        st["params"][k].reshape(6, -1)[:, 0] = np.arange(6)
    sc = np.full((6, 3), np.log(0.01), np.float32)
    sc[2] = np.log(0.2); sc[5] = np.log(0.3)
    st["params"]["scaling"] = sc
    return st


def test_densify_row_order_and_moments():
    st = _tiny()
    before = dc.copy_state(st)
    noise = np.zeros((4, 3), np.float32)
    clone_sel, split_sel = dr.densify(st, 0.5, 5.0, 0.01, lambda count: noise[:2 * count])
    assert clone_sel.tolist() == [False, True, False, False, True, False]
    assert split_sel[:6].tolist() == [False, False, True, False, False, True]
    # survivors 0 1 3 4 | clones 1 4 | first children 2 5 | second children 2 5
    order = [0, 1, 3, 4, 1, 4, 2, 5, 2, 5]
    for k in ("f_dc", "f_rest", "opacity", "rotation"):
        assert np.array_equal(st["params"][k], before["params"][k][order]), k
    # zero noise: the children sit on the parent, scale divided by 1.6
    assert np.array_equal(st["params"]["xyz"], before["params"]["xyz"][order])
    want = before["params"]["scaling"][order].copy()
    want[6:] = np.log(np.exp(want[6:]) / np.float32(1.6))
    assert np.allclose(st["params"]["scaling"], want, rtol=0, atol=1e-6)
    for m in ("exp_avg", "exp_avg_sq"):
        for k in dc.NAMES:
            assert np.array_equal(st[m][k][:4], before[m][k][[0, 1, 3, 4]])
            assert not st[m][k][4:].any()
    assert st["xyz_gradient_accum"].shape == (10, 1) and not st["xyz_gradient_accum"].any()
    assert st["denom"].shape == (10, 1) and not st["denom"].any()
    assert st["max_radii2D"].shape == (10,) and not st["max_radii2D"].any()
    assert st["deformation_accum"].shape == (10, 3) and not st["deformation_accum"].any()
    assert np.array_equal(st["deformation_table"], before["deformation_table"][order])


def test_split_children_follow_rotation_and_scale():
    st = _tiny()
    st["params"]["rotation"][:] = 0
    st["params"]["rotation"][:, 0] = 2.0            # identity rotation, un-normalised
    st["params"]["rotation"][5] = [1.0, 0.0, 0.0, 1.0]   # 90 degrees about z: x -> y
    before = dc.copy_state(st)
    noise = np.zeros((4, 3), np.float32)
    noise[0] = [1, 0, 0]        # first child of row 2
    noise[1] = [1, 0, 0]        # first child of row 5
    noise[3] = [0, 0, -2]       # second child of row 5
    dr.densify(st, 0.5, 5.0, 0.01, lambda count: noise)
    x = st["params"]["xyz"]
    p2, p5 = before["params"]["xyz"][2], before["params"]["xyz"][5]
    assert np.allclose(x[6], p2 + [0.2, 0, 0], atol=1e-6)
    assert np.allclose(x[7], p5 + [0, 0.3, 0], atol=1e-6)
    assert np.allclose(x[8], p2, atol=1e-6)
    assert np.allclose(x[9], p5 + [0, 0, -0.6], atol=1e-6)


def test_prune_and_reset():
    st = dc.make_state(50, seed=5)
    before = dc.copy_state(st)
    mask = dr.prune(st, 0.2, 5.0, None)
    sig = 1 / (1 + np.exp(-before["params"]["opacity"].reshape(-1).astype(np.float64)))
    assert np.array_equal(mask, sig.astype(np.float32) < np.float32(0.2)) or abs((mask != (sig < 0.2)).sum()) <= 1
    keep = ~mask
    for k in dc.NAMES:
        assert np.array_equal(st["params"][k], before["params"][k][keep])
        assert np.array_equal(st["exp_avg"][k], before["exp_avg"][k][keep])
    assert np.array_equal(st["max_radii2D"], before["max_radii2D"][keep])        # statistics are carried
    assert np.array_equal(st["deformation_accum"], before["deformation_accum"][keep])
    st2 = dc.copy_state(before)
    mask2 = dr.prune(st2, 0.2, 5.0, 20)
    assert mask2.sum() > mask.sum() and (mask2 | ~mask).all()
    assert (before["max_radii2D"][~mask2] <= 20).all()
    dr.reset_opacity(st2)
    o = 1 / (1 + np.exp(-st2["params"]["opacity"]))
    assert o.max() <= 0.01 + 1e-6
    assert not st2["exp_avg"]["opacity"].any() and not st2["exp_avg_sq"]["opacity"].any()
    assert st2["exp_avg"]["xyz"].any()


# ---- against the reference's own methods ---------------------------------------------------------------------
# tests/golden/densify_small.npz holds what Gaussian4DTrainer._densify / _prune / _prune_points produced on the
# CPU for the seeded states below (tests/golden/make_densify_golden.py); its cases are repeated here as data.
import os

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "densify_small.npz"))
GOLD_CASES = {"densify": dict(n=600, seed=101, max_grad=2e-4, extent=5.0),
              "prune": dict(n=500, seed=102, min_opacity=0.2, extent=5.0, max_screen_size=20),
              "prune_no_size": dict(n=500, seed=103, min_opacity=0.2, extent=5.0, max_screen_size=None),
              "prune_points": dict(n=300, seed=104)}


def run_oracle_case(tag):
    c = GOLD_CASES[tag]
    st = dc.make_state(c["n"], seed=c["seed"])
    if tag == "densify":
        noise = GOLD["densify_unit_noise"]
        dr.densify(st, c["max_grad"], c["extent"], 0.01, lambda count: noise[:2 * count])
    elif tag.startswith("prune_points"):
        dr.prune_points(st, np.arange(c["n"]) % 3 == 1)
    else:
        dr.prune(st, c["min_opacity"], c["extent"], c["max_screen_size"])
    return st


def check_against_golden(tag, get, child_rows=None, tol=1e-6):
    """get(kind, name) -> numpy array of the result; kind in {"param", "exp_avg", "exp_avg_sq", "stat"}."""
    for k in dc.NAMES:
        want = GOLD["%s_%s" % (tag, k)]
        got = get("param", k)
        assert got.shape == want.shape, (tag, k, got.shape, want.shape)
        if child_rows is not None and k in ("xyz", "scaling"):
            assert np.array_equal(got[:child_rows], want[:child_rows]), (tag, k)
            err = np.abs(got[child_rows:] - want[child_rows:]).max()
            assert err <= tol * max(np.abs(want[child_rows:]).max(), 1.0), (tag, k, err)
        else:
            assert np.array_equal(got, want), (tag, k)
        for m in ("exp_avg", "exp_avg_sq"):
            assert np.array_equal(get(m, k), GOLD["%s_%s_%s" % (tag, k, m)]), (tag, k, m)
    for k in ("xyz_gradient_accum", "denom", "max_radii2D", "deformation_table", "deformation_accum"):
        want = GOLD["%s_%s" % (tag, k)]
        got = get("stat", k)
        assert got.shape == want.shape and np.array_equal(got, want), (tag, k)


def test_restatement_matches_reference_methods():
    for tag in GOLD_CASES:
        st = run_oracle_case(tag)
        children = GOLD["densify_unit_noise"].shape[0] if tag == "densify" else 0

        def get(kind, name):
            return st["params"][name] if kind == "param" else (st[name] if kind == "stat" else st[kind][name])
        rows = st["params"]["xyz"].shape[0]
        check_against_golden(tag, get, child_rows=rows - children if children else None)
    assert GOLD["densify_unit_noise"].shape[0] > 100 and GOLD["densify_xyz"].shape[0] > 700
