"""Seeded per-Gaussian state for the densification / pruning tests (numpy, CPU)."""
import numpy as np

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,)}


def make_state(n, seed=0, moments=True, zero_denom_every=7):
    """Scales around exp(-3) +- so that, with percent_dense * extent = 0.05, both small and large Gaussians
    occur; gradient statistics around 2e-4 so that a threshold of 2e-4 selects about half; some denom == 0."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    params = {k: rng.standard_normal((n,) + SHAPES[k]).astype(f32) for k in NAMES}
    params["scaling"] = (rng.standard_normal((n, 3)) * 0.8 - 3.2).astype(f32)
    params["opacity"] = (rng.standard_normal((n, 1)) * 2.5 - 1.0).astype(f32)
    st = {"params": params}
    if moments:
        st["exp_avg"] = {k: rng.standard_normal((n,) + SHAPES[k]).astype(f32) for k in NAMES}
        st["exp_avg_sq"] = {k: (rng.random((n,) + SHAPES[k]) + 0.01).astype(f32) for k in NAMES}
    else:
        st["exp_avg"] = None
        st["exp_avg_sq"] = None
    denom = rng.integers(1, 40, (n, 1)).astype(f32)
    if zero_denom_every:
        denom[::zero_denom_every] = 0
    accum = (rng.random((n, 1)) * 4e-4).astype(f32) * denom
    st["xyz_gradient_accum"] = accum
    st["denom"] = denom
    st["max_radii2D"] = (rng.random(n) * 40).astype(f32)
    st["deformation_table"] = rng.random(n) > 0.3
    st["deformation_accum"] = rng.standard_normal((n, 3)).astype(f32)
    return st


def copy_state(st):
    out = {}
    for k, v in st.items():
        if isinstance(v, dict):
            out[k] = {kk: vv.copy() for kk, vv in v.items()}
        elif v is None:
            out[k] = None
        else:
            out[k] = v.copy()
    return out
