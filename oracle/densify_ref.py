"""TEST INFRASTRUCTURE ONLY - CPU restatement (numpy) of the reference's densification / pruning surgery.

It follows the reference step by step, in the reference's own order (clone, postfix, split, postfix, prune),
with boolean masks and concatenations as the reference does them, so that the one-pass row map of the HIP
path can be checked against the sequence it replaces:

  densify            src/runner/fdgs_trainer.py:487-492
  densify_and_clone  src/runner/fdgs_trainer.py:426-449
  densify_and_split  src/runner/fdgs_trainer.py:388-424   (build_rotation: src/utils/general_utils.py:101-124)
  postfix            src/runner/fdgs_trainer.py:451-475   (cat_tensors_to_optimizer :341-366)
  prune_points       src/runner/fdgs_trainer.py:368-386   (prune_optimizer :319-339)
  prune              src/runner/fdgs_trainer.py:477-484
  reset_opacity      src/model/gaussian_model.py:173-176  (replace_tensor_to_optimizer: fdgs_trainer.py:304-317)

Pinned by tests/golden/densify_small.npz: the outputs of the reference's own Gaussian4DTrainer._densify /
_prune / _prune_points, run on the CPU by tests/golden/make_densify_golden.py (tests/test_oracle_densify.py).
The random draw of the split (torch.normal(mean=0, std), fdgs_trainer.py:404) is std * N(0,1): the unit
normal samples are an argument so that both sides of a parity test use the same ones.  reset_opacity has no
reference output: the reference's method calls one that does not exist (gaussian_model.py:175).

State: a dict with
  params  {name: [N, ...] float32}  names xyz, f_dc, f_rest, opacity, scaling, rotation
  exp_avg, exp_avg_sq  {name: array like the parameter}  (Adam moments; may be absent -> None)
  xyz_gradient_accum [N,1], denom [N,1], max_radii2D [N], deformation_table [N] bool, deformation_accum [N,3]
"""
import numpy as np

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
f32 = np.float32


def build_rotation(r):
    """src/utils/general_utils.py:101-124"""
    r = r.astype(f32)
    norm = np.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = np.zeros((q.shape[0], 3, 3), f32)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    two = f32(2)
    R[:, 0, 0] = 1 - two * (y * y + z * z)
    R[:, 0, 1] = two * (x * y - w * z)
    R[:, 0, 2] = two * (x * z + w * y)
    R[:, 1, 0] = two * (x * y + w * z)
    R[:, 1, 1] = 1 - two * (x * x + z * z)
    R[:, 1, 2] = two * (y * z - w * x)
    R[:, 2, 0] = two * (x * z - w * y)
    R[:, 2, 1] = two * (y * z + w * x)
    R[:, 2, 2] = 1 - two * (x * x + y * y)
    return R


def _get_scaling(st):
    return np.exp(st["params"]["scaling"].astype(f32))


def _cat_tensors_to_optimizer(st, d):
    """fdgs_trainer.py:341-366: parameters extended, moments extended with zeros"""
    for name in NAMES:
        ext = d[name].astype(f32)
        st["params"][name] = np.concatenate([st["params"][name], ext], 0)
        for m in ("exp_avg", "exp_avg_sq"):
            if st.get(m) is not None:
                st[m][name] = np.concatenate([st[m][name], np.zeros_like(ext)], 0)


def _densification_postfix(st, d, new_table):
    """fdgs_trainer.py:451-475: statistics start again from zero at the new size"""
    _cat_tensors_to_optimizer(st, d)
    n = st["params"]["xyz"].shape[0]
    st["xyz_gradient_accum"] = np.zeros((n, 1), f32)
    st["max_radii2D"] = np.zeros((n,), f32)
    st["denom"] = np.zeros((n, 1), f32)
    st["deformation_table"] = np.concatenate([st["deformation_table"], new_table], -1)
    st["deformation_accum"] = np.zeros((n, 3), f32)


def prune_points(st, mask):
    """fdgs_trainer.py:368-386 with :319-339"""
    valid = ~mask
    for name in NAMES:
        st["params"][name] = st["params"][name][valid]
        for m in ("exp_avg", "exp_avg_sq"):
            if st.get(m) is not None:
                st[m][name] = st[m][name][valid]
    for k in ("xyz_gradient_accum", "max_radii2D", "denom", "deformation_accum", "deformation_table"):
        st[k] = st[k][valid]


def densify_and_clone(st, grads, grad_threshold, scene_extent, percent_dense):
    """fdgs_trainer.py:426-449"""
    sel = np.linalg.norm(grads.astype(f32), axis=-1) >= f32(grad_threshold)
    sel = np.logical_and(sel, _get_scaling(st).max(1) <= f32(percent_dense * scene_extent))
    d = {name: st["params"][name][sel] for name in NAMES}
    _densification_postfix(st, d, st["deformation_table"][sel])
    return sel


def densify_and_split(st, grads, grad_threshold, scene_extent, percent_dense, unit_noise, N=2):
    """fdgs_trainer.py:388-424.  unit_noise: [N * selected, 3] N(0,1) samples (see the module docstring)."""
    n_init = st["params"]["xyz"].shape[0]
    padded = np.zeros((n_init,), f32)
    padded[:grads.shape[0]] = grads.reshape(-1)
    sel = padded >= f32(grad_threshold)
    sel = np.logical_and(sel, _get_scaling(st).max(1) > f32(percent_dense * scene_extent))
    if not sel.any():
        return sel
    stds = np.tile(_get_scaling(st)[sel], (N, 1))
    samples = unit_noise.astype(f32) * stds
    rots = np.tile(build_rotation(st["params"]["rotation"][sel]), (N, 1, 1))
    new_xyz = np.einsum("nij,nj->ni", rots, samples).astype(f32) + np.tile(st["params"]["xyz"][sel], (N, 1))
    new_scaling = np.log(np.tile(_get_scaling(st)[sel], (N, 1)) / f32(0.8 * N))
    d = {"xyz": new_xyz,
         "f_dc": np.tile(st["params"]["f_dc"][sel], (N, 1, 1)),
         "f_rest": np.tile(st["params"]["f_rest"][sel], (N, 1, 1)),
         "opacity": np.tile(st["params"]["opacity"][sel], (N, 1)),
         "scaling": new_scaling,
         "rotation": np.tile(st["params"]["rotation"][sel], (N, 1))}
    _densification_postfix(st, d, np.tile(st["deformation_table"][sel], N))
    prune_filter = np.concatenate([sel, np.zeros(N * int(sel.sum()), bool)])
    prune_points(st, prune_filter)
    return sel


def densify(st, max_grad, extent, percent_dense, unit_noise_fn):
    """fdgs_trainer.py:487-492.  unit_noise_fn(count) -> [2 * count, 3] unit normal samples."""
    with np.errstate(divide="ignore", invalid="ignore"):
        grads = st["xyz_gradient_accum"].astype(f32) / st["denom"].astype(f32)
    grads[np.isnan(grads)] = 0.0
    clone_sel = densify_and_clone(st, grads, max_grad, extent, percent_dense)
    n_init = st["params"]["xyz"].shape[0]
    padded = np.zeros((n_init,), f32)
    padded[:grads.shape[0]] = grads.reshape(-1)
    count = int(np.logical_and(padded >= f32(max_grad),
                               _get_scaling(st).max(1) > f32(percent_dense * extent)).sum())
    split_sel = densify_and_split(st, grads, max_grad, extent, percent_dense, unit_noise_fn(count))
    return clone_sel, split_sel


def prune(st, min_opacity, extent, max_screen_size):
    """fdgs_trainer.py:477-484"""
    o = st["params"]["opacity"].astype(f32)
    mask = (f32(1) / (f32(1) + np.exp(-o)) < f32(min_opacity)).reshape(-1)
    if max_screen_size:
        big_vs = st["max_radii2D"] > f32(max_screen_size)
        big_ws = _get_scaling(st).max(1) > f32(0.1 * extent)
        mask = np.logical_or(np.logical_or(mask, big_vs), big_ws)
    prune_points(st, mask)
    return mask


def reset_opacity(st, cap=0.01):
    """gaussian_model.py:173-176 + fdgs_trainer.py:304-317 (moments of the replaced tensor are zeroed)"""
    o = st["params"]["opacity"].astype(f32)
    sig = f32(1) / (f32(1) + np.exp(-o))
    x = np.minimum(sig, np.ones_like(sig) * f32(cap))
    st["params"]["opacity"] = np.log(x / (f32(1) - x)).astype(f32)
    for m in ("exp_avg", "exp_avg_sq"):
        if st.get(m) is not None:
            st[m]["opacity"] = np.zeros_like(st["params"]["opacity"])
