"""Drop-in for `src.gaussian_renderer.render` of the reference
(src/gaussian_renderer/__init__.py:21-120): same signature, same return dict
{"render", "viewspace_points", "visibility_filter", "radii", "depth"}.

`pc` is any object with the GaussianModel tensor layout
(src/model/gaussian_model.py:70-89: get_xyz, _scaling, _rotation, _opacity,
get_features, active_sh_degree, the three activation callables); `deforms` is
any object whose `deformation_fields` is callable like DeformationFields.

The reference forwards `shs` together with `colors_precomp` when
`override_color` / `convert_SHs_python` are used (:89-113), which its
rasterizer wrapper rejects; here `shs` is dropped in those two cases so the
flags do what their comments describe.
"""
import math

import torch

from . import _lib
from . import deformation as _deformation
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def prepare(viewpoint_camera, pc, deforms, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
            stage="fine", cam_type=None, debug=False, convert_SHs_python=False, compute_cov3D_python=False,
            screenspace=True):
    """Everything render() does before it calls the rasterizer (gaussian_renderer/__init__.py:21-102): the settings
    tuple, the deformation field, the activations.  Returns (raster_settings, the rasterizer's keyword arguments,
    the gradient carrier of the 2-D means).  screenspace=False: no carrier is made (a caller that only wants the
    image saves the [N,3] fill per frame); None is returned in its place."""
    means3D = pc.get_xyz
    dev = means3D.device
    # gradient carrier for the 2-D means (gaussian_renderer/__init__.py:27-31)
    # a leaf, so .grad is kept without retain_grad() and without the reference's "+ 0" launch
    screenspace_points = (torch.zeros_like(means3D, dtype=means3D.dtype, requires_grad=True, device=dev)
                          if screenspace else None)

    if cam_type != "PanopticSports":
        tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
        tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
        raster_settings = GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height),
            image_width=int(viewpoint_camera.image_width),
            tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
            viewmatrix=viewpoint_camera.world_view_transform.to(dev),
            projmatrix=viewpoint_camera.full_proj_transform.to(dev),
            sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center.to(dev),
            prefiltered=False, debug=debug)
        time_value = viewpoint_camera.time
    else:
        raster_settings = viewpoint_camera["camera"]
        time_value = viewpoint_camera["time"]

    means2D = screenspace_points
    opacity = pc._opacity
    # Without gradients, with this package's deformation field and a model that keeps the SH coefficients as the
    # reference's GaussianModel does, the two tensors go to the deformation kernel as they are: get_features is a
    # concatenation into a fresh [N,16,3] tensor on every call (gaussian_model.py:86-89)
    split_sh = ("fine" in stage and not torch.is_grad_enabled() and override_color is None and not convert_SHs_python
                and isinstance(getattr(deforms, "deformation_fields", None), _deformation.DeformationFields)
                and isinstance(getattr(pc, "_features_dc", None), torch.Tensor)
                and isinstance(getattr(pc, "_features_rest", None), torch.Tensor)
                and pc._features_dc.dim() == 3 and pc._features_dc.shape[1] == 1 and pc._features_rest.shape[1] == 15)
    shs = _deformation.SplitSH(pc._features_dc, pc._features_rest) if split_sh else pc.get_features
    scales = None
    rotations = None
    cov3D_precomp = None
    if compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc._scaling
        rotations = pc._rotation

    if "coarse" in stage:
        means3D_final, scales_final, rotations_final, opacity_final, shs_final = means3D, scales, rotations, opacity, shs
    elif "fine" in stage:
        if compute_cov3D_python:
            # the reference hands scales=None to the deformation field here and fails inside it
            # (src/gaussian_renderer/__init__.py:70-81); a precomputed covariance cannot be deformed
            raise ValueError("compute_cov3D_python is only usable with stage='coarse': the deformation field "
                             "needs scales and rotations")
        # the reference materialises time as an [N,1] tensor (:52); a python float reaches the same
        # kernel through its scalar-time argument without the N-element allocation
        means3D_final, scales_final, rotations_final, opacity_final, shs_final = deforms.deformation_fields(
            means3D, scales, rotations, opacity, shs, float(time_value))
    else:
        raise NotImplementedError

    # the activations of :83-86.  The reference's own three callables (src/model/gaussian_model.py:22-26) are
    # evaluated inside the rasterizer's preprocess kernels; anything else is called here as the reference does
    activations = 0
    if scales_final is not None:
        if pc.scaling_activation is torch.exp:
            activations |= _lib.ACT_SCALE_EXP
        else:
            scales_final = pc.scaling_activation(scales_final)
        if pc.rotation_activation is torch.nn.functional.normalize:
            activations |= _lib.ACT_ROTATION_NORMALIZE
        else:
            rotations_final = pc.rotation_activation(rotations_final)
    if pc.opacity_activation is torch.sigmoid:
        activations |= _lib.ACT_OPACITY_SIGMOID
        opacity = opacity_final
    else:
        opacity = pc.opacity_activation(opacity_final)

    colors_precomp = None
    if override_color is None:
        if convert_SHs_python:      # SH -> RGB on the host side, as gaussian_renderer/__init__.py:91-96
            from .sh import eval_sh
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.to(dev).repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)
            colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
            shs_final = None
    else:
        colors_precomp = override_color
        shs_final = None

    return raster_settings, dict(means3D=means3D_final, means2D=means2D, shs=shs_final, colors_precomp=colors_precomp,
                                 opacities=opacity, scales=scales_final, rotations=rotations_final,
                                 cov3D_precomp=cov3D_precomp, activations=activations), screenspace_points


def render(viewpoint_camera, pc, deforms, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
           stage="fine", cam_type=None, debug=False, convert_SHs_python=False, compute_cov3D_python=False):
    raster_settings, kwargs, screenspace_points = prepare(
        viewpoint_camera, pc, deforms, bg_color, scaling_modifier=scaling_modifier, override_color=override_color,
        stage=stage, cam_type=cam_type, debug=debug, convert_SHs_python=convert_SHs_python,
        compute_cov3D_python=compute_cov3D_python)
    rendered_image, radii, depth = GaussianRasterizer(raster_settings=raster_settings)(**kwargs)
    return {"render": rendered_image,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii,
            "depth": depth}
