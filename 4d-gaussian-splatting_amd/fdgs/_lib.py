"""ctypes binding of libfdgs.so (the C ABI declared in include/fdgs.h).

The library is the only compute path: there is no CPU or PyTorch fallback.  If
the shared object is missing or a symbol cannot be resolved, importing this
module raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDGS_LIBRARY") or os.path.join(_HERE, "libfdgs.so")   # override: a build of the same ABI

MAX_LEVELS = 4
GEOM_FLOATS = 12
TILE = 16

c_f = C.c_float
c_i32 = C.c_int32
c_i64 = C.c_int64
c_p = C.c_void_p
c_sz = C.c_size_t


class PreprocessFwdArgs(C.Structure):
    _fields_ = [("num_points", c_i32), ("sh_degree", c_i32), ("image_width", c_i32), ("image_height", c_i32),
                ("tanfovx", c_f), ("tanfovy", c_f), ("scale_modifier", c_f),
                ("viewmatrix", c_p), ("projmatrix", c_p), ("campos", c_p),
                ("means3D", c_p), ("scales", c_p), ("rotations", c_p), ("opacities", c_p), ("shs", c_p),
                ("colors_precomp", c_p), ("cov3D_precomp", c_p),
                ("geom", c_p), ("radii", c_p), ("tiles_touched", c_p), ("point_offsets", c_p),
                ("scan_workspace", c_p), ("activations", c_i32)]


class BinSortArgs(C.Structure):
    _fields_ = [("num_points", c_i32), ("image_width", c_i32), ("image_height", c_i32), ("num_rendered", c_i64),
                ("geom", c_p), ("point_offsets", c_p),
                ("keys_unsorted", c_p), ("vals_unsorted", c_p), ("keys_sorted", c_p), ("vals_sorted", c_p),
                ("ranges", c_p), ("sort_workspace", c_p), ("sort_workspace_bytes", c_sz),
                ("depth_bits_min", C.c_uint32), ("depth_bits_max", C.c_uint32), ("tile_order", c_p), ("phase", c_i32)]


class BlendFwdArgs(C.Structure):
    _fields_ = [("image_width", c_i32), ("image_height", c_i32),
                ("ranges", c_p), ("point_list", c_p), ("geom", c_p), ("bg", c_p),
                ("out_color", c_p), ("out_depth", c_p), ("final_T", c_p), ("n_contrib", c_p), ("tile_order", c_p),
                ("reach_masks", c_p)]


class BlendBwdArgs(C.Structure):
    _fields_ = [("image_width", c_i32), ("image_height", c_i32),
                ("ranges", c_p), ("point_list", c_p), ("geom", c_p), ("bg", c_p),
                ("final_T", c_p), ("n_contrib", c_p), ("dL_dcolor", c_p), ("dL_dgeom", c_p), ("tile_order", c_p),
                ("reach_masks", c_p)]


class PreprocessBwdArgs(C.Structure):
    _fields_ = [("num_points", c_i32), ("sh_degree", c_i32), ("image_width", c_i32), ("image_height", c_i32),
                ("tanfovx", c_f), ("tanfovy", c_f), ("scale_modifier", c_f),
                ("viewmatrix", c_p), ("projmatrix", c_p), ("campos", c_p),
                ("means3D", c_p), ("scales", c_p), ("rotations", c_p), ("shs", c_p),
                ("colors_precomp", c_p), ("cov3D_precomp", c_p),
                ("geom", c_p), ("dL_dgeom", c_p),
                ("dL_dmeans3D", c_p), ("dL_dmeans2D", c_p), ("dL_dscales", c_p), ("dL_drotations", c_p),
                ("dL_dopacities", c_p), ("dL_dshs", c_p), ("dL_dcolors", c_p), ("dL_dcov3D", c_p),
                ("activations", c_i32), ("clear_dL_dgeom", c_i32)]


L1_STATIC_MAX_N = 16384 * 4096          # FDGS_L1_STATIC_MAX_N
ETOOLARGE = -4

ACT_SCALE_EXP, ACT_ROTATION_NORMALIZE, ACT_OPACITY_SIGMOID = 1, 2, 4


class DeformParams(C.Structure):
    _fields_ = [("num_levels", c_i32), ("res", (c_i32 * 4) * MAX_LEVELS), ("aabb", c_p),
                ("planes", (c_p * 6) * MAX_LEVELS), ("w0", c_p), ("b0", c_p),
                ("w1", c_p * 5), ("b1", c_p * 5), ("w2", c_p * 5), ("b2", c_p * 5), ("matmul", c_i32)]


MATMUL_F32, MATMUL_SPLIT_BF16 = 0, 1


class DeformParamGrads(C.Structure):
    _fields_ = [("planes", (c_p * 6) * MAX_LEVELS), ("w0", c_p), ("b0", c_p),
                ("w1", c_p * 5), ("b1", c_p * 5), ("w2", c_p * 5), ("b2", c_p * 5)]


class DeformFwdArgs(C.Structure):
    _fields_ = [("num_points", c_i32), ("p", DeformParams),
                ("xyz", c_p), ("scales", c_p), ("rotations", c_p), ("opacity", c_p), ("shs", c_p),
                ("time", c_p), ("time_scalar", c_f),
                ("out_xyz", c_p), ("out_scales", c_p), ("out_rotations", c_p), ("out_opacity", c_p),
                ("out_shs", c_p), ("save_hidden", c_p), ("save_act", c_p), ("save_mask", c_p), ("save_yhid", c_p), ("save_yfeat", c_p), ("workspace", c_p), ("workspace_bytes", c_sz), ("shs_rest", c_p)]


class DeformBwdArgs(C.Structure):
    _fields_ = [("num_points", c_i32), ("p", DeformParams),
                ("xyz", c_p), ("time", c_p), ("time_scalar", c_f), ("hidden", c_p), ("act", c_p), ("mask", c_p), ("yhid", c_p), ("yfeat", c_p), ("planes_channel_last", c_p),
                ("g_xyz", c_p), ("g_scales", c_p), ("g_rotations", c_p), ("g_opacity", c_p), ("g_shs", c_p),
                ("d_xyz", c_p), ("dp", DeformParamGrads), ("workspace", c_p), ("workspace_bytes", c_sz),
                ("side_stream", c_p), ("fork_event", c_p), ("join_event", c_p)]


class HexFeaturesArgs(C.Structure):
    _fields_ = [("num_points", c_i32), ("p", DeformParams), ("xyz", c_p), ("time", c_p), ("time_scalar", c_f),
                ("features", c_p), ("workspace", c_p), ("workspace_bytes", c_sz)]


class L1FwdArgs(C.Structure):
    _fields_ = [("n", c_i64), ("a", c_p), ("b", c_p), ("scale", c_f), ("loss", c_p), ("partial", c_p), ("psnr", c_p),
                ("grad_a", c_p), ("add", c_p)]


class L1BwdArgs(C.Structure):
    _fields_ = [("n", c_i64), ("a", c_p), ("b", c_p), ("scale", c_f), ("grad_loss", c_p), ("grad_a", c_p)]


class DensifyStatsArgs(C.Structure):
    _fields_ = [("num_points", c_i32), ("radii", c_p), ("viewspace_grad", c_p), ("max_radii2D", c_p),
                ("xyz_gradient_accum", c_p), ("denom", c_p)]


REG_MAX_PLANES = 12


class PlaneRegArgs(C.Structure):
    _fields_ = [("num_planes", c_i32), ("channels", c_i32), ("plane", c_p * REG_MAX_PLANES),
                ("h", c_i32 * REG_MAX_PLANES), ("w", c_i32 * REG_MAX_PLANES),
                ("w_smooth", c_f * REG_MAX_PLANES), ("w_l1", c_f * REG_MAX_PLANES),
                ("grad", c_p * REG_MAX_PLANES), ("grad_scale", c_p), ("accumulate", c_i32),
                ("loss", c_p), ("loss_partial", c_p)]


CLS_KEEP, CLS_CLONE, CLS_SPLIT, CLS_DROP = 0, 1, 2, 3
SURGERY_MAX_ARRAYS = 24
SURGERY_ZERO_NEW = 1


class DensifyClassifyArgs(C.Structure):
    _fields_ = [("num_points", c_i32), ("xyz_gradient_accum", c_p), ("denom", c_p), ("scaling", c_p),
                ("grad_threshold", c_f), ("size_limit", c_f), ("cls", c_p)]


class PruneClassifyArgs(C.Structure):
    _fields_ = [("num_points", c_i32), ("opacity", c_p), ("max_radii2D", c_p), ("scaling", c_p),
                ("min_opacity", c_f), ("max_screen_size", c_f), ("world_limit", c_f), ("cls", c_p)]


class SurgeryPlanArgs(C.Structure):
    _fields_ = [("num_points", c_i32), ("cls", c_p), ("src", c_p), ("counts", c_p),
                ("workspace", c_p), ("workspace_bytes", c_sz)]


class SurgeryApplyArgs(C.Structure):
    _fields_ = [("rows_after", c_i32), ("rows_carried", c_i32), ("src", c_p), ("num_arrays", c_i32),
                ("in_", c_p * SURGERY_MAX_ARRAYS), ("out", c_p * SURGERY_MAX_ARRAYS),
                ("row_bytes", c_i32 * SURGERY_MAX_ARRAYS), ("flags", c_i32 * SURGERY_MAX_ARRAYS)]


class SplitChildrenArgs(C.Structure):
    _fields_ = [("num_children", c_i32), ("first_child_row", c_i32), ("src", c_p),
                ("xyz", c_p), ("scaling", c_p), ("rotation", c_p), ("noise", c_p),
                ("xyz_out", c_p), ("scaling_out", c_p)]


SSIM_MAX_RADIUS = 7


class SsimFwdArgs(C.Structure):
    _fields_ = [("planes", c_i32), ("height", c_i32), ("width", c_i32), ("img1", c_p), ("img2", c_p),
                ("radius", c_i32), ("window", c_f * (2 * SSIM_MAX_RADIUS + 1)), ("c1", c_f), ("c2", c_f),
                ("plane_sum", c_p), ("partial", c_p), ("dmaps", c_p)]


class SsimBwdArgs(C.Structure):
    _fields_ = [("planes", c_i32), ("height", c_i32), ("width", c_i32), ("img1", c_p), ("img2", c_p),
                ("radius", c_i32), ("window", c_f * (2 * SSIM_MAX_RADIUS + 1)), ("dmaps", c_p),
                ("plane_scale", c_p), ("grad_img1", c_p)]


ADAM_MAX_TENSORS = 48


class AdamArgs(C.Structure):
    _fields_ = [("num_tensors", c_i32), ("step", c_i32 * ADAM_MAX_TENSORS), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("lr", c_f * ADAM_MAX_TENSORS), ("numel", c_i64 * ADAM_MAX_TENSORS),
                ("param", c_p * ADAM_MAX_TENSORS), ("grad", c_p * ADAM_MAX_TENSORS),
                ("exp_avg", c_p * ADAM_MAX_TENSORS), ("exp_avg_sq", c_p * ADAM_MAX_TENSORS),
                ("grad_row_len", c_i32 * ADAM_MAX_TENSORS), ("grad_row_stride", c_i64 * ADAM_MAX_TENSORS)]


# every symbol include/fdgs.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "fdgs_strerror": (C.c_char_p, [C.c_int]),
    "fdgs_abi_version": (C.c_int, []),
    "fdgs_raster_scan_workspace_bytes": (c_sz, [c_i32]),
    "fdgs_raster_preprocess_fwd": (C.c_int, [C.POINTER(PreprocessFwdArgs), c_p]),
    "fdgs_raster_sort_workspace_bytes": (c_sz, [c_i64]),
    "fdgs_raster_bin_sort": (C.c_int, [C.POINTER(BinSortArgs), c_p]),
    "fdgs_raster_blend_fwd": (C.c_int, [C.POINTER(BlendFwdArgs), c_p]),
    "fdgs_raster_blend_bwd": (C.c_int, [C.POINTER(BlendBwdArgs), c_p]),
    "fdgs_raster_preprocess_bwd": (C.c_int, [C.POINTER(PreprocessBwdArgs), c_p]),
    "fdgs_deform_fwd_workspace_bytes": (c_sz, [C.POINTER(DeformParams), c_i32]),
    "fdgs_deform_fwd": (C.c_int, [C.POINTER(DeformFwdArgs), c_p]),
    "fdgs_deform_bwd_workspace_bytes": (c_sz, [C.POINTER(DeformParams), c_i32]),
    "fdgs_hexplane_features_workspace_bytes": (c_sz, [C.POINTER(DeformParams), c_i32]),
    "fdgs_hexplane_features": (C.c_int, [C.POINTER(HexFeaturesArgs), c_p]),
    "fdgs_l1_loss_partials": (c_sz, [c_i64]),
    "fdgs_l1_loss_fwd": (C.c_int, [C.POINTER(L1FwdArgs), c_p]),
    "fdgs_l1_loss_bwd": (C.c_int, [C.POINTER(L1BwdArgs), c_p]),
    "fdgs_l1_loss_fwd_static": (C.c_int, [C.POINTER(L1FwdArgs), c_p]),
    "fdgs_image_to8b": (C.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p]),
    "fdgs_densify_stats": (C.c_int, [C.POINTER(DensifyStatsArgs), c_p]),
    "fdgs_plane_reg_partials": (c_sz, [c_i32]),
    "fdgs_plane_regulation": (C.c_int, [C.POINTER(PlaneRegArgs), c_p]),
    "fdgs_knn_mean_dist2": (C.c_int, [c_p, c_i32, c_p, c_p]),
    "fdgs_adam_step": (C.c_int, [C.POINTER(AdamArgs), c_p]),
    "fdgs_densify_classify": (C.c_int, [C.POINTER(DensifyClassifyArgs), c_p]),
    "fdgs_prune_classify": (C.c_int, [C.POINTER(PruneClassifyArgs), c_p]),
    "fdgs_surgery_workspace_bytes": (c_sz, [c_i32]),
    "fdgs_surgery_plan": (C.c_int, [C.POINTER(SurgeryPlanArgs), c_p]),
    "fdgs_surgery_apply": (C.c_int, [C.POINTER(SurgeryApplyArgs), c_p]),
    "fdgs_split_children": (C.c_int, [C.POINTER(SplitChildrenArgs), c_p]),
    "fdgs_reset_opacity": (C.c_int, [c_p, c_i32, c_f, c_p]),
    "fdgs_ssim_tiles": (c_sz, [c_i32, c_i32]),
    "fdgs_ssim_fwd": (C.c_int, [C.POINTER(SsimFwdArgs), c_p]),
    "fdgs_ssim_bwd": (C.c_int, [C.POINTER(SsimBwdArgs), c_p]),
    "fdgs_deform_bwd": (C.c_int, [C.POINTER(DeformBwdArgs), c_p]),
}

BINSORT_ALL, BINSORT_COUNT, BINSORT_FINISH = 0, 1, 2      # fdgs_binsort_args.phase

ABI_VERSION = 25


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                          f"(hipcc --offload-arch=gfx950); there is no fallback path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.fdgs_abi_version() != ABI_VERSION:
        raise ImportError(f"libfdgs.so ABI {lib.fdgs_abi_version()} != binding ABI {ABI_VERSION}")
    return lib


lib = _load()


def check(status: int, what: str):
    if status != 0:
        raise RuntimeError(f"{what}: {lib.fdgs_strerror(status).decode()} ({status})")


# When set to a dict, every C-ABI call is bracketed by a pair of timing events
# recorded on the stream the kernels are launched on (name -> [(start, end), ...]).
PROFILE = None
# None, or the set of entry points to bracket (the others are called as if PROFILE were None): a pair of events is two
# more packets in the stream, and the kernels on either side of them start several microseconds apart
PROFILE_ONLY = None


def call(name, args):
    """Invoke entry point `name` with its argument struct on torch's current stream."""
    fn = getattr(lib, name)
    sp = torch.cuda.current_stream().cuda_stream
    if PROFILE is None or (PROFILE_ONLY is not None and name not in PROFILE_ONLY):
        rc = fn(C.byref(args), sp)
    else:
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(C.byref(args), sp)
        e.record()
        PROFILE.setdefault(name, []).append((s, e))
    check(rc, name)


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "libfdgs needs contiguous tensors"
    return t.data_ptr()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libfdgs kernels run on the GPU only; got a tensor on " + str(t.device))


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


_SIDE = {}


def side_resources():
    """A helper stream and two events for the calls enqueued on torch's current stream (fdgs_deform_bwd runs its
    HexPlane scatter beside its weight-gradient products when it is lent them).  The library owns no stream: these
    are PyTorch's, one triple per (device, stream), kept for the life of the process; calls on different streams
    get different triples and may overlap."""
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.cuda_stream)
    res = _SIDE.get(key)
    if res is None:
        with torch.cuda.device(cur.device):
            side = torch.cuda.Stream()
            fork, join = torch.cuda.Event(), torch.cuda.Event()
            fork.record(cur)        # torch creates the hipEvent_t on first use
            join.record(side)
        res = _SIDE[key] = (side, fork, join)
    side, fork, join = res
    return side.cuda_stream, fork.cuda_event, join.cuda_event
