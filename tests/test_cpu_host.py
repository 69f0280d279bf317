"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol
include/fdgs.h declares, argument validation works without touching a device,
the synthetic scene generator follows the reference's camera conventions, the
drop-in module keeps the reference's state_dict layout, and the product code
never imports the oracle."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from fdgs import _lib
    header = open(os.path.join(ROOT, "include", "fdgs.h")).read()
    declared = set(re.findall(r"\b(fdgs_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for name in declared:
        assert getattr(_lib.lib, name) is not None
    assert _lib.lib.fdgs_abi_version() == _lib.ABI_VERSION
    assert _lib.lib.fdgs_strerror(0) == b"ok"
    assert b"workspace" in _lib.lib.fdgs_strerror(-3)


def test_argument_validation_without_device():
    """Null / negative arguments are rejected before any launch (no GPU needed)."""
    from fdgs import _lib
    a = _lib.PreprocessFwdArgs()
    a.num_points = -1
    assert _lib.lib.fdgs_raster_preprocess_fwd(C.byref(a), None) == -1
    a.num_points, a.image_width, a.image_height, a.sh_degree = 10, 64, 64, 7
    assert _lib.lib.fdgs_raster_preprocess_fwd(C.byref(a), None) == -1
    a.sh_degree = 3          # pointers still null
    assert _lib.lib.fdgs_raster_preprocess_fwd(C.byref(a), None) == -1
    b = _lib.BlendFwdArgs()
    assert _lib.lib.fdgs_raster_blend_fwd(C.byref(b), None) == -1
    d = _lib.DeformFwdArgs()
    d.num_points = 5
    assert _lib.lib.fdgs_deform_fwd(C.byref(d), None) == -1      # num_levels == 0
    assert _lib.lib.fdgs_raster_sort_workspace_bytes(1000) >= 256 * 4
    assert _lib.lib.fdgs_raster_scan_workspace_bytes(300000) >= 4


def test_cpu_tensors_are_refused():
    from fdgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    S = GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.ones(3), 1.0, torch.eye(4), torch.eye(4), 3,
                                      torch.zeros(3), False, False)
    with pytest.raises(RuntimeError, match="GPU only"):
        GaussianRasterizer(S)(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), shs=torch.zeros(4, 16, 3),
                              colors_precomp=None, opacities=torch.ones(4, 1), scales=torch.ones(4, 3),
                              rotations=torch.ones(4, 4), cov3D_precomp=None)


def test_deformation_module_state_dict_layout():
    """39 keys, names and shapes of the reference's DeformationFields (fixture recorded from the reference)."""
    import deform_cases as dc
    from fdgs.deformation import DeformationFields
    gold = np.load(os.path.join(ROOT, "tests", "golden", "deform_dynerf.npz"))
    net = DeformationFields(dc.net_kwargs("dynerf"))
    sd = net.state_dict()
    assert sorted(sd.keys()) == sorted(gold["state_keys"].tolist())
    assert len(sd) == 39
    ref_sd = dc.make_state("dynerf")
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref_sd[k].shape), k
    g = net.deformation_net.grid.grids
    assert float(g[0][2].detach().min()) == 1.0 == float(g[0][2].detach().max())   # time planes start at one
    assert 0.1 <= float(g[0][0].detach().min()) and float(g[0][0].detach().max()) <= 0.5
    net.load_state_dict(ref_sd)
    assert len(net.get_grid_parameters()) == 13 and len(net.get_mlp_parameters()) == 22   # aabb counts as "grid"
    net.deformation_net.set_aabb([1, 2, 3], [-1, -2, -3])
    assert net.get_aabb[0].tolist() == [1, 2, 3] and not net.deformation_net.grid.aabb.requires_grad
    assert tuple(g[0][2].shape) == (1, 16, 150, 64) and tuple(g[1][0].shape) == (1, 16, 128, 128)


def test_synthetic_cameras_follow_reference_conventions():
    from fdgs import scene_synth as ss
    from oracle import raster_ref as rr
    cams = ss.make_cameras(5, 1352, 1014)
    for cam in cams:
        wvt, full, center = rr.camera_tensors(cam.R, cam.T, cam.FoVx, cam.FoVy)
        assert torch.allclose(cam.world_view_transform, wvt, atol=1e-6)
        assert torch.allclose(cam.full_proj_transform, full, atol=1e-5)
        assert torch.allclose(cam.camera_center, center, atol=1e-5)
        # the scene origin is in front of the camera and projects to the image centre
        p = torch.tensor([0.0, 0.0, 0.0, 1.0]) @ cam.full_proj_transform
        assert p[3] > 0.2 and abs(float(p[0] / p[3])) < 1e-4 and abs(float(p[1] / p[3])) < 1e-4
    assert abs(math.degrees(cams[0].FoVx) - 85.6) < 0.2 and 0.0 <= cams[3].time < 1.0


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "4d-gaussian-splatting_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text, f"{f} mentions the oracle"


def test_scene_files_round_trip(tmp_path):
    """fdgs.io: point_cloud.ply in the reference's layout (src/model/gaussian_model.py:101-171) and the
    deformation checkpoint files (src/model/deformation_wrapper.py:31-47)."""
    from fdgs import io as fio
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    g = ss.GaussianParams(257, device="cpu")
    path = str(tmp_path / "iteration_7" / "point_cloud.ply")
    fio.save_ply(path, g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation)
    raw = open(path, "rb").read()
    head, payload = raw.split(b"end_header\n", 1)
    lines = head.decode().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 257"]
    assert [l.split()[-1] for l in lines[3:-1]] == fio.ply_attributes() and len(fio.ply_attributes()) == 62
    # payload as the reference assembles it: rows of (xyz, normals, f_dc^T, f_rest^T, opacity, scale, rotation)
    rows = np.concatenate((g._xyz.detach().numpy(), np.zeros((257, 3), np.float32),
                           g._features_dc.detach().transpose(1, 2).flatten(start_dim=1).numpy(),
                           g._features_rest.detach().transpose(1, 2).flatten(start_dim=1).numpy(),
                           g._opacity.detach().numpy(), g._scaling.detach().numpy(), g._rotation.detach().numpy()), axis=1)
    assert payload == rows.astype("<f4").tobytes()
    back = fio.load_ply(path)
    for name, ref in (("xyz", g._xyz), ("features_dc", g._features_dc), ("features_rest", g._features_rest),
                      ("opacity", g._opacity), ("scaling", g._scaling), ("rotation", g._rotation)):
        assert back[name].shape == ref.shape and torch.equal(back[name], ref.detach()), name
    with pytest.raises(ValueError):
        fio.load_ply(path, max_sh_degree=2)

    net = ss.perturb_deformation_(DeformationFields(ss.DYNERF_NET_KWARGS))
    table = torch.rand(257) > 0.5
    fio.save_deformation(str(tmp_path / "ckpt"), net, deformation_table=table)
    net2 = DeformationFields(ss.DYNERF_NET_KWARGS)
    t2, acc2 = fio.load_deformation(str(tmp_path / "ckpt"), net2, 257)
    assert torch.equal(t2, table) and acc2.shape == (257, 3) and float(acc2.abs().sum()) == 0.0
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k


def test_schedules_match_reference_values():
    """fdgs/schedule.py against tests/golden/schedule.json: learning rates written by the reference's own
    Gaussian4DTrainer._update_learning_rate / get_expon_lr_func (tests/golden/make_schedule_golden.py)."""
    import json
    import torch
    from fdgs import schedule
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "schedule.json")))
    names = ["xyz", "deformation", "grid", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    groups = [{"params": [torch.nn.Parameter(torch.zeros(1))], "lr": 0.123, "name": n} for n in names]
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    lrs = schedule.LearningRates(spatial_lr_scale=gold["spatial_lr_scale"])
    for it in gold["iters"]:
        lrs.update(opt, it)
        for g in opt.param_groups:
            want = gold["lr"][str(it)][g["name"]]
            assert abs(g["lr"] - want) <= 1e-15 + 4e-16 * abs(want) * 10, (it, g["name"], g["lr"], want)
    f = schedule.get_expon_lr_func(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.05, max_steps=3000)
    for it, want in gold["delayed"].items():
        assert abs(f(int(it)) - want) <= 1e-14 * max(abs(want), 1e-3), (it, f(int(it)), want)
    assert schedule.get_expon_lr_func(0.0, 0.0)(10) == gold["zero"] == 0.0
    # fdgs_trainer.py:188-189 at both ends and in the middle
    o, d = schedule.fine_thresholds(0, 10000, 0.005, 0.005, 0.0002, 0.0002)
    assert (o, d) == (0.005, 0.0002)
    o, d = schedule.fine_thresholds(5000, 10000, 0.01, 0.005, 0.0004, 0.0002)
    assert abs(o - 0.0075) < 1e-15 and abs(d - 0.0003) < 1e-15


def test_ctypes_structures_match_the_header(tmp_path):
    """Every argument structure of fdgs/_lib.py against the C header: a small C program (gcc, host only) prints
    sizeof and the offset of every field of the structure's C counterpart; names, order, offsets and sizes must
    agree, so a field added to one side only fails here and not as a corrupted pointer on the GPU."""
    import ctypes
    import shutil
    import subprocess
    from fdgs import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = {"PreprocessFwdArgs": "fdgs_preprocess_fwd_args", "BinSortArgs": "fdgs_binsort_args",
             "BlendFwdArgs": "fdgs_blend_fwd_args", "BlendBwdArgs": "fdgs_blend_bwd_args",
             "PreprocessBwdArgs": "fdgs_preprocess_bwd_args", "DeformParams": "fdgs_deform_params",
             "DeformParamGrads": "fdgs_deform_param_grads", "DeformFwdArgs": "fdgs_deform_fwd_args",
             "DeformBwdArgs": "fdgs_deform_bwd_args", "HexFeaturesArgs": "fdgs_hexplane_features_args",
             "L1FwdArgs": "fdgs_l1_fwd_args", "L1BwdArgs": "fdgs_l1_bwd_args",
             "DensifyStatsArgs": "fdgs_densify_stats_args", "PlaneRegArgs": "fdgs_plane_reg_args",
             "AdamArgs": "fdgs_adam_args", "DensifyClassifyArgs": "fdgs_densify_classify_args",
             "PruneClassifyArgs": "fdgs_prune_classify_args", "SurgeryPlanArgs": "fdgs_surgery_plan_args",
             "SurgeryApplyArgs": "fdgs_surgery_apply_args", "SplitChildrenArgs": "fdgs_split_children_args",
             "SsimFwdArgs": "fdgs_ssim_fwd_args", "SsimBwdArgs": "fdgs_ssim_bwd_args"}
    declared = {name for name in dir(_lib) if isinstance(getattr(_lib, name), type)
                and issubclass(getattr(_lib, name), ctypes.Structure) and getattr(_lib, name) is not ctypes.Structure}
    assert declared == set(pairs), declared ^ set(pairs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "fdgs.h"', 'int main(void) {']
    for cls, cname in pairs.items():
        lines.append(f'  printf("{cls} sizeof %zu\\n", sizeof({cname}));')
        for field, _ in getattr(_lib, cls)._fields_:
            # a trailing underscore only keeps a Python keyword out of the way (in_ is the C field `in`)
            lines.append(f'  printf("{cls} {field} %zu\\n", offsetof({cname}, {field.rstrip("_")}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True,
                   capture_output=True, text=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.splitlines():
        cls, field, value = line.split()
        c = getattr(_lib, cls)
        want = ctypes.sizeof(c) if field == "sizeof" else getattr(c, field).offset
        assert int(value) == want, (cls, field, int(value), want)
        seen += 1
    assert seen == sum(1 + len(getattr(_lib, c)._fields_) for c in pairs)


def test_library_has_no_packed_fp32_arithmetic():
    """The built library's gfx950 code holds no packed fp32 instruction (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32,
    v_pk_mov_b32): csrc/Makefile switches both vectorizers off because such instructions returned wrong low halves
    in lanes 48-63 when a wave issuing bf16 matrix instructions shared the compute unit (tools/share_probe.sh).  A
    flag dropped from one translation unit, or vector arithmetic written by hand, fails here."""
    import importlib.util
    import shutil
    spec = importlib.util.spec_from_file_location("isa_scan", os.path.join(ROOT, "tools", "isa_scan.py"))
    isa_scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(isa_scan)
    if not os.path.exists(isa_scan.OBJDUMP) and shutil.which("llvm-objdump") is None:
        pytest.skip("no llvm-objdump")
    from fdgs import _lib
    per = isa_scan.disassemble(_lib.LIB_PATH)
    kernels = [k for k in per if "kernel" in k]
    assert len(kernels) > 40, "the disassembly did not find the library's kernels"
    packed = {k: sorted({o for o in ops if o.startswith("v_pk_") and (o.endswith("_f32") or o == "v_pk_mov_b32")})
              for k, ops in per.items()}
    packed = {k: v for k, v in packed.items() if v}
    assert not packed, packed


def test_integration_document_matches_header_and_abi():
    """The integration guide's ctypes example is the output of tools/gen_integration_stub.py for the current header (a reader
    who copies it passes a structure of the right length), the structure it defines has the binding's own layout,
    and every ABI number the document quotes is the library's."""
    import ctypes
    import subprocess
    import sys
    from fdgs import _lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_integration_stub.py"), "--check"])
    assert r.returncode == 0, "the integration guide is stale: run python tools/gen_integration_stub.py"
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    quoted = [int(v) for v in re.findall(r"fdgs_abi_version\(\)`? (?:is|==) (\d+)", doc)]
    assert quoted and all(v == _lib.lib.fdgs_abi_version() for v in quoted), quoted
    block = doc[doc.index("class BlendFwdArgs(C.Structure)"):]
    block = block[:block.index("\n\n")]
    scope = {"C": ctypes}
    exec(block, scope)
    mine, theirs = scope["BlendFwdArgs"], _lib.BlendFwdArgs
    assert [f for f, _ in mine._fields_] == [f for f, _ in theirs._fields_]
    assert ctypes.sizeof(mine) == ctypes.sizeof(theirs)
    assert all(getattr(mine, f).offset == getattr(theirs, f).offset for f, _ in mine._fields_)
