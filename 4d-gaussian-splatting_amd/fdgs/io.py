"""On-disk formats of a trained scene, so that scenes written by the reference play back here and vice versa.

* `point_cloud.ply` — GaussianModel.save_ply / load_ply (src/model/gaussian_model.py:101-171): one `vertex`
  element of 62 little-endian float32 properties, x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2
  rot_0..3; the SH blocks are stored channel-major (`[N,3,K]` flattened), i.e. transposed with respect to the
  `[N,K,3]` parameters.  The reference goes through the `plyfile` package; the format is written and parsed
  here directly (binary_little_endian 1.0, the only flavour `plyfile` writes for this call).
* `deformation.pth`, `deformation_table.pth`, `deformation_accum.pth` — DeformationWrapper.save_deformation /
  load_model (src/model/deformation_wrapper.py:31-47): `torch.save` of the field's state_dict and of two
  tensors.  Loading uses `weights_only=True` (nothing in the file is executed).
"""
import os

import numpy as np
import torch


def ply_attributes(n_dc=3, n_rest=45, n_scale=3, n_rot=4):
    """Property names in file order (construct_list_of_attributes, gaussian_model.py:101-113)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def save_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """Tensors in the GaussianModel layout: xyz [N,3], features_dc [N,1,3], features_rest [N,K-1,3], opacity [N,1]
    (logits), scaling [N,3] (log), rotation [N,4] (raw)."""
    def a(t):
        return t.detach().to(torch.float32).cpu().numpy()
    xyz_ = a(xyz)
    f_dc = a(features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    f_rest = a(features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    table = np.concatenate((xyz_, np.zeros_like(xyz_), f_dc, f_rest, a(opacity).reshape(len(xyz_), -1), a(scaling),
                            a(rotation)), axis=1).astype("<f4")
    names = ply_attributes(f_dc.shape[1], f_rest.shape[1], scaling.shape[1], rotation.shape[1])
    assert table.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {table.shape[0]}"]
    header += [f"property float {n}" for n in names] + ["end_header"]
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(table).tobytes())


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_ply_vertices(path):
    """The first element of a binary little-endian or ASCII PLY file as a dict name -> float64 array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first, elements = None, None, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header without end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements += 1
                in_first = elements == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "binary_little_endian":
            dt = np.dtype(props)
            data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
            return {n: np.asarray(data[n], dtype=np.float64) for n, _ in props}
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2)
            return {n: rows[:, i].astype(np.float64) for i, (n, _) in enumerate(props)}
        raise ValueError(f"{path}: unsupported PLY format {fmt}")


def load_ply(path, max_sh_degree=3, device="cpu"):
    """Returns the six GaussianModel tensors (xyz, features_dc [N,1,3], features_rest [N,K-1,3], opacity [N,1],
    scaling, rotation), assembled exactly as load_ply (gaussian_model.py:133-171) does."""
    v = read_ply_vertices(path)
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
    opacity = v["opacity"][..., None]
    f_dc = np.stack((v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]), axis=1)[..., None]                 # [N,3,1]
    key = lambda s: int(s.split("_")[-1])
    rest_names = sorted((n for n in v if n.startswith("f_rest_")), key=key)
    if len(rest_names) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest_names)} f_rest properties do not match SH degree {max_sh_degree}")
    f_rest = np.stack([v[n] for n in rest_names], axis=1).reshape(xyz.shape[0], 3, (max_sh_degree + 1) ** 2 - 1)
    scales = np.stack([v[n] for n in sorted((n for n in v if n.startswith("scale_")), key=key)], axis=1)
    rots = np.stack([v[n] for n in sorted((n for n in v if n.startswith("rot")), key=key)], axis=1)
    t = lambda x: torch.tensor(x, dtype=torch.float, device=device)
    return {"xyz": t(xyz), "features_dc": t(f_dc).transpose(1, 2).contiguous(),
            "features_rest": t(f_rest).transpose(1, 2).contiguous(), "opacity": t(opacity),
            "scaling": t(scales), "rotation": t(rots)}


def save_deformation(path, deformation_fields, deformation_table=None, deformation_accum=None):
    """deformation_wrapper.py:43-47 (the two bookkeeping tensors are written when given)."""
    os.makedirs(path, exist_ok=True)
    torch.save(deformation_fields.state_dict(), os.path.join(path, "deformation.pth"))
    if deformation_table is not None:
        torch.save(deformation_table, os.path.join(path, "deformation_table.pth"))
    if deformation_accum is not None:
        torch.save(deformation_accum, os.path.join(path, "deformation_accum.pth"))


def load_deformation(model_dir, deformation_fields, n_points, device="cpu"):
    """deformation_wrapper.py:31-41: loads the state_dict in place and returns (deformation_table,
    deformation_accum), defaulting to all-true / zeros as the reference does."""
    sd = torch.load(os.path.join(model_dir, "deformation.pth"), map_location=device, weights_only=True)
    deformation_fields.load_state_dict(sd)
    table = torch.gt(torch.ones(n_points, device=device), 0)
    accum = torch.zeros(n_points, 3, device=device)
    p = os.path.join(model_dir, "deformation_table.pth")
    if os.path.exists(p):
        table = torch.load(p, map_location=device, weights_only=True)
    p = os.path.join(model_dir, "deformation_accum.pth")
    if os.path.exists(p):
        accum = torch.load(p, map_location=device, weights_only=True)
    return table, accum
