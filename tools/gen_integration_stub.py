"""Writes the ctypes example of the integration guide from include/fdgs.h, so that the document cannot drift from
the header: the fields of fdgs_blend_fwd_args in declaration order (int32_t -> c_int32, float -> c_float, pointers ->
c_void_p) and the ABI number from csrc/api.hip.  The block between the two marker comments is replaced.

    python tools/gen_integration_stub.py            rewrites the guide in place
    python tools/gen_integration_stub.py --check    exits 1 if the file would change
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- stub:begin (tools/gen_integration_stub.py) -->", "<!-- stub:end -->"


def struct_fields(header, name):
    """[(field, ctypes type name)] of `typedef struct { ... } name;` (comments stripped; one or more declarators per
    statement)."""
    text = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    m = re.search(r"typedef struct\s*\{([^{}]*)\}\s*" + re.escape(name) + r"\s*;", text)
    if not m:
        raise ValueError(name + " not found")
    out = []
    for stmt in m.group(1).split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        base, names = stmt.rsplit(" ", 1)[0], None
        first, _, rest = stmt.partition(",")
        decls = [first.rsplit(" ", 1)[1]] + [d.strip() for d in rest.split(",") if d.strip()]
        base = first.rsplit(" ", 1)[0]
        for d in decls:
            pointer = "*" in base or d.startswith("*")
            field = d.lstrip("*")
            if pointer:
                ct = "C.c_void_p"
            else:
                ct = {"int32_t": "C.c_int32", "int64_t": "C.c_int64", "uint32_t": "C.c_uint32", "float": "C.c_float",
                      "size_t": "C.c_size_t", "int": "C.c_int"}[base.replace("const ", "")]
            out.append((field, ct))
    return out


def abi_version():
    src = open(os.path.join(ROOT, "4d-gaussian-splatting_amd", "csrc", "api.hip")).read()
    return int(re.search(r"fdgs_abi_version\s*\([^)]*\)\s*\{\s*return\s+(\d+)", src).group(1))


def stub():
    header = open(os.path.join(ROOT, "include", "fdgs.h")).read()
    fields = struct_fields(header, "fdgs_blend_fwd_args")
    rows, line = [], "    _fields_ = ["
    for i, (f, t) in enumerate(fields):
        item = f'("{f}", {t})' + ("," if i + 1 < len(fields) else "]")
        if len(line) + len(item) + 1 > 110:
            rows.append(line.rstrip())
            line = "                " + item + " "
        else:
            line += item + " "
    rows.append(line.rstrip())
    names = [f for f, _ in fields]
    call = ", ".join({"image_width": "W", "image_height": "H", "out_color": "color.data_ptr()",
                      "out_depth": "depth.data_ptr()", "tile_order": "None", "reach_masks": "None"}.get(
                          n, n + ".data_ptr()") for n in names)
    lines = ["```python", "import ctypes as C, torch", 'lib = C.CDLL(".../libfdgs.so")',
             f"assert lib.fdgs_abi_version() == {abi_version()}", "",
             "class BlendFwdArgs(C.Structure):            # fdgs_blend_fwd_args of include/fdgs.h, field for field"]
    lines += rows
    lines += ["", "lib.fdgs_raster_blend_fwd.restype = C.c_int",
              "lib.fdgs_raster_blend_fwd.argtypes = [C.POINTER(BlendFwdArgs), C.c_void_p]",
              "# tile_order / reach_masks are optional (NULL): the launch order of the tiles and the block verdicts kept",
              "# for the backward pass"]
    wrapped, cur = [], "args = BlendFwdArgs("
    for piece in call.split(", "):
        if len(cur) + len(piece) + 2 > 110:
            wrapped.append(cur.rstrip())
            cur = "                    " + piece + ", "
        else:
            cur += piece + ", "
    wrapped.append(cur.rstrip(", ") + ")")
    lines += wrapped
    lines += ["rc = lib.fdgs_raster_blend_fwd(C.byref(args), torch.cuda.current_stream().cuda_stream)",
              "if rc: raise RuntimeError(lib.fdgs_strerror(rc).decode())", "```"]
    return "\n".join(lines)


def render(doc):
    a, b = doc.index(BEGIN), doc.index(END)
    doc = doc[:a + len(BEGIN)] + "\n" + stub() + "\n" + doc[b:]
    return re.sub(r"`fdgs_abi_version\(\)` is \d+", f"`fdgs_abi_version()` is {abi_version()}", doc)


if __name__ == "__main__":
    path = os.path.join(ROOT, "INTEGRATION.md")
    old = open(path).read()
    new = render(old)
    if "--check" in sys.argv:
        sys.exit(0 if new == old else 1)
    open(path, "w").write(new)
