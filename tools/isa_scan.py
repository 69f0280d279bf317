"""Disassembles the gfx950 code objects inside a HIP shared library and counts instructions by mnemonic.

A HIP library carries its device code as clang offload bundles in the `.hip_fatbin` section (one bundle per
translation unit): magic `__CLANG_OFFLOAD_BUNDLE__`, a 64-bit entry count, then per entry (offset, size, triple
length, triple), offsets relative to the bundle's start.  The code objects whose triple names gfx950 are written to a
temporary directory and disassembled with llvm-objdump.

    python tools/isa_scan.py <lib.so> [regex]     prints mnemonic counts (those matching the regex), per kernel
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(path, arch="gfx950"):
    data = open(path, "rb").read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if arch in triple and size > 0:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(MAGIC, pos + len(MAGIC))
    return out


def disassemble(path, arch="gfx950"):
    """{kernel or function symbol: [mnemonic, ...]} over all gfx950 code objects of the library."""
    per = {}
    with tempfile.TemporaryDirectory() as tmp:
        for i, blob in enumerate(code_objects(path, arch)):
            f = os.path.join(tmp, f"{i}.co")
            open(f, "wb").write(blob)
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f], capture_output=True, text=True,
                                 check=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    per.setdefault(cur, [])
                    continue
                m = re.match(r"^\s+([a-z_0-9]+)\b", line)
                if m and cur is not None:
                    per[cur].append(m.group(1))
    return per


def count(path, pattern):
    rx = re.compile(pattern)
    hits = {}
    for sym, ops in disassemble(path).items():
        n = sum(1 for o in ops if rx.search(o))
        if n:
            hits[sym] = n
    return hits


if __name__ == "__main__":
    lib = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "."
    hits = count(lib, pat)
    for sym, n in sorted(hits.items(), key=lambda kv: -kv[1]):
        print(f"{n:8d}  {sym}")
    print("total", sum(hits.values()), "in", len(hits), "symbols")
