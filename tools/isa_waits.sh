#!/bin/bash
# Reads a kernel's gfx950 assembly the way that found the hidden waits of the weight-gradient kernels:
#   1. per kernel of the file: scratch bytes against spilled registers (scratch without spills is a local array the
#      compiler could not keep in registers: every access is a vector-memory operation, and a wait on it waits for
#      every request in flight);
#   2. for kernels whose mangled name contains <pattern>: the sequence of loads (LD), waits (W(n) = s_waitcnt
#      vmcnt(n)), stores, scratch accesses (SL / SS), branches and runs of MFMAs, in program order.  In a loop that
#      requests its operands k iterations ahead, n should be about the number of requests meant to be in flight;
#      W(0) or W(1) inside it means the wave stops for a memory latency per iteration.
# Usage: bash tools/isa_waits.sh <csrc file, e.g. deform.hip> [pattern]      (cross-compiles; needs no GPU)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; PAT=${2:-}
CS="$ROOT/4d-gaussian-splatting_amd/csrc"
TMP=$(mktemp -d)
EXTRA=""
case "$SRC" in
  raster_preprocess.hip) EXTRA="-ffp-contract=off" ;;
esac
(cd "$TMP" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -fno-vectorize $EXTRA \
    -I"$ROOT/include" -I"$CS" --save-temps -c "$CS/$SRC" -o x.o 2>/dev/null)
ASM=$(ls "$TMP"/*-hip-amdgcn-amd-amdhsa-gfx950.s)
python3 - "$ASM" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
print("kernel".ljust(64), "vgpr", "spilled", "scratch bytes")
for b in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b); ps = re.search(r"\.private_segment_fixed_size:\s+(\d+)", b)
    sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", b); vg = re.search(r"\.vgpr_count:\s+(\d+)", b)
    if name and ps:
        flag = "   <- array in scratch" if int(ps.group(1)) > 4 * int(sp.group(1)) + 16 else ""
        print(name.group(1)[:62].ljust(64), vg.group(1).rjust(4), sp.group(1).rjust(7), ps.group(1).rjust(8) + flag)
PY
[ -z "$PAT" ] && { rm -rf "$TMP"; exit 0; }
for K in $(grep -o "^_Z[A-Za-z0-9_]*$PAT[A-Za-z0-9_]*:" "$ASM" | tr -d ':' | sort -u); do
  echo; echo "== $K"
  awk "/^$K:/,/^\\.Lfunc_end/" "$ASM" |
    grep -n "s_waitcnt vmcnt\|v_mfma\|global_load\|global_store\|global_atomic\|scratch_load\|scratch_store\|s_cbranch\|s_barrier" |
    awk -F'\t' '{print $2}' |
    awk '{ if ($0 ~ /v_mfma/) {m++} else if ($0 ~ /global_store/) {st++} else {
             if (m > 0) {printf "[%d mfma] ", m; m = 0}; if (st > 0) {printf "[%d st] ", st; st = 0}; printf "%s; ", $0 } }
         END { if (m > 0) printf "[%d mfma] ", m; print "" }' |
    sed 's/global_load_dwordx[234][^;]*/LD4/g; s/global_load_dword[^;]*/LD1/g; s/global_atomic[^;]*/AT/g;
         s/s_waitcnt vmcnt(\([0-9]*\))[^;]*/W(\1)/g; s/scratch_load[^;]*/SL/g; s/scratch_store[^;]*/SS/g;
         s/s_cbranch[^;]*/BR/g' | fold -w 150
done
rm -rf "$TMP"
