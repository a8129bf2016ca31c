#!/usr/bin/env bash
# Register, scratch and spill figures of every kernel of the shipped libraries, from the code objects inside them (no GPU needed):
#   profiles/kernel_resources.sh [lib.so] > profiles/rNN_kernel_resources.txt
# The gfx950 code object is taken out of the fat binary with clang-offload-bundler; llvm-readelf --notes prints the kernels' metadata (AMDGPU code object v5).
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
LIB=${1:-$HERE/../x265-amod_amd/lib/libx265amd_main.so}
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
# the .hip_fatbin section holds one bundle per translation unit
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" "$TMP/fat.bin"
python3 - "$TMP/fat.bin" "$TMP" <<'PY'
import sys
raw = open(sys.argv[1], "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
at, n = 0, 0
while True:
    i = raw.find(magic, at)
    if i < 0:
        break
    j = raw.find(magic, i + 1)
    open("%s/bundle%03d.bin" % (sys.argv[2], n), "wb").write(raw[i:j if j > 0 else len(raw)])
    n += 1
    at = i + 1
print(n, "bundles", file=sys.stderr)
PY
printf "%-64s %6s %6s %8s %8s %8s %8s\n" kernel vgpr sgpr "lds B" "scratch B" "vgpr spl" "sgpr spl"
for b in "$TMP"/bundle*.bin; do
    $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$b" --output="$b.co" --unbundle 2>/dev/null || continue
    [ -s "$b.co" ] || continue
    $LLVM/llvm-readelf --notes "$b.co" 2>/dev/null | python3 -c '
import re, sys
txt = sys.stdin.read()
for k in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    def f(name):
        m = re.search(r"\." + name + r":\s+(\S+)", k)
        return m.group(1) if m else "?"
    name = f("name")
    print("%-64s %6s %6s %8s %8s %8s %8s" % (name[:64], f("vgpr_count"), f("sgpr_count"), f("group_segment_fixed_size"), f("private_segment_fixed_size"), f("vgpr_spill_count"), f("sgpr_spill_count")))
'
done | sort -k5 -n -r
