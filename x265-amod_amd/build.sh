#!/usr/bin/env bash
# Builds libx265amd_main.so (8-bit pixels) and libx265amd_main10.so (10-bit pixels) for gfx950, in-tree.
# hipcc cross-compiles without a GPU.  Usage: build.sh [-j]   (rebuilds only when sources are newer)
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$HERE/csrc
OUT=$HERE/lib
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-missing-braces -I$HERE/../include ${X265AMD_EXTRA_FLAGS:-}"
SRCS=$(ls "$SRC"/*.hip)
HOSTSRCS=$(ls "$HERE"/host/*.cpp 2>/dev/null || true)
build_one() {
    local depth=$1 name=$2
    local lib=$OUT/$name
    local newest
    newest=$(ls -t "$SRC"/* "$HERE"/host/* "$HERE"/../include/*.h | head -1)
    if [ -z "${X265AMD_FORCE:-}" ] && [ -f "$lib" ] && [ "$lib" -nt "$newest" ]; then return 0; fi
    local objs="" pids=""
    for f in $SRCS; do
        local o=$OUT/$(basename "$f" .hip).$depth.o
        if [ -z "${X265AMD_FORCE:-}" ] && [ -f "$o" ] && [ "$o" -nt "$f" ] && [ -z "$(find "$SRC" "$HERE/../include" "$HERE/host" -name '*.h' -newer "$o" | head -1)" ]; then
            objs="$objs $o"; continue       # object newer than its source and every header
        fi
        rm -f "$o"
        $HIPCC $FLAGS -DX265AMD_DEPTH=$depth -c "$f" -o "$o" &
        pids="$pids $!"
        objs="$objs $o"
    done
    for f in $HOSTSRCS; do
        local o=$OUT/$(basename "$f" .cpp).$depth.host.o
        rm -f "$o"
        # fm_*.cpp: double-precision arithmetic that has to come out like the reference's, which is compiled -O2 -ffast-math (source/CMakeLists.txt:226-240)
        local fm=""
        case "$(basename "$f")" in fm_*) fm="-ffast-math";; esac
        g++ -O2 $fm -std=c++17 -fPIC -Wall -I"$HERE/../include" -DX265AMD_DEPTH=$depth -c "$f" -o "$o" &
        pids="$pids $!"
        objs="$objs $o"
    done
    local failed=0
    for p in $pids; do wait "$p" || failed=1; done      # a failed compile must not link yesterday's object
    if [ "$failed" != 0 ]; then echo "build.sh: compilation failed" >&2; rm -f "$lib"; exit 1; fi
    $HIPCC --offload-arch=gfx950 -shared -o "$lib" $objs
}
for d in ${X265AMD_DEPTHS:-8 10}; do
    if [ "$d" = 8 ]; then build_one 8 libx265amd_main.so; else build_one "$d" libx265amd_main$d.so; fi
done
# command line front end (host C++; loads the library for the input's bit depth with dlopen)
mkdir -p "$HERE/bin"
if [ ! -f "$HERE/bin/x265amd" ] || [ "$HERE/cli/x265amd_cli.cpp" -nt "$HERE/bin/x265amd" ] || [ "$HERE/../include/x265amd_encoder.h" -nt "$HERE/bin/x265amd" ]; then
    g++ -O2 -std=c++17 -Wall -o "$HERE/bin/x265amd" "$HERE/cli/x265amd_cli.cpp" -ldl
fi
echo "built: $(ls "$OUT"/*.so | tr '\n' ' ') $HERE/bin/x265amd"
