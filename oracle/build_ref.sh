#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds the *reference itself* (DJATOM/x265-aMod @ 3.6+1-aa7f602f7) from the
# sources where they lie under /root/reference into oracle/_ref/ (git-ignored).  Nothing is copied into the repo.
#
# The reference's own CMake project does not configure under cmake 4.x, so this is the
# "direct g++ on the reference's own translation units" recipe of SURVEY.md section 8(c).  Flags mirror
# source/CMakeLists.txt:226-362,449,472-485 and source/common/CMakeLists.txt:141-144.  No assembler exists in
# this image, so the result is the reference's C-primitive ([noasm]) build -- the parity target.
#
# x265_config.h is produced by instantiating the reference's OWN template source/x265_config.h.in with the
# value the reference's own CMakeLists.txt:32 sets (X265_BUILD 209) -- exactly what cmake's configure_file does.
#
# Outputs (per bit depth D in 8 10):
#   oracle/_ref/libx265_refD.so   full encoder library (x265_api_get_209 ...)
#   oracle/_ref/x265_refD         CLI
#   oracle/_ref/librefprimsD.so   tiny C-ABI driver (oracle/refprims.cpp) around the reference primitive table
#   oracle/_ref/x265_dropinD      reference encoder + optional x265amd_setup_primitives() override (oracle/ref_encode_with_table.cpp)
#   oracle/_ref/x265_rc_dumpD     the reference encoder + a per-picture record of its rate control's and lookahead's decisions (oracle/ref_rc_dump.cpp)
#   oracle/_ref/x265_abi_driverD  a libx265 client over another library's x265_api table (oracle/abi_driver.cpp; tests/test_x265_api_abi.py)
set -euo pipefail
REF=${X265_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
if [ ! -d "$REF/source" ]; then
    echo "build_ref.sh: $REF/source not present (GPU box?) -- keeping prebuilt oracle/_ref as is" >&2
    exit 0
fi
mkdir -p "$OUT"
SRC=$REF/source
BUILD=$(sed -n 's/^set(X265_BUILD \([0-9]*\)).*/\1/p' "$SRC/CMakeLists.txt")
mkdir -p "$OUT/cfg"
sed "s/\${X265_BUILD}/$BUILD/" "$SRC/x265_config.h.in" > "$OUT/cfg/x265_config.h"
VER="3.6+1-aa7f602f7"

LIBSRC=$(ls "$SRC"/common/*.cpp "$SRC"/encoder/*.cpp | grep -v winxp.cpp)
CLISRC="$SRC/input/input.cpp $SRC/input/y4m.cpp $SRC/input/yuv.cpp $(ls "$SRC"/output/*.cpp) $SRC/x265.cpp $SRC/x265cli.cpp $SRC/abrEncApp.cpp"

build_depth() {
    local D=$1 HBD=0
    [ "$D" != 8 ] && HBD=1
    local O=$OUT/obj$D
    mkdir -p "$O"
    local FLAGS="-O2 -std=gnu++11 -fPIC -ffast-math -mstackrealign -fno-exceptions -w \
      -DX265_ARCH_X86=1 -DX86_64=1 -DHAVE_INT_TYPES_H=1 -D__STDC_LIMIT_MACROS=1 \
      -DHIGH_BIT_DEPTH=$HBD -DX265_DEPTH=$D -DEXPORT_C_API=1 -DX265_NS=x265 -DHAVE_STRTOK_R=1 \
      -DX265_VERSION=$VER -I$OUT/cfg -I$SRC -I$SRC/common -I$SRC/encoder -I$SRC/input -I$SRC/output"
    local objs="" cliobjs="" pids=()
    local n=0
    for f in $LIBSRC $CLISRC; do
        local o=$O/$(basename "$(dirname "$f")")_$(basename "$f" .cpp).o
        case " $CLISRC " in *" $f "*) cliobjs="$cliobjs $o";; *) objs="$objs $o";; esac
        if [ ! -f "$o" ] || [ "$f" -nt "$o" ]; then
            g++ $FLAGS -c "$f" -o "$o" &
            n=$((n+1))
            if [ $((n % 8)) -eq 0 ]; then wait; fi
        fi
    done
    wait
    g++ -shared -o "$OUT/libx265_ref$D.so" $objs -lpthread -ldl
    g++ -o "$OUT/x265_ref$D" $cliobjs $objs -lpthread -ldl
    # C-ABI driver around the reference primitive table (our file, includes reference headers at compile time)
    g++ $FLAGS -shared -o "$OUT/librefprims$D.so" "$HERE/refprims.cpp" $objs -lpthread -ldl
    # the reference encoder as a program whose primitive table can be overridden through the drop-in hook (our file + reference objects)
    g++ $FLAGS -o "$OUT/x265_dropin$D" "$HERE/ref_encode_with_table.cpp" $objs -lpthread -ldl
    # a libx265 client that fills x265_param with the reference's own functions and encodes through ANOTHER library's x265_api table (our file + reference objects)
    g++ $FLAGS -o "$OUT/x265_abi_driver$D" "$HERE/abi_driver.cpp" $objs -lpthread -ldl
    # the reference encoder with its lookahead's and rate control's decisions per picture written out (our file + reference objects; CRF / AQ / cuTree ground truth)
    g++ $FLAGS -o "$OUT/x265_rc_dump$D" "$HERE/ref_rc_dump.cpp" $objs -lpthread -ldl
}
for D in 8 10; do build_depth $D; done
echo "oracle/_ref built: $(ls "$OUT" | tr '\n' ' ')"
