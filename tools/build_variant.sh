#!/bin/bash
# Builds a VARIANT of libomrdeskew.so for side-by-side timing on one GPU box (tools/ab_lib.py):
#   tools/build_variant.sh <name> [-DDEFINE ...]      with SLANE_* environment variables for tools/gen_slane_asm.py
# -> omr-img-corrector_amd/lib/variants/libomrdeskew_<name>.so: slane.hip recompiled with the given defines and, if any
# SLANE_* variable is set, with a freshly generated wave program; every other object is the release build's.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
CS=$ROOT/omr-img-corrector_amd/csrc
VD=$ROOT/omr-img-corrector_amd/lib/variants
mkdir -p "$VD"
make -s -C "$CS" -j8
INC="slane_asm.inc"
if env | grep -q '^SLANE_'; then
  INC="$VD/slane_asm_$NAME.inc"
  SLANE_ASM_OUT="$INC" python3 "$ROOT/tools/gen_slane_asm.py" > /dev/null
fi
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-inline-asm"
/opt/rocm/bin/hipcc $FLAGS "$@" -DSLANE_ASM_INC="\"$INC\"" -I"$CS" -c -o "$VD/slane_$NAME.o" "$CS/slane.hip"
OBJS=$(ls "$ROOT"/omr-img-corrector_amd/lib/obj/*.o | grep -v '/slane\.o$')
/opt/rocm/bin/hipcc $FLAGS -shared -o "$VD/libomrdeskew_$NAME.so" $OBJS "$VD/slane_$NAME.o"
echo "built $VD/libomrdeskew_$NAME.so"
