#!/bin/bash
# Build a variant of libbxmi.so with extra -D flags on intervals.hip only: build_variants/libbxmi_NAME.so
# usage: tools/build_variant.sh NAME "-DFOO=1 -DBAR=2"      (the other objects come from the default build's _obj/)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CSRC="$ROOT/bx-python_amd/csrc"
NAME=$1; DEFS=${2:-}
OUT="$ROOT/build_variants"; mkdir -p "$OUT/obj_$NAME"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 $DEFS -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -c "$CSRC/intervals.hip" -o "$OUT/obj_$NAME/intervals.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbxmi_$NAME.so" "$CSRC/_obj/core.o" "$OUT/obj_$NAME/intervals.o" "$CSRC/_obj/bitset.o" "$CSRC/_obj/bedparse.o" "$CSRC/_obj/comm.o" -ldl
echo "built $OUT/libbxmi_$NAME.so"
