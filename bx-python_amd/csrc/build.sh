#!/bin/bash
# Build libbxmi.so for gfx950 (MI355X) in-tree: bx-python_amd/bxmi/libbxmi.so.
# hipcc cross-compiles without a GPU; the .so travels to the GPU box with gpurun.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../bxmi/libbxmi.so"
OBJ="$HERE/_obj"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 ${BXMI_DEFS:-} -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${BXMI_EXTRA_FLAGS:-}"
mkdir -p "$OBJ"
# a change of flags (BXMI_DEFS experiments) must rebuild everything
if [ "$(cat "$OBJ/.flags" 2>/dev/null)" != "$FLAGS" ]; then rm -f "$OBJ"/*.o; echo "$FLAGS" > "$OBJ/.flags"; fi
pids=()
for f in core intervals bitset bedparse comm; do
  src="$HERE/$f.hip"; [ -f "$src" ] || src="$HERE/$f.cpp"
  if [ ! -f "$OBJ/$f.o" ] || [ "$src" -nt "$OBJ/$f.o" ] || [ "$HERE/common.hpp" -nt "$OBJ/$f.o" ] \
     || [ "$HERE/primitives.hpp" -nt "$OBJ/$f.o" ] || [ "$HERE/count_bitmap.hpp" -nt "$OBJ/$f.o" ] || [ "$HERE/count_slices.hpp" -nt "$OBJ/$f.o" ] || [ "$HERE/count_dense.hpp" -nt "$OBJ/$f.o" ] || [ "$HERE/find_exchange.hpp" -nt "$OBJ/$f.o" ] || [ "$HERE/offset_cells.hpp" -nt "$OBJ/$f.o" ] || [ "$HERE/../../include/bxmi.h" -nt "$OBJ/$f.o" ]; then
    $HIPCC $FLAGS -c "$src" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ/core.o" "$OBJ/intervals.o" "$OBJ/bitset.o" "$OBJ/bedparse.o" "$OBJ/comm.o" -ldl
echo "built $OUT"
