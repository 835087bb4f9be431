#!/bin/bash
# Build libbxmi.so for gfx950 (MI355X) in-tree: bx-python_amd/bxmi/libbxmi.so.
# hipcc cross-compiles without a GPU; the .so travels to the GPU box with gpurun.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${BXMI_OUT:-$HERE/../bxmi/libbxmi.so}"   # BXMI_OUT + BXMI_DEFS: an experiment's build beside the shipped one (objects under _obj_<name>)
OBJ="$HERE/_obj${BXMI_OBJ_SUFFIX:-}"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 ${BXMI_DEFS:-} -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${BXMI_EXTRA_FLAGS:-}"
mkdir -p "$OBJ"
# a change of flags (BXMI_DEFS experiments) must rebuild everything
if [ "$(cat "$OBJ/.flags" 2>/dev/null)" != "$FLAGS" ]; then rm -f "$OBJ"/*.o; echo "$FLAGS" > "$OBJ/.flags"; fi
pids=()
for f in core intervals bitset bedparse comm; do
  src="$HERE/$f.hip"; [ -f "$src" ] || src="$HERE/$f.cpp"
  stale=0
  [ -f "$OBJ/$f.o" ] || stale=1
  for dep in "$src" "$HERE"/*.hpp "$HERE/../../include/bxmi.h"; do [ "$dep" -nt "$OBJ/$f.o" ] && stale=1; done
  if [ $stale = 1 ]; then
    $HIPCC $FLAGS -c "$src" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ/core.o" "$OBJ/intervals.o" "$OBJ/bitset.o" "$OBJ/bedparse.o" "$OBJ/comm.o" -ldl
echo "built $OUT"
