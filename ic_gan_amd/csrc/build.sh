#!/bin/bash
# Build libicgan_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-sometimes-uninitialized -Wno-uninitialized"
pids=()
for src in "$HERE"/*.hip; do
  obj="$HERE/obj/$(basename "${src%.hip}").o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/icg_common.h" -nt "$obj" ] || [ "$HERE/../../include/icgan_hip.h" -nt "$obj" ]; then
    ( $HIPCC $FLAGS -c "$src" -o "$obj" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "$HERE"/obj/*.o -o "$OUT/libicgan_hip.so"
echo "built $OUT/libicgan_hip.so"
