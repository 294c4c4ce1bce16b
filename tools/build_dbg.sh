#!/bin/bash
# Ablation builds of libicgan_hip.so (tools only, never the product): tools/libdbg_<NAME>.so, selected by tools/*.py through ICG_LIB.
#   NOBLK : plane GEMMs with single-level accumulation (-DICG_PLANES_BLOCKED=0);  F1 / F4: first-level chains of 1 / 4 K-tiles;
#   L1_384 : plane GEMMs with K <= 384 on the single-level kernel;  NOPERSIST: one output tile per workgroup (generic body)
#   LB3 : 128-column two-level plane GEMM forced to 3 waves per SIMD (spills)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
SRC="$HERE/../ic_gan_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-sometimes-uninitialized -Wno-uninitialized"
mkdir -p "$HERE/obj"
build_variant() {   # name, define, source file
  /opt/rocm/bin/hipcc $FLAGS $2 -c "$SRC/$3.hip" -o "$HERE/obj/$3_$1.o"
  objs=""
  for o in "$SRC"/obj/*.o; do
    [ "$(basename "$o")" = "$3.o" ] || objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$HERE/obj/$3_$1.o" -o "$HERE/libdbg_$1.so"
  echo "built $HERE/libdbg_$1.so"
}
for v in "$@"; do
  case "$v" in
    NOBLK) build_variant NOBLK -DICG_PLANES_BLOCKED=0 gemm_conv ;;
    F1) build_variant F1 -DICG_PLANES_FLUSH_TILES=1 gemm_conv ;;
    F4) build_variant F4 -DICG_PLANES_FLUSH_TILES=4 gemm_conv ;;
    L1_384) build_variant L1_384 -DICG_PLANES_1LEVEL_MAX_K=384 gemm_conv ;;
    NOPERSIST) build_variant NOPERSIST -DICG_PLANES_PERSISTENT=0 gemm_conv ;;
    LB3) build_variant LB3 -DICG_PLANES_TN4_MIN_WAVES=3 gemm_conv ;;
    FWA*) build_variant "$v" "-DFWINO_ABLATE=${v#FWA}" fwino ;;      # fused Winograd kernel, ablation bits (csrc/fwino.hip)
    FWM) build_variant FWM -DFWINO_MASKBITS=1 fwino ;;
    FWS) build_variant FWS "-DFWINO_SCALAR=1 -fno-slp-vectorize" fwino ;;      # ... producers on plain floats instead of float2 (v_pk_*)
    FWT) build_variant FWT -DFWINO_TRACE=1 fwino ;;                    # ... with barrier timestamps of workgroup 0 (tools/fwino_trace.py)
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
