#!/bin/bash
# Builds librn_potgnn.so (HIP kernels + C ABI) for gfx950, in-tree.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# RN_BUILD_TAG=x RN_EXTRA_FLAGS="-D..." builds a variant next to the product library
# (librn_potgnn_x.so, objects in build_x/; select it with RN_POTGNN_LIB=... for A/B timing).
tag="${RN_BUILD_TAG:-}"
out="$here/../librn_potgnn${tag:+_$tag}.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function ${RN_EXTRA_FLAGS:-}"
bdir="build${tag:+_$tag}"
mkdir -p "$here/$bdir"
pids=()
# RN_EXTRA_FLAGS=-DRN_EXPERIMENTS=1 adds the opt-in round-3 experiment kernels (experiments/ at the repository root, outside the package); the product build has none.
hip_srcs="api kernels_agg kernels_gemm kernels_fused kernels_edge_ps kernels_node_atom kernels_narrow kernels_bwd kernels_train spectrum"
case " ${RN_EXTRA_FLAGS:-} " in *" -DRN_EXPERIMENTS=1 "*) hip_srcs="$hip_srcs ../../experiments/kernels_fused_experiments ../../experiments/kernels_edge_frame";; esac
objs=()
for f in $hip_srcs; do
  o="$here/$bdir/$(basename "$f").o"
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$here/$f.hip" -nt "$o" ] || [ "$here/fused_common.hpp" -nt "$o" ] || \
     [ "$here/kernels.hpp" -nt "$o" ] || [ "$here/device_utils.hpp" -nt "$o" ] || \
     [ "$here/../../include/rn_potgnn.h" -nt "$o" ]; then
    extra=""
    # (the two kernels with hand-counted vmcnt waits keep their assembly listing for tools/check_ps_isa.py, below)
    case "$f" in kernels_edge_ps|kernels_node_atom) extra="-save-temps=obj";; esac
    $HIPCC $FLAGS $extra -c "$here/$f.hip" -o "$o" &
    pids+=($!)
  fi
done
for f in ingest ingest_vasprun; do
  if [ ! -f "$here/$bdir/$f.o" ] || [ "$here/$f.cpp" -nt "$here/$bdir/$f.o" ] || \
     [ "$here/ingest_common.hpp" -nt "$here/$bdir/$f.o" ] || \
     [ "$here/../../include/rn_ingest.h" -nt "$here/$bdir/$f.o" ]; then
    g++ -O3 -std=c++17 -fPIC -Wall -pthread -c "$here/$f.cpp" -o "$here/$bdir/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# The hand-counted `s_waitcnt vmcnt(N)` of kernels_edge_ps.hip / kernels_node_atom.hip are right only for the instruction stream
# this compiler emitted: check it, and fail the build on a violation (probe builds switch requests off on purpose: skipped).
case " ${RN_EXTRA_FLAGS:-} " in
  *RN_PS_PROBE*|*RN_NA_PROBE*) ;;
  *) lst_ps="$here/$bdir/kernels_edge_ps-hip-amdgcn-amd-amdhsa-gfx950.s"
     lst_na="$here/$bdir/kernels_node_atom-hip-amdgcn-amd-amdhsa-gfx950.s"
     if [ ! -f "$lst_ps" ] || [ ! -f "$lst_na" ]; then
       # objects from a build that kept no listing (an older build.sh, a copied build directory): compile the two again
       for f in kernels_edge_ps kernels_node_atom; do
         $HIPCC $FLAGS -save-temps=obj -c "$here/$f.hip" -o "$here/$bdir/$f.o" &
       done
       wait
     fi
     if ! python3 "$here/../../tools/check_ps_isa.py" "$lst_ps" "$lst_na" > "$here/$bdir/check_ps_isa.log" 2>&1; then
       tail -n 20 "$here/$bdir/check_ps_isa.log"
       echo "build.sh: tools/check_ps_isa.py found violations in the emitted instruction stream (full list: $bdir/check_ps_isa.log)"
       exit 1
     fi
     tail -n 1 "$here/$bdir/check_ps_isa.log" ;;
esac
rm -f "$here/$bdir"/*.hipi "$here/$bdir"/*.bc "$here/$bdir"/*.out "$here/$bdir"/*.resolution.txt "$here/$bdir"/*-host-*.s "$here/$bdir"/*.hipfb
$HIPCC -shared -fPIC --offload-arch=gfx950 -o "$out" "${objs[@]}" "$here/$bdir/ingest.o" "$here/$bdir/ingest_vasprun.o" -lpthread -ldl
echo "built $out"
