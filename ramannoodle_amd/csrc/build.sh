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
for f in api kernels_agg kernels_gemm kernels_fused kernels_narrow kernels_bwd kernels_train spectrum; do
  if [ ! -f "$here/$bdir/$f.o" ] || [ "$here/$f.hip" -nt "$here/$bdir/$f.o" ] || \
     [ "$here/kernels.hpp" -nt "$here/$bdir/$f.o" ] || [ "$here/device_utils.hpp" -nt "$here/$bdir/$f.o" ] || \
     [ "$here/../../include/rn_potgnn.h" -nt "$here/$bdir/$f.o" ]; then
    $HIPCC $FLAGS -c "$here/$f.hip" -o "$here/$bdir/$f.o" &
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
$HIPCC -shared -fPIC --offload-arch=gfx950 -o "$out" "$here/$bdir/api.o" "$here/$bdir/kernels_agg.o" "$here/$bdir/kernels_gemm.o" "$here/$bdir/kernels_fused.o" "$here/$bdir/kernels_narrow.o" "$here/$bdir/kernels_bwd.o" "$here/$bdir/kernels_train.o" "$here/$bdir/spectrum.o" "$here/$bdir/ingest.o" "$here/$bdir/ingest_vasprun.o" -lpthread -ldl
echo "built $out"
