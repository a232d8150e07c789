#!/bin/bash
# Builds librn_potgnn.so (HIP kernels + C ABI) for gfx950, in-tree.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../librn_potgnn.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
mkdir -p "$here/build"
pids=()
for f in api kernels_agg kernels_gemm kernels_fused kernels_narrow kernels_bwd kernels_train spectrum; do
  if [ ! -f "$here/build/$f.o" ] || [ "$here/$f.hip" -nt "$here/build/$f.o" ] || \
     [ "$here/kernels.hpp" -nt "$here/build/$f.o" ] || [ "$here/device_utils.hpp" -nt "$here/build/$f.o" ] || \
     [ "$here/../../include/rn_potgnn.h" -nt "$here/build/$f.o" ]; then
    $HIPCC $FLAGS -c "$here/$f.hip" -o "$here/build/$f.o" &
    pids+=($!)
  fi
done
for f in ingest ingest_vasprun; do
  if [ ! -f "$here/build/$f.o" ] || [ "$here/$f.cpp" -nt "$here/build/$f.o" ] || \
     [ "$here/ingest_common.hpp" -nt "$here/build/$f.o" ] || \
     [ "$here/../../include/rn_ingest.h" -nt "$here/build/$f.o" ]; then
    g++ -O3 -std=c++17 -fPIC -Wall -pthread -c "$here/$f.cpp" -o "$here/build/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC -shared -fPIC --offload-arch=gfx950 -o "$out" "$here/build/api.o" "$here/build/kernels_agg.o" "$here/build/kernels_gemm.o" "$here/build/kernels_fused.o" "$here/build/kernels_narrow.o" "$here/build/kernels_bwd.o" "$here/build/kernels_train.o" "$here/build/spectrum.o" "$here/build/ingest.o" "$here/build/ingest_vasprun.o" -lpthread -ldl
echo "built $out"
