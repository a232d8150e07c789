#!/bin/bash
# usage: bash tools/train_ab.sh tag1 tag2 ...  -- tools/train_probe.py (config 5's shape, DeviceAdam) under rocprofv3 --kernel-trace, once per
# variant library librn_potgnn_<tag>.so ("" = the product library); prints the reverse / forward EdgeBlock lines and the last steps' wall clock
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export TMPDIR=/tmp
for tag in "$@"; do
  lib="$root/ramannoodle_amd/librn_potgnn${tag:+_$tag}.so"
  out="$root/gpurun_out/train_ab/${tag:-product}"
  rm -rf "$out"; mkdir -p "$out"
  RN_POTGNN_LIB="$lib" RN_PROBE_STEPS=8 RN_PROBE_256=1 RN_PROBE_DEVICE=1 rocprofv3 --kernel-trace -d "$out" -o t -- python3 "$root/tools/train_probe.py" perf 32 > "$out/probe.log" 2>&1 || { tail -5 "$out/probe.log"; exit 1; }
  db=$(find "$out" -name "*.db" | head -1)
  echo "== ${tag:-product}"
  python3 "$root/tools/train_summary.py" "$db" "$out/probe.log" > "$out/summary.txt"
  grep -E "edge_bwd|edge_block_ps|total kernel" "$out/summary.txt"
  grep "^step" "$out/summary.txt" | tail -2
done
