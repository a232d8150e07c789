#!/bin/bash
# GPU check of the role-specialised EdgeBlock: parity tests that exercise it, then a short A/B bench against the
# per-frame kernel (RN_POTGNN_EDGE_PS=0).  Usage (on the GPU box): bash tools/ps_check.sh
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_edge_block or bit_identical or other_widths or split_f16 or batch_size or widens" > gpurun_out/ps_tests.log 2>&1
echo "pytest exit $?"; tail -5 gpurun_out/ps_tests.log
for ps in 1 0; do
  RN_POTGNN_EDGE_PS=$ps timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 3 --warmup 1 > gpurun_out/ps_bench_$ps.log 2>&1
  echo "bench EDGE_PS=$ps exit $?"; tail -1 gpurun_out/ps_bench_$ps.log | cut -c1-600
done
