#!/bin/bash
# GPU check of the role-specialised EdgeBlock: parity tests that exercise it, then a short A/B bench of the product library
# against variant libraries (tags given as arguments, e.g. "r04" = librn_potgnn_r04.so) and against its own switches.
# Usage: bash tools/ps_check.sh [tags...]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -k "role_split or fused_edge_block or bit_identical or other_widths or split_f16 or batch_size or widens or default_device or config3 or full_size or eight_message" > gpurun_out/ps_tests.log 2>&1
rc=$?
echo "pytest exit $rc"; tail -5 gpurun_out/ps_tests.log
[ $rc -eq 0 ] || exit $rc
AB_ARGS="--no-extras" bash tools/ab_libs.sh "" "$@"
echo "RN_POTGNN_PS_GRAM=0:"; RN_POTGNN_PS_GRAM=0 AB_ARGS="--no-extras" bash tools/ab_libs.sh ""
