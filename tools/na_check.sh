#!/bin/bash
# GPU check of the atom-owning NodeBlock: the parity tests that exercise it, then a short A/B bench against the
# row-ordered kernel (RN_POTGNN_NODE_ATOM=0).  Usage (on the GPU box): bash tools/na_check.sh
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -k "atom_owning or bit_identical or widens or batch_size" > gpurun_out/na_tests.log 2>&1
rc=$?
echo "pytest exit $rc"; tail -5 gpurun_out/na_tests.log
[ $rc -eq 0 ] || exit $rc
for na in 1 0; do
  RN_POTGNN_NODE_ATOM=$na timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 3 --warmup 1 > gpurun_out/na_bench_$na.log 2>&1
  echo "bench NODE_ATOM=$na exit $?"; tail -1 gpurun_out/na_bench_$na.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
n = d.get('roofline_nodeblock') or {}
print('%8.0f structures/s   edge %.4f ms   node %s ms  frac %s' % (d['value'], d['roofline']['avg_launch_ms'], n.get('avg_launch_ms'), n.get('frac')))"
done
