#!/bin/bash
# GPU check of the split-f16 pair rows: parity tests, then an A/B bench against float32 rows (RN_POTGNN_PAIR_ROWS=0).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pair_rows or atom_owning or role_split or bit_identical or widens or batch_size or fused" > gpurun_out/pair_tests.log 2>&1
rc=$?
echo "pytest exit $rc"; tail -5 gpurun_out/pair_tests.log
[ $rc -eq 0 ] || exit $rc
for v in 1 0 1 0; do
  RN_POTGNN_PAIR_ROWS=$v timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
n = d.get('roofline_nodeblock') or {}
print('PAIR_ROWS=$v %8.0f structures/s   edge %.4f ms   node %s ms  frac %s' % (d['value'], d['roofline']['avg_launch_ms'], n.get('avg_launch_ms'), n.get('frac')))"
done
