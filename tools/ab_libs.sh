#!/bin/bash
# usage: bash tools/ab_libs.sh tag1 tag2 ...   -- one short bench.py run per variant library librn_potgnn_<tag>.so
# ("" = the product library); prints structures/s and the EdgeBlock / NodeBlock kernels' average launch time (ms)
for tag in "$@"; do
  lib="$PWD/ramannoodle_amd/librn_potgnn${tag:+_$tag}.so"
  RN_POTGNN_LIB="$lib" RN_POTGNN_MFMA=f16 python3 bench.py --no-cpu --steps 3 --warmup 1 ${AB_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
n = d.get('roofline_nodeblock') or {}
print('%-10s %8.0f structures/s   edge %.4f ms x %d   node %s ms' % ('${tag:-product}', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], n.get('avg_launch_ms')))" || exit 1
done
