#!/bin/bash
# usage: AB_ARGS="--hparams parity --no-extras" bash tools/ab_env.sh "VAR=1 OTHER=2" "" ...   -- one short bench.py run per
# environment setting ("" = defaults); prints structures/s and the EdgeBlock / NodeBlock kernels' average launch time
for envs in "$@"; do
  env $envs python3 bench.py --no-cpu --steps 3 --warmup 1 ${AB_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
n = d.get('roofline_nodeblock') or {}
print('%-40s %8.0f structures/s   edge %.4f ms x %d   node %s ms' % ('${envs:-defaults}', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], n.get('avg_launch_ms')))" || exit 1
done
