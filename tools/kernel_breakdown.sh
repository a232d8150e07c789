#!/bin/bash
# usage: bash tools/kernel_breakdown.sh "ENV=.. ENV=.." ... -- bench args   (per-kernel us per structure)
cfgs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do cfgs+=("$1"); shift; done; shift
for cfg in "${cfgs[@]}"; do
  echo "== $cfg"
  env $cfg python3 bench.py --no-cpu --steps 4 --warmup 1 --profile-all "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
n = d['config']['total_frames'] * d['steps'] if d['scaling']=='strong' else d['config']['frames_per_gpu'] * d['steps']
print(round(d['value']), 'structures/s;  us per structure:', {k: round(v / n * 1e3, 3) for k, v in d['kernel_ms'].items() if v})"
done
