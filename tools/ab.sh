#!/bin/bash
# usage: bash tools/ab.sh "ENV=.. ENV=.." ...   -- one bench.py --no-cpu run per configuration;
# prints structures/s, the EdgeBlock kernel's and the c3 projection's average launch time (ms)
for cfg in "$@"; do
  echo "$cfg"
  env $cfg python3 bench.py --no-cpu --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d.get('roofline_projection') or {}
print(round(d['value']), round(d['roofline']['avg_launch_ms'], 4), p.get('avg_launch_ms'))" || exit 1
done
