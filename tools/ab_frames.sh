#!/bin/bash
# usage: bash tools/ab_frames.sh FRAMES "ENV=.." ...  -- like ab.sh with --frames FRAMES
frames=$1; shift
for cfg in "$@"; do
  echo "$cfg"
  env $cfg python3 bench.py --no-cpu --steps 3 --warmup 1 --frames $frames 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['roofline']['avg_launch_ms'], 4))" || exit 1
done
