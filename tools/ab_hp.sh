#!/bin/bash
# usage: bash tools/ab_hp.sh HPARAMS FRAMES "ENV=.." ...
hp=$1; frames=$2; shift; shift
for cfg in "$@"; do
  echo "$cfg"
  env $cfg python3 bench.py --no-cpu --steps 5 --warmup 2 --hparams $hp --frames $frames 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'], 3))" || exit 1
done
