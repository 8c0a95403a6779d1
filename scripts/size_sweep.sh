#!/bin/bash
# usage: scripts/size_sweep.sh <spots> "ENV=.." ...   short bench at a given total spot count per environment setting
spots=$1; shift
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --spots $spots --steps 60 --warmup 15 --no-alt --no-train --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$spots', '$e', 'ms/step', d['ms_per_step'], 'spots/s', d['value'], 'lanes', d['config']['lanes_per_gpu'])
" || exit 1
done
