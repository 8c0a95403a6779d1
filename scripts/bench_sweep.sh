#!/bin/bash
# usage: scripts/bench_sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   one short bench per environment setting
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-alt --no-train --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$e', 'slab', d['config']['rows_per_slab'], 'lanes', d['config']['lanes_per_gpu'], 'ms/step', d['ms_per_step'], 'spots/s', d['value'], 'step TF', d['step_tflops'], 'gemm TF', d['roofline']['achieved'])
" || exit 1
done
