#!/bin/bash
# usage: scripts/train_sweep.sh "ENV=.." ...   hipGraph training iteration time (512 cells, bf16, hidden 1024) per setting
for e in "$@"; do
  echo -n "$e: "; env $e timeout -k 10 120 python -u scripts/train_only.py ${TRAIN_G:-1024} ${TRAIN_H:-1024} 40 1 2>/dev/null | tail -1 || exit 1
done
