#!/bin/bash
# Scenarios for scripts/diag/impute_leak.py (one fresh parent process each); stops at the first parent that fails.
set -u
out=gpurun_out/r3
mkdir -p $out
s() {
  name=$1; shift
  echo "=== $name: $* ARGS=$ARGS" >> $out/scen.log
  ( export "$@"; timeout -k 10 240 python -u scripts/diag/impute_leak.py $ARGS > $out/scen_$name.log 2>&1 ); rc=$?
  grep -h "RESULT" $out/scen_$name.log >> $out/scen.log
  echo "rc $rc" >> $out/scen.log
  [ $rc -eq 0 ]
}
ARGS="fp16,bf16 10000 200 5";      s reverse_order DIAG_SHORT=1 DIAG_VERBOSE=1 || exit 1
ARGS="bf16,bf16,bf16 10000 200 5"; s bf16_only DIAG_SHORT=1 || exit 1
ARGS="bf16,fp16 10000 200 5";      s eager DIAG_SHORT=1 DIAG_EAGER=1 || exit 1
ARGS="bf16,fp16 10000 200 5 1";    s one_lane DIAG_SHORT=1 || exit 1
ARGS="bf16,fp16 10000 200 5";      s no_probe DIAG_SHORT=1 STDIFF_LANE_PROBE=0 || exit 1
cat $out/scen.log
