#!/bin/bash
# One GPU-box session: GPU tests, the default bench line, the 2-rank rehearsal.  Every GPU step has its own time limit;
# after a fault / abort / timeout nothing further is started.
# usage: scripts/gpu_round.sh <tag> [steps...]   steps: tests bench reh
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
ok_rc() { [ "$1" -eq 0 ] || [ "$1" -eq 1 ]; }     # 1 = ordinary test / assertion failure: later steps may still run
for step in "$@"; do
  case $step in
    tests)
      timeout -k 10 1020 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread -p no:cacheprovider > $out/pytest.log 2>&1
      rc=$?; tail -15 $out/pytest.log; echo "tests rc=$rc"; ok_rc $rc || exit $rc ;;
    bench)
      timeout -k 10 420 python -u bench.py > $out/bench.json 2> $out/bench.err
      rc=$?; tail -c 3000 $out/bench.json; tail -5 $out/bench.err; echo "bench rc=$rc"; ok_rc $rc || exit $rc ;;
    reh)
      STDIFF_BENCH_REHEARSAL=1 STDIFF_BENCH_DUMP=150 timeout -k 10 420 python -u bench.py --gpus 2 --spots 131072 --steps 3 --warmup 1 --no-alt > $out/reh.out 2> $out/reh.err
      rc=$?; tail -c 2500 $out/reh.out; tail -30 $out/reh.err; echo "rehearsal rc=$rc"; ok_rc $rc || exit $rc ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
