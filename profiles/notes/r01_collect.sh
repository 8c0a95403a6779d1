#!/bin/bash
# How the files under profiles/ are produced (run on the GPU box through gpurun from the repository root):
#   bash profiles/collect.sh <tag> [dtype]        e.g.  bash profiles/collect.sh i fp16
# Output goes to gpurun_out/<tag>/; the summaries are then copied into profiles/ as r01_<tag>_* (see the end).
# Every GPU step has its own time limit and the chain stops at the first failure.
set -e
TAG=${1:-x}
DT=${2:-fp16}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B1="bench.py --dtype $DT --no-graph --lanes 1 --steps 1 --warmup 0 --no-train --no-cpu-baseline"
timeout -k 10 300 python3 bench.py --dtype $DT > $OUT/bench_$DT.json 2> $OUT/bench_$DT.err
cat $OUT/bench_$DT.json
# kernel trace + stats of the default command (sampling leg at 2 lanes, roofline launches, training leg)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --dtype $DT --steps 20 --warmup 5 --no-cpu-baseline > $OUT/kt.log 2>&1
# PMC passes, one counter group per run (gfx950: FETCH_SIZE and WRITE_SIZE do not share a pass)
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $B1 > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $B1 > $OUT/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o p -- python3 $B1 > $OUT/mfma.log 2>&1
ls $OUT $OUT/kt
