#!/bin/bash
# How the files under profiles/ are produced (on the GPU box, from the repository root, through gpurun):
#   scripts/collect_profiles.sh <tag> [dtype]
# Output: gpurun_out/<tag>/; tools/pmc_traffic.py, tools/roofline_check.py and tools/mfma_busy.py then make the
# summaries that are committed under profiles/ as r<round>_<tag>_*.  Every GPU step has its own time limit; the chain stops
# at the first failure.  PMC passes run one counter group each, without any trace option.
set -e
TAG=${1:-x}
DT=${2:-fp16}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the default command (what the driver runs)
timeout -k 10 400 python3 bench.py --dtype $DT > $OUT/bench_$DT.json 2> $OUT/bench_$DT.err
cut -c1-600 $OUT/bench_$DT.json
# kernel trace + stats of the same workload (sampling leg at 2 lanes, then the roofline launches)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --dtype $DT --steps 5 --warmup 2 --no-alt --no-train --no-cpu-baseline > $OUT/kt.log 2>&1
# PMC passes over one eager step of two row slabs + the roofline launches (45 x 4 GEMM dispatches of a full slab)
B1="bench.py --dtype $DT --spots 262144 --no-graph --lanes 1 --steps 1 --warmup 0 --no-alt --no-train --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $B1 > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $B1 > $OUT/write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o p -- python3 $B1 > $OUT/mfma.log 2>&1
ls $OUT $OUT/kt
