set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f16
timeout -k 10 600 python -u -m pytest tests/test_gpu_model.py -x -q -s --timeout 120 --timeout-method thread -k "forward_golden or teacher_forced or final_quality or ragged" > gpurun_out/f16/tests.log 2>&1 || { tail -30 gpurun_out/f16/tests.log; exit 1; }
grep -E "rel err|passed|failed" gpurun_out/f16/tests.log
timeout -k 10 200 python bench.py --dtype fp16 --no-cpu-baseline --no-train > gpurun_out/f16/bench_fp16.json 2> gpurun_out/f16/bench_fp16.err
cat gpurun_out/f16/bench_fp16.json
timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-train > gpurun_out/f16/bench_bf16.json 2> gpurun_out/f16/bench_bf16.err
cat gpurun_out/f16/bench_bf16.json
