#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== diag batch invariance"; timeout 300 python tools/diag_batch_invariance.py 512 8 2>&1 | tail -9
echo "=== bf16lab PIPE4"; LWG_LAB_PIPE4=1 timeout 600 python tools/bf16lab.py --no-f32 --convs-only 2>&1 | tee gpurun_out/bf16lab_pipe4.txt | tail -18
echo "=== rocprof bf16 1024"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_bf16" -o lwg -- python "$OLDPWD/bench.py" --precision bf16 --size 1024 --workload novel_view --steps 2 --warmup 1 --no-extras --cpu-frames 0 --no-conv-events > "$OLDPWD/gpurun_out/prof_bf16_bench.log" 2>&1 )
f=$(find gpurun_out/prof_bf16 -name "*kernel_stats*" | head -1); [ -n "$f" ] && python tools/prof_summary.py "$f" 270 30 | tee gpurun_out/prof_bf16_summary.txt
echo "=== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -30 gpurun_out/pytest_gpu.log
echo "=== bench default"; timeout 900 python bench.py --steps 6 --warmup 2 > gpurun_out/bench_call3.json 2> gpurun_out/bench_call3.err; echo "bench exit $?"; tail -c 5000 gpurun_out/bench_call3.json; tail -3 gpurun_out/bench_call3.err
