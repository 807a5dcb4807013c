#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_pers" -o pers -- python "$R/bench_personalize.py" --steps 10 --warmup 3 > "$R/gpurun_out/prof_pers.log" 2>&1 )
tail -1 gpurun_out/prof_pers.log | cut -c1-200
f=$(find gpurun_out/prof_pers -name "*kernel_trace*" | head -1); python tools/trace_gaps.py "$f" 0.35
g=$(find gpurun_out/prof_pers -name "*kernel_stats*" | head -1); python tools/prof_summary.py "$g" 18 26 | tee gpurun_out/r02_personalize_kernel_stats.txt
find gpurun_out/prof_pers -type f -size +3M -delete
