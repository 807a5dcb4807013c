#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=$PWD; O=gpurun_out; mkdir -p $O
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bf16 -o bf16 -- python $R/bench.py --precision bf16 --size 1024 --workload novel_view --steps 2 --warmup 1 --no-extras --cpu-frames 0 --no-self-check > $R/$O/prof_bf16.log 2>&1 )
python tools/prof_summary.py $O/prof_bf16/bf16_kernel_stats.csv 3 16 2>/dev/null | cut -c1-200 | tee $O/r06_e_kernel_stats_bf16_1024_fused.txt
find $O/prof_bf16 -type f -size +3M -delete
