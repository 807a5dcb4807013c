#!/bin/bash
# lab: kernel stats of the captured personalization step + the per-shape conv breakdown of one eager step
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
bash tools/prof_pers.sh > gpurun_out/pers_stats.txt 2>&1
python bench_personalize.py --steps 4 --warmup 2 --breakdown > gpurun_out/pers_breakdown.json 2> gpurun_out/pers_breakdown.err
tail -3 gpurun_out/pers_stats.txt
