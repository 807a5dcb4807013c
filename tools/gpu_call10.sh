#!/bin/bash
# LDS bank probe, graph-capture traceback, per-shape step breakdown, hit-list rasterizer (parity + time), rocprofv3 kernel stats (csv)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "=== lds bank probe"; timeout 60 build/lds_bank_probe 2>&1 | tee gpurun_out/lds_bank_probe.txt
echo "=== raster parity"; timeout 900 python -m pytest tests -m gpu -x -q -k "raster or pipeline or smoke or novel" 2>&1 | tail -3
echo "=== personalize default"; timeout 600 python bench_personalize.py --steps 10 --warmup 3 2>gpurun_out/pers_default.err | tail -1 | cut -c1-400; grep -A12 "Raised at" gpurun_out/pers_default.err | head -30
echo "=== personalize breakdown"; timeout 600 python bench_personalize.py --breakdown 2>/dev/null | tail -64
echo "=== rocprof headline"; ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_r02_f32" -o r02_f32 -- python "$R/bench.py" --steps 5 --warmup 2 --no-extras --cpu-frames 0 --no-conv-events > "$R/gpurun_out/prof_r02_f32.log" 2>&1 ); tail -1 gpurun_out/prof_r02_f32.log | cut -c1-300
f=$(find gpurun_out/prof_r02_f32 -name "*kernel_stats*" | head -1); echo "stats file: $f"; python tools/prof_summary.py "$f" 7 24 2>&1 | head -40
find gpurun_out/prof_r02_f32 -type f -size +4M -delete
