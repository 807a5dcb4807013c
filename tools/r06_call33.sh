#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -k "check_winograd_mode" 2>&1 | tail -40 > gpurun_out/r06_dbg.txt
timeout 300 python bench.py --steps 2 --warmup 1 --no-extras --cpu-frames 0 2>&1 | tail -12 >> gpurun_out/r06_dbg.txt
cat gpurun_out/r06_dbg.txt
