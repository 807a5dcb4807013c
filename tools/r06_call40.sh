#!/bin/bash
# Round-6 call 40: block timeline of the transposed Winograd kernel after the exchange-slot permutation / hazard fix / XCD order
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
timeout 300 python tools/up4lab.py --lib tools/lab/ctw_ts.so --ts --frames 48 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_aw_convt_block_timeline.txt
cat gpurun_out/r06_aw_convt_block_timeline.txt | cut -c1-700
