#!/bin/bash
# Round-6 call 45: F(4x4,3x3) lab: every second launch walks its tiles backwards (variant -DLWG_W4_REVERSE=1) against the tree
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_bc_wino4_reverse.txt; : > $O
tools/ab_bench.sh tools/lab/liblwg_w4_rev.so --steps 5 --warmup 2 >> $O 2>&1
cat $O
