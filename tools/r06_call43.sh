#!/bin/bash
# Round-6 call 43: F(4x4,3x3) residual / xn loads non-temporal (variant -DW4_NT_RES=2) against the tree
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ba_wino4_res_nt.txt; : > $O
tools/ab_bench.sh tools/lab/liblwg_w4_res2.so --steps 5 --warmup 2 >> $O 2>&1
cat $O
