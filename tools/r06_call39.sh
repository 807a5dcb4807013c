#!/bin/bash
# Round-6 call 39: F(4x4,3x3) lab: every second workgroup of an XCD starts half a block late (de-phased output bursts; variant -DLWG_W4_STAGGER=1) against the tree
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_av_wino4_stagger.txt; : > $O
echo "== bench A/B (A = staggered variant)" >> $O
tools/ab_bench.sh tools/lab/liblwg_w4_stagger.so --steps 5 --warmup 2 >> $O 2>&1
cat $O
